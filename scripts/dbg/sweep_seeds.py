"""Re-run given parity-sweep seeds (GPU box): python scripts/dbg/sweep_seeds.py [wide] 2084 2352   [OCC_HIP_LIB selects the library]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts"))
from occlusionenv_amd import _native as nat
if os.environ.get("OCC_HIP_LIB"):  # older builds of this round lack the operator-level entry points added later
    for k in ("occ_rasterize_meshes_tiled", "occ_rasterize_meshes_backward"):
        nat.SYMBOLS.pop(k, None)
from parity_sweep import case_of, case_of_wide
from tests import parity_utils as PU
from tests.parity_utils import run_parity_case, violations
from collections import Counter
wide = len(sys.argv) > 1 and sys.argv[1] == "wide"
for seed in map(int, sys.argv[2 if wide else 1:]):
    c = case_of_wide(seed) if wide else case_of(seed)
    PU.REASON_LOG = []
    res = run_parity_case(**c)
    v = violations(res)
    print(os.path.basename(os.environ.get("OCC_HIP_LIB", "HEAD")), seed, c, "BAD" if v else "ok", "alpha %.2e obs0 %.2e obs %.2e" % (res["alpha_maxabs"], res["obs0_maxabs"], res["obs_maxabs"]), v[:2],
          {k: res[k] for k in ("tie_pixels", "tie_decisions", "upstream_pixels", "fs_maxabs", "fs_arith", "alpha0_maxabs") if k in res}, flush=True)
    print("   reasons:", dict(Counter((k,) + w for k, o, y, x, w in PU.REASON_LOG)), flush=True)
    print("   pixels:", sorted(set((k, o, y, x) for k, o, y, x, w in PU.REASON_LOG))[:40], flush=True)
