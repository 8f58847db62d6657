"""Re-run given parity-sweep seeds (GPU box): python scripts/dbg/sweep_seeds.py 2084 2352   [OCC_HIP_LIB selects the library]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts"))
from occlusionenv_amd import _native as nat
if os.environ.get("OCC_HIP_LIB"):  # older builds of this round lack the operator-level entry points added later
    for k in ("occ_rasterize_meshes_tiled", "occ_rasterize_meshes_backward"):
        nat.SYMBOLS.pop(k, None)
from parity_sweep import case_of
from tests.parity_utils import run_parity_case, violations
for seed in map(int, sys.argv[1:]):
    c = case_of(seed)
    res = run_parity_case(**c)
    v = violations(res)
    print(os.path.basename(os.environ.get("OCC_HIP_LIB", "HEAD")), seed, c, "BAD" if v else "ok", "alpha %.2e obs0 %.2e obs %.2e" % (res["alpha_maxabs"], res["obs0_maxabs"], res["obs_maxabs"]), v[:2], flush=True)
