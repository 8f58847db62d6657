"""Diagnostic (GPU box): one wide-sweep case through the parity check, printing the gradient arbiter's numbers.
   [OCC_HIP_LIB=other.so] python scripts/dbg/one_case.py SEED [SEED ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scripts.parity_sweep import case_of_wide
from tests.parity_utils import run_parity_case, violations

for seed in map(int, sys.argv[1:]):
    c = case_of_wide(seed)
    res = run_parity_case(**c)
    print(seed, c, "violations:", violations(res))
    for a in res["grad_arbiter"]:
        print("   env %d rel32 %.3e |g64| %.3e mass %.3e e_gpu %.3e (%.0f eps*M) e_orc32 %.3e (%.0f eps*M) bound %.3e ok %s" % (
            a["env"], a["rel32"], a["g64"], a["mass"], a["e_gpu"], a["e_gpu"] / max(2.0 ** -24 * a["mass"], 1e-300), a["e_orc32"],
            a["e_orc32"] / max(2.0 ** -24 * a["mass"], 1e-300), a["bound"], a["ok"]))
