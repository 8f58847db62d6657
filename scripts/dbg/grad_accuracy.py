"""Diagnostic (GPU box): action gradient of the HIP path and of the f32 oracle against the f64 oracle.
   python scripts/dbg/grad_accuracy.py seed [seed ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scripts.parity_sweep import case_of
from tests import parity_utils as PU
from oracle import p3d_restate as O

for seed in map(int, sys.argv[1:]):
    c = case_of(seed)
    case = PU.make_case(c["n_env"], seed, c["mesh"], c["az_range"])
    got = PU.run_engine(case, c["img"], radius=c["radius"])
    for i in range(c["n_env"]):
        g = {}
        for dt in (torch.float32, torch.float64):
            env = PU.oracle_env(case, i, c["img"])
            env.__init__([(v.to(dt), f) for v, f in env.objs], c["img"], dtype=dt)
            env.reset(radius=c["radius"], azimuth=float(case["az"][i]))
            a = case["actions"][i].clone().to(dt).requires_grad_(True)
            _, r, _, _ = env.step(a)
            r.backward()
            g[dt] = a.grad.double()
        gg = got["grad"][i].double()
        n = g[torch.float64].norm()
        print("seed %d env %d |g64| %.3e  gpu-vs-64 %.2e  orc32-vs-64 %.2e  gpu-vs-orc32 %.2e" % (
            seed, i, float(n), float((gg - g[torch.float64]).norm() / n), float((g[torch.float32] - g[torch.float64]).norm() / n),
            float((gg - g[torch.float32]).norm() / n)), flush=True)
