"""Diagnose one parity-sweep case at pixel level (GPU box): python scripts/dbg/explain_case.py SEED ENV OBJ Y X [Y X ...]
For every pixel: the oracle's candidates (count, alpha over the K nearest, alpha over all, alpha over K-1 / K+1) next to
the engine's alpha, before (reset) and after the step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
from parity_sweep import case_of
from tests.parity_utils import make_case, run_engine, oracle_env, _Faces
from oracle import p3d_restate as O

seed, env_i, obj = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pix = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(4, len(sys.argv), 2)]
c = case_of(seed)
case = make_case(c["n_env"], seed, c["mesh"], c["az_range"])
S, K = c["img"], 100
got = run_engine(case, S, radius=c["radius"])
env = oracle_env(case, env_i, S)
env.reset(radius=c["radius"], azimuth=float(case["az"][env_i]))
a = case["actions"][env_i].clone().requires_grad_(True)
env.step(a)
al = torch.stack([im[0, ..., 3] for im in env.alphas]).detach()
faces = _Faces(env.objs[obj][0], env.objs[obj][1], env.R[0], env.T[0])
for (y, x) in pix:
    cd = O.pixel_candidates(faces.fv, S, y, x, O.BLUR_RADIUS, band=0.0)
    cand = (cd["flags"] & 2) != 0
    z, d, f = cd["z"][cand], cd["dist"][cand], cd["f"][cand]
    inside = ((cd["flags"] & 1) != 0)[cand]
    order = np.lexsort((f, z))
    z, d, f, inside = z[order], d[order], f[order], inside[order]
    sd = np.where(inside, -d, d).astype(np.float64)
    p = 1.0 / (1.0 + np.exp(sd / 1e-4))
    def alpha(k): return 1.0 - np.prod(1.0 - p[:k])
    n = len(z)
    print("pixel (%d,%d): oracle candidates %d; alpha oracle %.6f  engine %.6f  | K %.6f  K-1 %.6f  K+1 %.6f  all %.6f" % (
        y, x, n, float(al[obj, y, x]), float(got["alphas"][env_i, obj, y, x]), alpha(K), alpha(K - 1), alpha(K + 1), alpha(n)))
    if n > K:
        print("    z around the K boundary:", " ".join("%.7f/f%d/p%.3g" % (z[i], f[i], p[i]) for i in range(max(0, K - 3), min(n, K + 3))))
    # which single omission / addition explains the engine's value?
    tgt = float(got["alphas"][env_i, obj, y, x])
    base = np.prod(1.0 - p[:K])
    best = min(((abs((1 - base / max(1 - p[i], 1e-30)) - tgt), "without #%d (f%d z %.6f p %.4g)" % (i, f[i], z[i], p[i])) for i in range(min(n, K))), default=(9, ""))
    best2 = min(((abs((1 - base * (1 - p[i])) - tgt), "with extra #%d (f%d z %.6f p %.4g)" % (i, f[i], z[i], p[i])) for i in range(K, n)), default=(9, ""))
    print("    closest single change:", "%.2e %s" % best, "|", "%.2e %s" % best2)
