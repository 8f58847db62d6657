"""Diagnostic (CPU): per-pixel d alpha / d(el, az) of the engine (dump of scripts/dbg/dump_grad_case.py) against the f64
forward-mode emulation on the oracle's geometry: WHERE does the action gradient differ?
    python scripts/dbg/locate_grad_diff.py seed:mesh:img:az:radius dump.npz [env]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import p3d_restate as O
from tests import parity_utils as PU
from scripts.dbg.fwd_grad_emul import ndc_and_tangents, emulate

parts = sys.argv[1].split(":")
seed, mesh, img, azr, radius = int(parts[0]), parts[1], int(parts[2]), float(parts[3]), float(parts[4])
D = np.load(sys.argv[2])
i = int(sys.argv[3]) if len(sys.argv) > 3 else 0
S = img
case = PU.make_case(2, seed, mesh, azr, device="cpu")
e32 = PU.oracle_env(case, i, S)
env = O.OracleEnv([(v.double(), f) for v, f in e32.objs], S, dtype=torch.float64)
env.reset(radius=radius, azimuth=float(case["az"][i]))
a = case["actions"][i].clone().double().requires_grad_(True)
env.step(a)
el, az = float(env.elevation), float(env.azimuth)
al64 = [im[0, ..., 3].detach().double().numpy() for im in env.alphas]
for o, (v, f) in enumerate(e32.objs):
    fv, tan, nb = ndc_and_tangents(v, f.long(), el, az, radius, torch.float64)
    ndc64 = O.world_to_ndc(v.double(), env.R[0].detach(), env.T[0].detach())
    fvc, c2u, nbb, _, _ = O.clip_faces(ndc64[f.long()], O.Z_CLIP, True)
    p2f, zb, _, _ = O._Rasterize.apply(fvc.contiguous(), nbb, S, float(O.BLUR_RADIUS), 100, True, True, True)
    prod, sums, _ = emulate(fv, tan, p2f, S, torch.float64, "plain")
    dal = (-(prod / 1e-4))[None] * sums  # (2,S,S)
    eng = D["obj_grad"][i][o].astype(np.float64)  # (S,S,2)
    ea = D["alphas"][i][o].astype(np.float64)
    cover = (ea > 0) | (al64[o] > 0)
    diff = np.abs(eng.transpose(2, 0, 1) - dal) * cover[None]
    cnt = (p2f.numpy() >= 0).sum(-1)
    print("object %d: |d alpha| sum %.3e   sum |diff| %.3e   max diff %.3e   alpha max diff %.2e" % (
        o, np.abs(dal).sum(), diff.sum(), diff.max(), np.abs(ea - al64[o]).max()))
    idx = np.argsort(-diff.max(0).ravel())[:8]
    for k in idx:
        y, x = divmod(int(k), S)
        nbv = nbb[p2f[y, x][p2f[y, x] >= 0]] if nbb is not None else None
        npair = int((nbv >= 0).sum()) if nbv is not None else 0
        print("   px (%3d,%3d) cands %3d (clipped-pair faces %d)  alpha eng %.6f orc %.6f  dalpha eng (%.4e, %.4e) orc (%.4e, %.4e)" % (
            y, x, cnt[y, x], npair, ea[y, x], al64[o][y, x], eng[y, x, 0], eng[y, x, 1], dal[0, y, x], dal[1, y, x]))
