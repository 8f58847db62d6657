"""Diagnostic (CPU): the forward-mode gradient recomputed in f64 / f32 from the ENGINE's dumped records (positions and
tangents as the setup kernel wrote them) on the f64 oracle's K-nearest sets: is the excess in the gradient arithmetic
or in the records?    python scripts/dbg/emul_engine_records.py seed:mesh:img:az:radius dump.npz [env]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import p3d_restate as O
from tests import parity_utils as PU
from scripts.dbg.fwd_grad_emul import ndc_and_tangents, emulate, EPS

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
_, r, _, _ = env.step(a)
r.backward()
g64 = a.grad.numpy()
el, az = float(env.elevation), float(env.azimuth)
cam = D["cam"][i]
print("oracle f64 el/az %.9f %.9f   engine %.9f %.9f" % (el, az, cam[43], cam[44]))
a0 = case["actions"][i].double(); n = a0.norm()
J = 0.05 * (torch.eye(2, dtype=torch.float64) / n - torch.outer(a0, a0) / n ** 3).numpy()
om = float(env.objectMass)
al = [im[0, ..., 3].detach().double().numpy() for im in env.alphas]
I = al[0] * al[1] + al[1] * al[2] + al[0] * al[2]
gsum = [al[1] + al[2], al[0] + al[2], al[0] + al[1]]
print("|g64| %.5e  engine grad err %.3e" % (np.linalg.norm(g64), np.linalg.norm(D["grad"][i] - g64)))
for label in ("oracle64-geometry", "engine-records", "engine-pos+oracle-tan", "oracle-pos+engine-tan"):
    net = np.zeros(2); mass = np.zeros(2)
    for o, (v, f) in enumerate(e32.objs):
        fv, tan, nb = ndc_and_tangents(v, f.long(), el, az, radius, torch.float64)
        ndc64 = O.world_to_ndc(v.double(), env.R[0].detach(), env.T[0].detach())
        fvc, c2u, nbb, _, _ = O.clip_faces(ndc64[f.long()], O.Z_CLIP, True)
        p2f, _, _, _ = O._Rasterize.apply(fvc.contiguous(), nbb, S, float(O.BLUR_RADIUS), 100, True, True, True)
        if label != "oracle64-geometry":
            rec = D["rec%d" % (3 * i + o)]
            if c2u is None:
                c2u = torch.arange(fvc.shape[0])
            first = {}
            for j, u in enumerate(c2u.tolist()):
                first.setdefault(u, j)
            ids = rec[:, 9].view(np.int32); flags = rec[:, 10].view(np.int32)
            idx = np.array([first[int(u)] + (1 if (fl & 2) else 0) for u, fl in zip(ids, flags)], dtype=np.int64)
            fv2 = fv.clone(); tan2 = tan.clone()
            pos_e = np.stack([rec[:, [0, 1, 2]], rec[:, [3, 4, 5]], rec[:, [6, 7, 8]]], 1).astype(np.float64)
            te = rec[:, 20:32].reshape(-1, 3, 4).astype(np.float64)
            if label in ("engine-records", "engine-pos+oracle-tan") and len(idx):
                fv2[idx] = torch.from_numpy(pos_e)
            if label in ("engine-records", "oracle-pos+engine-tan") and len(idx):
                tan2[0][idx, :, 0] = torch.from_numpy(te[..., 0]); tan2[0][idx, :, 1] = torch.from_numpy(te[..., 1])
                tan2[1][idx, :, 0] = torch.from_numpy(te[..., 2]); tan2[1][idx, :, 1] = torch.from_numpy(te[..., 3])
            fv, tan = fv2, tan2
        prod, sums, _ = emulate(fv, tan, p2f, S, torch.float64, "plain")
        dal = (-(prod / 1e-4))[None] * sums
        term = (2 * I * gsum[o])[None] * dal
        net += term.sum((1, 2)); mass += np.abs(term).sum((1, 2))
    ga = -(J.T @ net) / om
    Ma = np.linalg.norm(np.abs(J).T @ mass) / om
    err = np.linalg.norm(ga - g64)
    print("  %-24s err vs f64 autograd %.3e = %.1f eps*M" % (label, err, err / (EPS * Ma)))
