"""Diagnostic (CPU): the engine's dumped face records (scripts/dbg/dump_grad_case.py) against the f64 oracle's clipped NDC
faces and their tangents.    python scripts/dbg/cmp_records.py seed:mesh:img:az:radius dump.npz [env]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import p3d_restate as O
from tests import parity_utils as PU
from scripts.dbg.fwd_grad_emul import ndc_and_tangents

parts = sys.argv[1].split(":")
seed, mesh, img, azr, radius = int(parts[0]), parts[1], int(parts[2]), float(parts[3]), float(parts[4])
D = np.load(sys.argv[2])
i = int(sys.argv[3]) if len(sys.argv) > 3 else 0
case = PU.make_case(2, seed, mesh, azr, device="cpu")
e32 = PU.oracle_env(case, i, img)
cam = D["cam"][i]
el, az = float(cam[43]), float(cam[44])
print("engine el/az", el, az, "cam C", cam[12:15])
for o, (v, f) in enumerate(e32.objs):
    rec = D["rec%d" % (3 * i + o)]
    fv, tan, nb = ndc_and_tangents(v, f.long(), el, az, radius, torch.float64)
    # original face of every clipped face
    ndc = fv  # clipped
    import torch.autograd.forward_ad as fwAD
    # recompute c2u
    C = torch.tensor([[radius * np.sin(az) * np.cos(el), radius * np.sin(az) * np.sin(el), radius * np.cos(az)]], dtype=torch.float64)
    R = O.look_at_rotation(C); T = O.translation_from(R, C)
    nd = O.world_to_ndc(v.double(), R[0], T[0])
    fvc, c2u, nbb, _, _ = O.clip_faces(nd[f.long()], O.Z_CLIP, True)
    if c2u is None:
        c2u = torch.arange(fvc.shape[0])
    first = {}
    for j, u in enumerate(c2u.tolist()):
        first.setdefault(u, j)
    ids = rec[:, 9].view(np.int32); flags = rec[:, 10].view(np.int32)
    idx = np.array([first[int(u)] + (1 if (fl & 2) else 0) for u, fl in zip(ids, flags)])
    pos_e = np.stack([rec[:, [0, 1, 2]], rec[:, [3, 4, 5]], rec[:, [6, 7, 8]]], 1).astype(np.float64)  # (n,3,3)
    pos_o = fvc[idx].numpy()
    tan_e = rec[:, 20:32].reshape(-1, 3, 4).astype(np.float64)  # per vertex dx/del dy/del dx/daz dy/daz
    tan_o = np.stack([tan[0][idx][..., 0].numpy(), tan[0][idx][..., 1].numpy(), tan[1][idx][..., 0].numpy(), tan[1][idx][..., 1].numpy()], -1)
    dp = np.abs(pos_e - pos_o)
    dt = np.abs(tan_e - tan_o)
    mag = np.abs(tan_o).max()
    clipped = (flags & 4) != 0
    print("obj %d: %d records (%d z-clipped)  pos err max %.2e (xy %.2e)  tangent: max |t| %.2f  err max %.2e  rel-to-own max %.2e  mean %.2e" % (
        o, len(rec), int(clipped.sum()), dp.max(), dp[..., :2].max(), mag, dt.max(), (dt / np.maximum(np.abs(tan_o), 1e-3)).max(), dt.mean()))
    if clipped.any():
        print("   clipped only: pos xy err max %.2e  tangent err max %.2e ; unclipped: pos %.2e tan %.2e" % (
            dp[clipped][..., :2].max(), dt[clipped].max(), dp[~clipped][..., :2].max() if (~clipped).any() else 0, dt[~clipped].max() if (~clipped).any() else 0))
        w = np.argsort(-dt.reshape(len(rec), -1).max(1))[:5]
        for j in w:
            print("     rec %d id %d flags %d  z %s  tan_e %s tan_o %s" % (j, ids[j], flags[j], pos_e[j][:, 2].round(3), tan_e[j].round(4).tolist(), tan_o[j].round(4).tolist()))
