"""Diagnostic (GPU box): why do GPU and oracle differ at the unexplained pixels of a sweep case?
   python scripts/dbg/explain.py [wide] seed [seed ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scripts.parity_sweep import case_of, case_of_wide
from tests import parity_utils as PU
from oracle import p3d_restate as O

wide = sys.argv[1] == "wide"
for seed in map(int, sys.argv[2 if wide else 1:]):
    c = case_of_wide(seed) if wide else case_of(seed)
    res = PU.run_parity_case(**c)
    print("seed", seed, c, {k: v for k, v in res.items() if k != "unexplained"})
    case = PU.make_case(c["n_env"], seed, c["mesh"], c["az_range"])
    S, K = c["img"], c.get("faces_per_pixel", 100)
    got = PU.run_engine(case, c["img"], radius=c["radius"], faces_per_pixel=K)
    print("   engine: loss0 %s loss %s reward %s" % (got["loss0"].tolist(), got["loss"].tolist(), got["reward"].tolist()))
    for (i, phase, kind, o, y, x, err) in res["unexplained"][:6]:
        env = PU.oracle_env(case, i, S, "flat", K)
        env.reset(radius=c["radius"], azimuth=float(case["az"][i]))
        if phase == "step":
            env.step(case["actions"][i].clone())
        R, T = env.R[0], env.T[0]
        if kind == "alpha":
            v, f = env.objs[o]
        else:
            v, f = env.scene
        ndc = O.world_to_ndc(v, R, T)
        fv = ndc[f]
        x_, y_ = fv[..., 0], fv[..., 1]
        area = (x_[:, 2] - x_[:, 0]) * (y_[:, 1] - y_[:, 0]) - (y_[:, 2] - y_[:, 0]) * (x_[:, 1] - x_[:, 0])
        yf = -1 + (2 * (S - 1 - y) + 1) / S
        xf = -1 + (2 * (S - 1 - x) + 1) / S
        sq = O.BLUR_RADIUS ** 0.5
        inb = (x_.min(1).values - 2 * sq <= xf) & (xf <= x_.max(1).values + 2 * sq) & (y_.min(1).values - 2 * sq <= yf) & (yf <= y_.max(1).values + 2 * sq)
        near = inb & (area.abs() < 1e-6)
        print(" ", (i, phase, kind, o, y, x, err), "faces near pixel with |area|<1e-6:", [(int(j), float(area[j])) for j in torch.nonzero(near).reshape(-1)[:12]])
        blur = O.BLUR_RADIUS if kind == "alpha" else 0.0
        pc = O.pixel_candidates(fv, S, y, x, blur, band=1e-2)
        order = np.argsort(pc["z"])
        cand = (pc["flags"] & 2) != 0
        zc = np.sort(pc["z"][cand])
        print("    n_cand", int(cand.sum()), "rows", len(order))
        if kind == "alpha" and zc.size > K:
            print("    z[K-3..K+3]:", zc[K - 3:K + 3], "gaps", np.diff(zc[K - 3:K + 3]))
            fz = pc["f"][cand][np.argsort(pc["z"][cand])]
            for ff in fz[max(K - 2, 0):K + 2]:
                print("      face %d around the K boundary: conditioning bound of its depth %.3e" % (int(ff), PU.sliver_depth_bound(fv[int(ff)].detach().numpy())))
        nb = np.abs(pc["dist"] - O.BLUR_RADIUS) / O.BLUR_RADIUS
        idx = np.argsort(nb)[:4]
        print("    closest |dist-blur|/blur:", [(int(pc["f"][j]), float(nb[j]), int(pc["flags"][j])) for j in idx])
        mb = np.argsort(np.abs(pc["minb"]))[:4]
        print("    smallest |minb|:", [(int(pc["f"][j]), float(pc["minb"][j]), float(pc["z"][j]), int(pc["flags"][j])) for j in mb])
        if kind == "obs":
            ins = (pc["flags"] & 1) != 0
            zi = np.sort(pc["z"][ins])
            print("    inside depths:", zi[:5], " gpu obs", got["obs0" if phase == "reset" else "obs"][i][:, y, x].tolist())
            ob = env.reset(radius=c["radius"], azimuth=float(case["az"][i])) if phase == "reset" else None
            if ob is not None:
                print("    oracle obs", ob.reshape(4, S, S)[:, y, x].tolist())
            for j in np.nonzero(ins)[0][np.argsort(pc["z"][ins])][:3]:
                ff = int(pc["f"][j]); v = fv[ff].detach().numpy().astype(np.float64)
                e = [float(np.hypot(*(v[(k + 1) % 3, :2] - v[k, :2]))) for k in range(3)]
                print("    inside face %d: z %.7f minb %.3e area %.4e edges %s zs %s" % (ff, pc["z"][j], pc["minb"][j], float(area[ff]), np.round(e, 5).tolist(), v[:, 2].round(5).tolist()))
