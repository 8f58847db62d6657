"""Which face records does the ENGINE evaluate as candidates at a pixel, and which of them does the oracle not have?
   python scripts/dbg/explain_records.py SEED ENV OBJ Y X   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
from parity_sweep import case_of
from tests.parity_utils import make_case, run_engine, oracle_env, _Faces
from oracle import p3d_restate as O

seed, env_i, obj, y, x = [int(v) for v in sys.argv[1:6]]
c = case_of(seed)
case = make_case(c["n_env"], seed, c["mesh"], c["az_range"])
S = c["img"]
got = run_engine(case, S, radius=c["radius"])
eng = got["engine"]
torch.cuda.synchronize()
eo = env_i * 3 + obj
nrec = eng._ws_tensors["nrec"].cpu().numpy()[: eng.NT * 3]
rec_off = eng._rec_tensors["rec_off"].cpu().numpy().view(np.int64)
n = int(nrec[eo]); base = int(rec_off[eo])
rec = eng._rec_tensors["rec"].view(torch.float32).cpu().numpy().reshape(-1, 32)[base: base + n]
ids = rec[:, 9].view(np.int32); flags = rec[:, 10].view(np.int32)
print("engine records of (env %d, obj %d): %d" % (env_i, obj, n))
env = oracle_env(case, env_i, S)
env.reset(radius=c["radius"], azimuth=float(case["az"][env_i]))
a = case["actions"][env_i].clone().requires_grad_(True)
env.step(a)
faces = _Faces(env.objs[obj][0], env.objs[obj][1], env.R[0], env.T[0])
# the oracle's evaluation of the ENGINE's records (their own NDC vertices) at the pixel
fv_eng = torch.from_numpy(np.stack([rec[:, [0, 1, 2]], rec[:, [3, 4, 5]], rec[:, [6, 7, 8]]], axis=1).copy())
ce = O.pixel_candidates(fv_eng, S, y, x, O.BLUR_RADIUS, band=0.0, cull_backfaces=False)
co = O.pixel_candidates(faces.fv, S, y, x, O.BLUR_RADIUS, band=0.0)
ce_c = {int(ids[f]): (float(z), float(d)) for f, z, d, fl in zip(ce["f"], ce["z"], ce["dist"], ce["flags"]) if fl & 2}
co_c = {int(f): (float(z), float(d)) for f, z, d, fl in zip(co["f"], co["z"], co["dist"], co["flags"]) if fl & 2}
print("pixel (%d,%d): candidates by the oracle on the engine's records %d, by the oracle on its own faces %d" % (y, x, len(ce_c), len(co_c)))
print("  only in the engine's records:", {k: v for k, v in ce_c.items() if k not in co_c})
print("  only in the oracle's faces:  ", {k: v for k, v in co_c.items() if k not in ce_c})
for k in [k for k in ce_c if k not in co_c][:4]:
    j = int(np.nonzero(ids == k)[0][0])
    print("  record %d (face id %d, flags %d): verts" % (j, k, flags[j]), rec[j, :9].round(6).tolist(), "inv_area %.6g" % rec[j, 11])
    fo = faces.fv[k] if k < faces.fv.shape[0] else None
    if fo is not None:
        v = fo.numpy()
        area = (v[2, 0] - v[0, 0]) * (v[1, 1] - v[0, 1]) - (v[2, 1] - v[0, 1]) * (v[1, 0] - v[0, 0])
        print("     oracle's face %d: verts" % k, v.reshape(-1).round(6).tolist(), "signed area %.3e" % area)
# coordinate noise between the two fp32 pipelines: engine record vertices vs the oracle's NDC vertices of the same face id
# (unclipped faces only: flags == 0 and id within the oracle's list)
sel = [j for j in range(n) if flags[j] == 0 and ids[j] < faces.fv.shape[0]]
ev = rec[sel, :9].astype(np.float64); ov = faces.fv[ids[sel]].numpy().reshape(-1, 9).astype(np.float64)
d = np.abs(ev - ov)
print("vertex noise over %d unclipped records: max |dx| %.3e |dy| %.3e |dz| %.3e; mean %.3e" % (len(sel), d[:, [0, 3, 6]].max(), d[:, [1, 4, 7]].max(), d[:, [2, 5, 8]].max(), d.mean()))
