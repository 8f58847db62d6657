"""Why does the upstream rule (tests/parity_utils.py: _Upstream) accept or refuse a pixel?  (GPU box)
   python scripts/dbg/explain_upstream.py [wide] SEED ENV PHASE(reset|step) OBJ Y X"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
from parity_sweep import case_of, case_of_wide
from tests import parity_utils as PU
from oracle import p3d_restate as O

a = sys.argv[1:]
wide = a[0] == "wide"
a = a[1:] if wide else a
seed, i, phase, o, y, x = int(a[0]), int(a[1]), a[2], int(a[3]), int(a[4]), int(a[5])
c = case_of_wide(seed) if wide else case_of(seed)
case = PU.make_case(c["n_env"], seed, c["mesh"], c["az_range"])
S, K = c["img"], c.get("faces_per_pixel", 100)
got = PU.run_engine(case, S, radius=c["radius"], faces_per_pixel=K)
env = PU.oracle_env(case, i, S, "flat", K)
env.reset(radius=c["radius"], azimuth=float(case["az"][i]))
if phase == "step":
    env.step(case["actions"][i].clone())
faces = PU._Faces(env.objs[o][0], env.objs[o][1], env.R[0], env.T[0])
rec = got["records0" if phase == "reset" else "records"][3 * i + o]
ok, worst, bounds = PU.upstream_check(faces, rec)
print("records %d, upstream ok %s, worst |diff| / bound %.3f; faces with bound > TVERT: %d, largest bound %.3e" % (
    rec["fv"].shape[0], ok, worst, int((bounds > PU.TVERT).sum()), float(bounds.max())))
al = PU.alpha_of_records(rec, S, K)
ga = got["alphas0" if phase == "reset" else "alphas"][i, o]
oa = torch.stack([im[0, ..., 3] for im in env.alphas]).detach()[o]
print("pixel (%d,%d): engine %.6f  oracle %.6f  oracle's raster of the engine's records %.6f" % (y, x, float(ga[y, x]), float(oa[y, x]), float(al[y, x])))
d = (al - ga).abs()
print("whole object: max |engine - raster of its records| %.3e at %s; pixels beyond TOL: %d" % (float(d.max()), np.unravel_index(int(d.argmax()), d.shape), int((d > PU.TOL).sum())))
cnd = O.pixel_candidates(faces.fv, S, y, x, O.BLUR_RADIUS, band=10 * PU.TB_REL, area_band=PU.TAREA, vert_band=PU.TVERT)
orig = cnd["f"] if faces.c2u is None else faces.c2u.numpy()[cnd["f"]]
print("candidates at the pixel (orig face, z, dist, bound):", [(int(f), round(float(z), 5), float(dd), float(bounds[f])) for f, z, dd in zip(orig, cnd["z"], cnd["dist"])][:12])
# worst records
ev = rec["fv"].double().numpy()
ofv = faces.fv.double().numpy()
c2u = None if faces.c2u is None else faces.c2u.numpy()
rows = []
for j in range(ev.shape[0]):
    k = int(rec["ids"][j])
    cs = [k] if c2u is None else np.nonzero(c2u == k)[0].tolist()
    best = min((float(np.abs(ev[j, :, None, :2] - ofv[cc, None, :, :2]).max(-1).min(1).max()), cc) for cc in cs) if cs else (9.0, -1)
    rows.append((best[0] / bounds[k], j, k, int(rec["flags"][j]), cs, best[1]))
rows.sort(reverse=True)
for ratio, j, k, fl, cs, cc in rows[:5]:
    print("record %d id %d flags %d: oracle pieces %s, best match %d at %.3f of the bound %.3e" % (j, k, fl, cs, cc, ratio, bounds[k]))
    print("   engine", ev[j].round(6).tolist())
    for c_ in cs:
        print("   oracle", c_, ofv[c_].round(6).tolist())
    print("   unclipped", faces.fv_unclipped[k].numpy().round(6).tolist())
# the checker's own matching (parity_utils.upstream_check), record by record
first = np.arange(ofv.shape[0]) if c2u is None else None
if first is None:
    first = np.full(bounds.shape[0], -1, dtype=np.int64)
    for c in range(c2u.shape[0] - 1, -1, -1):
        if c2u[c] >= 0:
            first[c2u[c]] = c
for j in range(ev.shape[0]):
    k = int(rec["ids"][j]); c = int(first[k]) + (1 if rec["flags"][j] & 2 else 0)
    dm = np.abs(ev[j, :, None, :] - ofv[c, None, :, :]); m = dm[..., :2].max(-1).argmin(1)
    d = dm[np.arange(3), m, :2].max(); dz = dm[np.arange(3), m, 2].max()
    if d / bounds[k] > 1 or dz / (2 * PU.TVIEW) > 1:
        print("checker: record %d id %d flags %d -> piece %d: d %.3e (bound %.3e) dz %.3e; engine z %s oracle z %s" % (j, k, int(rec["flags"][j]), c, d, bounds[k], dz, ev[j, :, 2].tolist(), ofv[c, :, 2].tolist()))
