"""tests/grad_explain.py -- TEST INFRASTRUCTURE ONLY: per-pixel explanation of a failed action-gradient comparison.

``tests/parity_utils.py`` compares d reward / d action in aggregate (1e-4 against the f32 oracle, else the f64 oracle
arbitrates with the fp32 noise floor of the sum).  Round 5's wide sweep (seeds 9000-9319) found cases that fail that
with every image inside tolerance.  This module does for the gradient what the tie classifier does for the images: the
engine's per-object, per-pixel d alpha / d(el, az) planes are compared with an f64 FORWARD-MODE evaluation of the oracle's
own fragments (the K nearest faces of every pixel as the f64 oracle lists them; the same formulas as PyTorch3D's backward
pushed forward - checked against the oracle's autograd every time it runs), and every pixel that differs must be EXPLAINED
by one of two near-ties, from the oracle's own numbers, or the case fails:

CLOSEST-EDGE TIE.  dists is the minimum over the three edges of a face; its gradient flows through the ARGMIN edge
 and jumps where two edges are equally far: on the bisector of a corner.  Two squared distances d_a <= d_b
whose difference is below the effect of the positional noise both sides carry - the f32 pixel centre
``-1 + (2 (S-1-i) + 1) / S`` is off by up to an ulp of a number in [1, 2) (TCENTRE = 1.2e-7: seed 9276, pixel (63, 112),
yf = 0.20625 + 6e-8 moved d_01 and d_02 by 1.2e-5 of themselves in opposite directions where they differ by 1.6e-5 - the
engine's arithmetic and the oracle's C code, both valid f32, fell on different sides) plus the vertex noise of
``parity_utils.face_noise_bounds`` - can be ordered either way: ``d_b - d_a <= 4 sqrt(d_b) (TCENTRE + bound(face))``.
Only matters where faces are large on screen (the camera inside the scene: one pixel's term then carries per cent of the
gradient); a flip between sub-pixel faces moves nothing.

NEAR / Z-CLIPPED FACE.  The rule of the images (parity_utils: NEAR AND Z-CLIPPED FACES) carried over: accepted only if
(1) every engine record of the object equals the oracle's face within its noise bound, (2) the f64 forward-mode evaluation
ON THE ENGINE'S OWN RECORDS (positions and tangents as the setup kernel wrote them) reproduces the engine's d alpha at
that pixel - the raster stage's gradient arithmetic agrees on identical geometry - and (3) a face whose bound exceeds
TVERT is a candidate there.

Explained pixels get weight 0 in the loss on both sides, like image ties, and the aggregate criterion is applied again.
"""
from __future__ import annotations

import numpy as np
import torch

TCENTRE = 1.2e-7     # absolute error of an f32 pixel-centre coordinate: one ulp of a number in [1, 2)
TGRAD_PIX = 1e-3     # (2) of NEAR / Z-CLIPPED FACE: the records' own gradient reproduces the engine's d alpha at the pixel to this (relative)
TGRAD_SHARE = 0.1    # a pixel is examined if its d alpha difference ALONE moves the action gradient by more than this x TOL of its norm
GRAD_TIE_FRAC = 2e-4  # explained gradient-tie pixels per env: at most this share of the object-pixels (floor 4), as TIE_FRAC


def ndc_and_tangents(verts, faces, el, az, radius, dt=torch.float64):
    """Clipped NDC faces (Fc,3,3), their tangents d/d(el), d/d(az) (2,Fc,3,3) by torch forward-mode AD through the
    oracle's own camera / projection / clip_faces, clipped->original map and neighbour array."""
    import torch.autograd.forward_ad as fwAD

    from oracle import p3d_restate as O

    out = []
    for which in range(2):
        with fwAD.dual_level():
            one, zero = torch.ones(1, dtype=dt), torch.zeros(1, dtype=dt)
            e = fwAD.make_dual(torch.tensor([el], dtype=dt), one if which == 0 else zero)
            a = fwAD.make_dual(torch.tensor([az], dtype=dt), one if which == 1 else zero)
            r = torch.tensor([radius], dtype=dt)
            C = torch.stack([r * torch.sin(a) * torch.cos(e), r * torch.sin(a) * torch.sin(e), r * torch.cos(a)], dim=1)
            R = O.look_at_rotation(C)
            T = O.translation_from(R, C)
            ndc = O.world_to_ndc(verts.to(dt), R[0], T[0])
            fvc, c2u, nb, _, _ = O.clip_faces(ndc[faces], O.Z_CLIP, True)
            p, t = fwAD.unpack_dual(fvc)
            out.append((p.detach(), t.detach() if t is not None else torch.zeros_like(p), c2u, nb))
    return out[0][0], torch.stack([out[0][1], out[1][1]]), out[0][2], out[0][3]


def forward_planes(fv, tan, p2f, S, face_bound=None):
    """f64 forward-mode sweep over the listed faces of every pixel (eval_face's GRAD part).
    fv (Fc,3,3), tan (2,Fc,3,3) numpy f64, p2f (S,S,K) clipped face ids (-1 = empty).  Returns
    (prod (S,S), dalpha (2,S,S), edge_tie (S,S) bool): edge_tie marks pixels with a candidate whose two nearest edges are
    closer than 4 sqrt(d) (TCENTRE + face_bound[face]) (face_bound: per clipped face, default TVERT of parity_utils)."""
    f = np.float64
    idx = np.asarray(p2f)
    valid = idx >= 0
    ii = np.where(valid, idx, 0)
    ys, xs = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    xf = (-1.0 + (2.0 * (S - 1 - xs) + 1.0) / S)[..., None]
    yf = (-1.0 + (2.0 * (S - 1 - ys) + 1.0) / S)[..., None]
    v = fv[ii]
    x0, y0, z0 = v[..., 0, 0], v[..., 0, 1], v[..., 0, 2]
    x1, y1, z1 = v[..., 1, 0], v[..., 1, 1], v[..., 1, 2]
    x2, y2, z2 = v[..., 2, 0], v[..., 2, 1], v[..., 2, 2]
    dx0, dy0, dx1, dy1, dx2, dy2 = xf - x0, yf - y0, xf - x1, yf - y1, xf - x2, yf - y2
    ex01, ey01, ex02, ey02, ex12, ey12 = x1 - x0, y1 - y0, x2 - x0, y2 - y0, x2 - x1, y2 - y1
    area = (x2 - x0) * (y1 - y0) - (y2 - y0) * (x1 - x0)
    ia = 1.0 / (area + 1e-8)
    b0 = (dx1 * ey12 - dy1 * ex12) * ia
    b1 = (dy2 * ex02 - dx2 * ey02) * ia
    b2 = (dx0 * ey01 - dy0 * ex01) * ia
    w0, w1, w2 = b0 * z1 * z2, z0 * b1 * z2, z0 * z1 * b2
    den = np.maximum(w0 + w1 + w2, 1e-8)
    inside = (w0 / den > 0) & (w1 / den > 0) & (w2 / den > 0)
    l01, l02, l12 = ex01 * ex01 + ey01 * ey01, ex02 * ex02 + ey02 * ey02, ex12 * ex12 + ey12 * ey12

    def seg(ex, ey, dx, dy, l2):
        ok = l2 > 1e-8
        t = np.where(ok, np.clip((ex * dx + ey * dy) / np.where(ok, l2, 1.0), 0.0, 1.0), 1.0)
        qx, qy = t * ex - dx, t * ey - dy
        return qx * qx + qy * qy, qx, qy

    with np.errstate(over="ignore", invalid="ignore"):
        (d01, qx01, qy01), (d02, qx02, qy02), (d12, qx12, qy12) = (seg(ex01, ey01, dx0, dy0, l01), seg(ex02, ey02, dx0, dy0, l02),
                                                                   seg(ex12, ey12, dx1, dy1, l12))
        dist = np.minimum(np.minimum(d01, d02), d12)
        s01 = (d01 <= d02) & (d01 <= d12)
        s02 = ~s01 & (d02 <= d01) & (d02 <= d12)
        s12 = ~s01 & ~s02
        sd = np.where(inside, -dist, dist)
        p = 1.0 / (1.0 + np.exp(sd / 1e-4))
        bax = np.where(s01, ex01, np.where(s02, ex02, ex12))
        bay = np.where(s01, ey01, np.where(s02, ey02, ey12))
        pax, pay = np.where(s12, dx1, dx0), np.where(s12, dy1, dy0)
        l2 = np.where(s01, l01, np.where(s02, l02, l12))
        tb = np.clip((bax * pax + bay * pay) / (l2 + 1e-8), 0.0, 1.0)
        gx, gy = 2.0 * (tb * bax - pax), 2.0 * (tb * bay - pay)
        t4 = tan[:, ii]  # (2,S,S,K,3,3)
        ia_, ib_ = np.where(s12, 1, 0), np.where(s01, 1, 2)

        def pick(comp, which):
            return np.take_along_axis(t4[..., comp], np.broadcast_to(which[None, ..., None], t4.shape[:-2] + (1,)), -1)[..., 0]

        tax, tay, tbx, tby = pick(0, ia_), pick(1, ia_), pick(0, ib_), pick(1, ib_)
        mx, my = tax + tb * (tbx - tax), tay + tb * (tby - tay)
        sp = np.where(inside, -p, p) * valid
        prod = np.prod(np.where(valid, 1.0 - p, 1.0), axis=-1)
        sums = (sp[None] * (gx[None] * mx + gy[None] * my)).sum(-1)
    dalpha = (-(prod / 1e-4))[None] * sums
    # closest-edge near-ties: the two smallest of (d01, d02, d12) of a listed face - with DIFFERENT nearest points (two edges
    # whose nearest point is the vertex they share are equally far by construction and give the same gradient: no decision)
    dstack = np.stack([d01, d02, d12], -1)
    order = np.argsort(dstack, axis=-1)
    ds = np.take_along_axis(dstack, order, -1)
    qxs = np.take_along_axis(np.stack([qx01, qx02, qx12], -1), order, -1)
    qys = np.take_along_axis(np.stack([qy01, qy02, qy12], -1), order, -1)
    if face_bound is None:
        bound = np.full(idx.shape, 2.5e-7)
    else:
        bound = np.asarray(face_bound, dtype=f)[ii]
    noise = TCENTRE + bound
    gap_ok = (ds[..., 1] - ds[..., 0]) <= 4.0 * np.sqrt(ds[..., 1]) * noise
    apart = np.hypot(qxs[..., 1] - qxs[..., 0], qys[..., 1] - qys[..., 0]) > 8.0 * noise
    edge_tie = (gap_ok & apart & valid).any(-1)
    return prod, dalpha, edge_tie


def oracle_planes(env64, objs, S, K, radius):
    """Per object of an f64 OracleEnv that has just stepped: (fragments, f64 forward planes on the oracle's geometry).
    Returns a list of dicts(fv, tan, c2u, nb, p2f, prod, dalpha, edge_tie, bounds)."""
    from oracle import p3d_restate as O
    from tests.parity_utils import face_noise_bounds

    el, az = float(env64.elevation.detach()), float(env64.azimuth.detach())
    out = []
    for v, f in objs:
        fv, tan, c2u, nb = ndc_and_tangents(v, f.long(), el, az, radius)
        ndc = O.world_to_ndc(v.double(), env64.R[0].detach(), env64.T[0].detach())
        bounds_u = face_noise_bounds(ndc[f.long()])  # per ORIGINAL face
        if fv.shape[0] == 0:
            out.append(None)
            continue
        p2f, _, _, _ = O._Rasterize.apply(fv.contiguous(), nb, S, float(O.BLUR_RADIUS), K, True, True, True)
        cb = bounds_u if c2u is None else bounds_u[c2u.numpy()]
        prod, dal, tie = forward_planes(fv.numpy(), tan.numpy(), p2f, S, face_bound=cb)
        out.append(dict(fv=fv, tan=tan, c2u=c2u, nb=nb, p2f=p2f, prod=prod, dalpha=dal, edge_tie=tie, bounds=cb, bounds_u=bounds_u))
    return out


def planes_on_engine_records(op, rec, S):
    """(2) of NEAR / Z-CLIPPED FACE: the f64 forward sweep over the ORACLE's lists with the ENGINE's own records (positions
    and tangents) in place of the oracle's faces.  op: one entry of oracle_planes; rec: snapshot_records entry with "tan"."""
    fv, tan = op["fv"].clone().numpy(), op["tan"].clone().numpy()
    c2u = op["c2u"].numpy() if op["c2u"] is not None else np.arange(fv.shape[0])
    first = {}
    for j, u in enumerate(c2u.tolist()):
        first.setdefault(u, j)
    ids, flags = rec["ids"], rec["flags"]
    keep = np.array([int(u) in first for u in ids], dtype=bool)
    idx = np.array([first[int(u)] + (1 if (fl & 2) else 0) for u, fl in zip(ids[keep], flags[keep])], dtype=np.int64)
    if idx.size:
        fv[idx] = rec["fv"].double().numpy()[keep]
        te = rec["tan"].astype(np.float64)[keep]  # (n,3,4): dx/del dy/del dx/daz dy/daz per vertex
        tan[0][idx, :, 0], tan[0][idx, :, 1] = te[..., 0], te[..., 1]
        tan[1][idx, :, 0], tan[1][idx, :, 1] = te[..., 2], te[..., 3]
    _, dal, _ = forward_planes(fv, tan, op["p2f"], S)
    return dal


def explain_gradient(case, i, S, K, radius, got, w, faces_of=None):
    """Which pixels make env i's action gradient differ, and why.  ``got``: run_engine output (obj_grad planes, alphas,
    records with tangents); ``w`` (S,S) the pixel weights already in force.  Returns dict(ties (S,S) bool of explained
    pixels, unexplained list of (obj, y, x, diff), reasons Counter, selfcheck relative error of the f64 forward sweep
    against the f64 oracle's autograd)."""
    import collections

    from oracle import p3d_restate as O
    from tests import parity_utils as PU

    e32 = PU.oracle_env(case, i, S)
    env = O.OracleEnv([(v.double(), f) for v, f in e32.objs], S, dtype=torch.float64)
    env.faces_per_pixel = K
    env.reset(radius=radius, azimuth=float(case["az"][i]))
    a = case["actions"][i].clone().double().requires_grad_(True)
    env.step(a)
    wd = w.double()
    loss = torch.sum(wd * env.image[0, ..., 3] ** 2)
    loss.backward()  # d loss / d action through the f64 oracle's autograd
    planes = oracle_planes(env, e32.objs, S, K, radius)
    al = [im[0, ..., 3].detach().double().numpy() for im in env.alphas]
    I = al[0] * al[1] + al[1] * al[2] + al[0] * al[2]
    gsum = [al[1] + al[2], al[0] + al[2], al[0] + al[1]]
    # self-check: the forward sweep's d loss / d(el, az) through the action Jacobian equals autograd's d loss / d action
    net = np.zeros(2)
    for o, op in enumerate(planes):
        if op is not None:
            net += ((2 * I * gsum[o] * wd.numpy())[None] * op["dalpha"]).sum((1, 2))
    a0 = case["actions"][i].double()
    n = float(a0.norm())
    J = (O.STEP_SIZE * (torch.eye(2, dtype=torch.float64) / n - torch.outer(a0, a0) / n ** 3)).numpy() if n else O.STEP_SIZE * np.eye(2)
    g_fwd = J.T @ net
    g_auto = a.grad.numpy()
    selfcheck = float(np.linalg.norm(g_fwd - g_auto) / max(np.linalg.norm(g_auto), 1e-30))
    ties = torch.zeros(S, S, dtype=torch.bool)
    unexplained, reasons = [], collections.Counter()
    eng_dal = got["obj_grad"][i].double().numpy()  # (3,S,S,2)
    eng_al = got["alphas"][i].double().numpy()
    upstream_ok = {}
    for o, op in enumerate(planes):
        if op is None:
            continue
        ref = op["dalpha"].transpose(1, 2, 0)  # (S,S,2)
        cover = ((eng_al[o] > 0) | (al[o] > 0)) & (w.numpy() > 0)
        diff = (eng_dal[o] - ref) * cover[..., None]
        # what the pixel's difference alone does to d loss / d action: J^T (2 I dI/d alpha_o w diff)
        contrib = np.linalg.norm((((2 * I * gsum[o] * wd.numpy())[..., None] * diff) @ J), axis=-1)
        d = np.abs(diff).max(-1)
        scale = np.maximum(np.abs(ref).max(-1), 1e-3 * np.abs(ref).max())
        for y, x in zip(*np.nonzero(contrib > TGRAD_SHARE * PU.TOL * np.linalg.norm(g_auto))):
            y, x = int(y), int(x)
            why = []
            if op["edge_tie"][y, x]:
                why.append("closest-edge tie")
            else:
                ids = op["p2f"][y, x]
                ids = ids[ids >= 0].numpy()
                if ids.size and (op["bounds"][ids] > PU.TVERT).any():
                    if o not in upstream_ok:
                        faces = faces_of(o) if faces_of is not None else PU._Faces(e32.objs[o][0], e32.objs[o][1], env.R[0].detach().float(), env.T[0].detach().float())
                        ok, _, _ = PU.upstream_check(faces, got["records"][3 * i + o])
                        upstream_ok[o] = (ok, planes_on_engine_records(op, got["records"][3 * i + o], S) if ok else None)
                    ok, dal_rec = upstream_ok[o]
                    if ok and np.abs(dal_rec[:, y, x] - eng_dal[o][y, x]).max() <= TGRAD_PIX * scale[y, x]:
                        why.append("near / z-clipped face: vertex noise upstream, the gradient of the engine's own records agrees")
            if why:
                ties[y, x] = True
                for r in why:
                    reasons[r] += 1
            else:
                unexplained.append((o, y, x, float(d[y, x])))
    return dict(ties=ties, unexplained=unexplained, reasons=reasons, selfcheck=selfcheck)
