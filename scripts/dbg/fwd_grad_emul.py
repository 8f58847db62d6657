"""Diagnostic (CPU only, oracle + numpy): where does the forward-mode action gradient lose its digits?

Emulates the engine's forward-mode sweep (occ_eval.hpp: eval_face's GRAD part; occ_raster2.hpp's per-pixel sums;
occ_combine.hpp's dI/d theta; the final reduction) on the ORACLE's own fragments (pix_to_face of the K nearest, clipped
NDC faces), once in f32 and once in f64, in several arrangements of the arithmetic, and compares every one with the
f64 oracle's autograd gradient in units of eps * M (tests/parity_utils.py: gradient_mass).

    python scripts/dbg/fwd_grad_emul.py seed:mesh:img:az_range:radius[:env] ...
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import p3d_restate as O  # noqa: E402
from tests import parity_utils as PU  # noqa: E402

EPS = 2.0 ** -24


def ndc_and_tangents(verts, faces, el, az, radius, dt):
    """Clipped NDC faces (Fc,3,3) and their tangents d/d(el), d/d(az) (2,Fc,3,3) by forward-mode AD in dtype dt."""
    import torch.autograd.forward_ad as fwAD

    out = []
    for which in range(2):
        with fwAD.dual_level():
            e = torch.tensor([el], dtype=dt)
            a = torch.tensor([az], dtype=dt)
            one = torch.ones(1, dtype=dt)
            e = fwAD.make_dual(e, one if which == 0 else torch.zeros(1, dtype=dt))
            a = fwAD.make_dual(a, one if which == 1 else torch.zeros(1, dtype=dt))
            r = torch.tensor([radius], dtype=dt)
            C = torch.stack([r * torch.sin(a) * torch.cos(e), r * torch.sin(a) * torch.sin(e), r * torch.cos(a)], dim=1)
            R = O.look_at_rotation(C)
            T = O.translation_from(R, C)
            ndc = O.world_to_ndc(verts.to(dt), R[0], T[0])
            fvc, c2u, nb, _, _ = O.clip_faces(ndc[faces], O.Z_CLIP, True)
            p, t = fwAD.unpack_dual(fvc)
            out.append((p.detach(), t.detach() if t is not None else torch.zeros_like(p), nb))
    return out[0][0], torch.stack([out[0][1], out[1][1]]), out[0][2]


def emulate(fv, tan, p2f, S, dt, variant):
    """Per-pixel (prod, sum_el, sum_az) over the listed faces.  fv (Fc,3,3), tan (2,Fc,3,3), p2f (S,S,K) clipped ids."""
    f = np.float32 if dt == torch.float32 else np.float64
    fv = fv.numpy().astype(f)
    tn = tan.numpy().astype(f)
    idx = p2f.numpy()
    valid = idx >= 0
    ii = np.where(valid, idx, 0)
    ys, xs = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    xf = (f(-1.0) + (f(2.0) * (S - 1 - xs).astype(f) + f(1.0)) / f(S))[..., None]
    yf = (f(-1.0) + (f(2.0) * (S - 1 - ys).astype(f) + f(1.0)) / f(S))[..., None]
    v = fv[ii]  # (S,S,K,3,3)
    x0, y0, z0 = v[..., 0, 0], v[..., 0, 1], v[..., 0, 2]
    x1, y1, z1 = v[..., 1, 0], v[..., 1, 1], v[..., 1, 2]
    x2, y2, z2 = v[..., 2, 0], v[..., 2, 1], v[..., 2, 2]
    dx0, dy0, dx1, dy1, dx2, dy2 = xf - x0, yf - y0, xf - x1, yf - y1, xf - x2, yf - y2
    ex01, ey01, ex02, ey02, ex12, ey12 = x1 - x0, y1 - y0, x2 - x0, y2 - y0, x2 - x1, y2 - y1
    area = (x2 - x0) * (y1 - y0) - (y2 - y0) * (x1 - x0)
    ia = f(1.0) / (area + f(1e-8))
    b0 = (dx1 * ey12 - dy1 * ex12) * ia
    b1 = (dy2 * ex02 - dx2 * ey02) * ia
    b2 = (dx0 * ey01 - dy0 * ex01) * ia
    w0, w1, w2 = b0 * z1 * z2, z0 * b1 * z2, z0 * z1 * b2
    den = np.maximum(w0 + w1 + w2, f(1e-8))
    p0, p1, p2 = w0 / den, w1 / den, w2 / den
    inside = (p0 > 0) & (p1 > 0) & (p2 > 0)
    l01 = ex01 * ex01 + ey01 * ey01
    l02 = ex02 * ex02 + ey02 * ey02
    l12 = ex12 * ex12 + ey12 * ey12

    def seg(ex, ey, dx, dy, l2):
        t = np.clip((ex * dx + ey * dy) / np.where(l2 > 1e-8, l2, 1), 0, 1)
        t = np.where(l2 > 1e-8, t, 1).astype(f)
        qx, qy = t * ex - dx, t * ey - dy
        return qx * qx + qy * qy

    d01, d02, d12 = seg(ex01, ey01, dx0, dy0, l01), seg(ex02, ey02, dx0, dy0, l02), seg(ex12, ey12, dx1, dy1, l12)
    dist = np.minimum(np.minimum(d01, d02), d12)
    s01 = (d01 <= d02) & (d01 <= d12)
    s02 = ~s01 & (d02 <= d01) & (d02 <= d12)
    s12 = ~s01 & ~s02
    sd = np.where(inside, -dist, dist)
    e = np.exp(sd / f(1e-4))
    p = f(1.0) / (f(1.0) + e)
    q = f(1.0) - p
    bax = np.where(s01, ex01, np.where(s02, ex02, ex12))
    bay = np.where(s01, ey01, np.where(s02, ey02, ey12))
    pax = np.where(s12, dx1, dx0)
    pay = np.where(s12, dy1, dy0)
    l2 = np.where(s01, l01, np.where(s02, l02, l12))
    if variant.startswith("recip"):  # the engine's arithmetic: 1 / |edge|^2 stored in the record, t = dot * (il / (1 + eps il))
        il = (f(1.0) / l2).astype(f)
        ile = (il * (f(1.0) / (f(1.0) + f(1e-8) * il)).astype(f)).astype(f)
        dotv = (bax * pax + bay * pay).astype(f)
        tb = np.clip(dotv * ile, 0, 1).astype(f)
        if variant == "recipfix":  # one Newton step on the quotient: r = dot - t D, t += r y
            D = (l2 + f(1e-8)).astype(f)
            t0 = (dotv * ile).astype(f)
            r = (dotv.astype(np.float64) - t0.astype(np.float64) * D.astype(np.float64)).astype(f)  # fma: exact product, one rounding
            tb = np.clip((t0.astype(np.float64) + r.astype(np.float64) * ile.astype(np.float64)).astype(f), 0, 1).astype(f)
        variant = "plain"
    else:
        tb = np.clip((bax * pax + bay * pay) / (l2 + f(1e-8)), 0, 1).astype(f)
    gx, gy = f(2.0) * (tb * bax - pax), f(2.0) * (tb * bay - pay)
    ia_ = np.where(s12, 1, 0)
    ib_ = np.where(s01, 1, 2)
    t4 = tn[:, ii]  # (2,S,S,K,3,3)
    sel = lambda arr, k: np.take_along_axis(arr, k[None, ..., None, None].repeat(2, 0).repeat(1, -1), axis=-2)  # noqa: E731
    # vertex tangents (dx, dy) of vertex a / b of the closest edge, for el and az
    tax = np.take_along_axis(t4[..., 0], np.broadcast_to(ia_[None, ..., None], t4.shape[:-2] + (1,)), -1)[..., 0]
    tay = np.take_along_axis(t4[..., 1], np.broadcast_to(ia_[None, ..., None], t4.shape[:-2] + (1,)), -1)[..., 0]
    tbx = np.take_along_axis(t4[..., 0], np.broadcast_to(ib_[None, ..., None], t4.shape[:-2] + (1,)), -1)[..., 0]
    tby = np.take_along_axis(t4[..., 1], np.broadcast_to(ib_[None, ..., None], t4.shape[:-2] + (1,)), -1)[..., 0]
    mx = tax + tb * (tbx - tax)  # (2,S,S,K)
    my = tay + tb * (tby - tay)
    sp = np.where(inside, -p, p) * valid
    qv = np.where(valid, q, f(1.0))
    prod = np.prod(qv, axis=-1, dtype=f)
    if variant == "plain":
        g = (sp * (gx * mx + gy * my)).astype(f)  # (2,S,S,K)
        sums = np.zeros(g.shape[:-1], f)
        for k in range(g.shape[-1]):  # sequential f32 accumulation like a lane's RMW
            sums = (sums + g[..., k]).astype(f)
    elif variant == "common":
        # common-mode: reference tangent per pixel = tangent of the FIRST listed face's nearest point; H = sum p G in small numbers
        mrx, mry = mx[..., :1], my[..., :1]
        hx = np.zeros(prod.shape, f)
        hy = np.zeros(prod.shape, f)
        res = np.zeros(mx.shape[:-1], f)
        for k in range(mx.shape[-1]):
            hx = (hx + sp[..., k] * gx[..., k]).astype(f)
            hy = (hy + sp[..., k] * gy[..., k]).astype(f)
            res = (res + sp[..., k] * (gx[..., k] * (mx[..., k] - mrx[..., 0]) + gy[..., k] * (my[..., k] - mry[..., 0]))).astype(f)
        sums = (hx * mrx[..., 0] + hy * mry[..., 0] + res).astype(f)
    elif variant == "f64sum":  # f32 terms, f64 accumulation: is it the terms or the sums?
        g = (sp * (gx * mx + gy * my)).astype(f)
        sums = g.astype(np.float64).sum(-1)
    else:
        raise ValueError(variant)
    return prod, sums, np.abs(sp[None] * (gx * mx + gy * my)).astype(np.float64).sum(-1)


def run(spec):
    parts = spec.split(":")
    seed, mesh, img, azr, radius = int(parts[0]), parts[1], int(parts[2]), float(parts[3]), float(parts[4])
    envs = [int(parts[5])] if len(parts) > 5 else [0, 1]
    case = PU.make_case(2, seed, mesh, azr, device="cpu")
    S = img
    for i in envs:
        # the f64 oracle: state after the step, gradient by autograd
        e32 = PU.oracle_env(case, i, S)
        g = {}
        frag = {}
        for dt in (torch.float64, torch.float32):
            env = O.OracleEnv([(v.to(dt), f) for v, f in e32.objs], S, dtype=dt)
            env.reset(radius=radius, azimuth=float(case["az"][i]))
            a = case["actions"][i].clone().to(dt).requires_grad_(True)
            _, r, _, _ = env.step(a)
            r.backward()
            g[dt] = a.grad.double().numpy()
            frag[dt] = env
        env = frag[torch.float64]
        el, az = float(env.elevation), float(env.azimuth)
        # Jacobian of (el, az) w.r.t. the action
        a0 = case["actions"][i].double()
        n = a0.norm()
        J = 0.05 * (torch.eye(2, dtype=torch.float64) / n - torch.outer(a0, a0) / n ** 3).numpy()
        om = float(env.objectMass)
        al = [im[0, ..., 3].detach().double().numpy() for im in env.alphas]
        I = al[0] * al[1] + al[1] * al[2] + al[0] * al[2]
        gsum = [al[1] + al[2], al[0] + al[2], al[0] + al[1]]
        print("case %s env %d: |g64| %.4e   f32 oracle autograd vs f64: %.3e" % (spec, i, np.linalg.norm(g[torch.float64]),
              np.linalg.norm(g[torch.float32] - g[torch.float64])), flush=True)
        for dt, variant in ((torch.float64, "plain"), (torch.float32, "plain"), (torch.float32, "recip"), (torch.float32, "recipfix")):
            net = np.zeros(2)
            mass = np.zeros(2)
            fine = np.zeros(2)
            for o, (v, f) in enumerate(e32.objs):
                fv, tan, nb = ndc_and_tangents(v, f.long(), el, az, radius, dt)
                # fragments of the f64 oracle (the same K nearest for every variant): clipped face ids
                ndc64 = O.world_to_ndc(v.double(), env.R[0].detach(), env.T[0].detach())
                fvc, c2u, nbb, _, _ = O.clip_faces(ndc64[f.long()], O.Z_CLIP, True)
                p2f, _, _, _ = O._Rasterize.apply(fvc.contiguous(), nbb, S, float(O.BLUR_RADIUS), 100, True, True, True)
                prod, sums, fmass = emulate(fv, tan, p2f, S, dt, variant)
                dal = (-(prod / 1e-4))[None] * sums  # (2,S,S)
                term = (2 * I * gsum[o])[None] * dal.astype(np.float64)
                if dt == torch.float32 and variant != "f64sum":
                    t32 = ((2 * I * gsum[o]).astype(np.float32)[None] * dal.astype(np.float32))
                    acc = np.float32(0) * np.zeros(2, np.float32)
                    # blockwise f32 sums like the combine / reduce kernels (256-pixel blocks, then the blocks)
                    blk = t32.reshape(2, -1, 256).sum(-1, dtype=np.float32)
                    net += blk.sum(-1, dtype=np.float32)
                else:
                    net += term.sum((1, 2))
                mass += np.abs(term).sum((1, 2))
                fine += ((2 * I * gsum[o])[None] * (prod / 1e-4)[None] * fmass).sum((1, 2))
            ga = -(J.T @ net) / om
            Ma = np.linalg.norm(np.abs(J).T @ mass) / om
            Mf = np.linalg.norm(np.abs(J).T @ fine) / om
            err = np.linalg.norm(ga - g[torch.float64])
            print("   %-8s %-7s |g| %.4e  err vs f64 autograd %.3e = %.1f eps*M   (M %.3e, fine-grained mass %.3e = %.0f x M)" % (
                str(dt).split(".")[1], variant, np.linalg.norm(ga), err, err / (EPS * Ma), Ma, Mf, Mf / Ma), flush=True)


if __name__ == "__main__":
    for s in sys.argv[1:]:
        run(s)
