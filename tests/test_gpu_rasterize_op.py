"""Operator boundary: occlusionenv_amd.ops.rasterize_meshes (K-buffer in PyTorch3D's layout) vs the oracle's naive
rasteriser.  Index outputs must be identical; float outputs and the dists backward within 1e-5."""
import pytest
import torch

from oracle import p3d_restate as O

pytestmark = pytest.mark.gpu


def _teapot_scene_face_verts(teapot, radius=4.0, az=0.2):
    v, f = teapot
    if radius < 2:  # the second object of the scene (offset (x2, 0, 1)): straddles z = 0.5 for a camera at r = 1.2
        v = v + torch.tensor([0.3, 0.0, 1.0])
    R, T = O.look_at_view_transform(torch.tensor([radius]), torch.tensor([0.0]), torch.tensor([az]))
    ndc = O.world_to_ndc(v, R[0], T[0])
    fv, c2u, nb, conv, cidx = O.clip_faces(ndc[f])
    return fv.contiguous(), nb


@pytest.mark.parametrize("naive", [False, True])
@pytest.mark.parametrize("blur,K,radius", [(O.BLUR_RADIUS, 100, 4.0), (0.0, 1, 4.0), (O.BLUR_RADIUS, 20, 1.2)])
def test_kbuffer_matches_oracle(teapot, blur, K, radius, naive):
    from occlusionenv_amd.ops import rasterize_meshes

    S = 48
    fv, nb = _teapot_scene_face_verts(teapot, radius)
    clipb = blur > 0
    ref = O._Rasterize.apply(fv, nb, S, float(blur), K, True, clipb, True)
    F_ = fv.shape[0]
    got = rasterize_meshes(fv.cuda(), torch.tensor([0]), torch.tensor([F_]), S, blur, K, True, clipb, True,
                           None if nb is None else nb.cuda(), naive=naive)
    p2f, zbuf, bary, dists = [t[0].cpu() for t in got]
    assert torch.equal(p2f, ref[0]), "pix_to_face must be identical (index work is bit-exact)"
    assert torch.allclose(zbuf, ref[1], atol=1e-6) and torch.allclose(dists, ref[3], atol=1e-7)
    assert torch.allclose(bary, ref[2], atol=1e-5)
    if radius < 2:
        assert nb is not None and (nb >= 0).any()  # the clipped-pair rule was exercised


def test_two_meshes_and_backward(teapot):
    from occlusionenv_amd.ops import rasterize_meshes

    S, K = 32, 16
    fv, _ = _teapot_scene_face_verts(teapot)
    fv2 = torch.cat([fv, fv * torch.tensor([0.7, 0.7, 1.0])])  # second mesh: the same, shrunk on screen
    F1 = fv.shape[0]
    x = fv2.cuda().requires_grad_(True)
    p2f, zbuf, bary, dists = rasterize_meshes(x, torch.tensor([0, F1]), torch.tensor([F1, F1]), S, O.BLUR_RADIUS, K, True,
                                              True, True)
    assert p2f.shape == (2, S, S, K) and int(p2f[1][p2f[1] >= 0].min()) >= F1  # packed indices of mesh 1
    w = torch.rand(2, S, S, K, generator=torch.Generator().manual_seed(0)).cuda()
    (dists * w * (p2f >= 0)).sum().backward()
    # oracle, mesh by mesh
    for m in range(2):
        y = fv2[m * F1:(m + 1) * F1].clone().requires_grad_(True)
        r = O._Rasterize.apply(y, None, S, O.BLUR_RADIUS, K, True, True, True)
        assert torch.equal(r[0] + (m * F1) * (r[0] >= 0), p2f[m].cpu())
        (r[3] * w[m].cpu() * (r[0] >= 0)).sum().backward()
        g = x.grad[m * F1:(m + 1) * F1].cpu()
        assert torch.allclose(g, y.grad, atol=1e-4, rtol=1e-3)
        assert float(g[..., 2].abs().max()) == 0.0  # dists do not depend on z


def test_sigmoid_alpha_blend_matches_torch_and_oracle(teapot):
    """rasterize_meshes + sigmoid_alpha_blend = the reference's silhouette renderer at operator level
    (environment.py:258-264): forward and backward against a plain PyTorch fp32 restatement of the blend on the same
    K-buffers, and the image against the oracle's soft_silhouette."""
    from occlusionenv_amd.ops import rasterize_meshes, sigmoid_alpha_blend

    S, K = 48, 100
    fv, nb = _teapot_scene_face_verts(teapot)
    F_ = fv.shape[0]
    x = fv.cuda().requires_grad_(True)
    p2f, zbuf, bary, dists = rasterize_meshes(x, torch.tensor([0]), torch.tensor([F_]), S, O.BLUR_RADIUS, K, True, True, True)
    img = sigmoid_alpha_blend(dists, p2f, O.SIGMA)
    assert img.shape == (1, S, S, 4) and bool((img[..., :3] == 1).all())
    # plain PyTorch fp32 reference of the same op on the same K-buffers
    d_ref = dists.detach().clone().requires_grad_(True)
    prob = torch.sigmoid(-d_ref / O.SIGMA) * (p2f >= 0).float()
    alpha_ref = 1.0 - torch.prod(1.0 - prob, dim=-1)
    assert torch.allclose(img[..., 3], alpha_ref, atol=1e-6)
    w = torch.rand(1, S, S, generator=torch.Generator().manual_seed(1)).cuda()
    (alpha_ref * w).sum().backward()
    d_hip = dists.detach().clone().requires_grad_(True)
    (sigmoid_alpha_blend(d_hip, p2f, O.SIGMA)[..., 3] * w).sum().backward()
    scale = float(d_ref.grad.abs().max())
    assert scale > 0 and float((d_hip.grad - d_ref.grad).abs().max()) <= 1e-4 * scale
    # the whole chain back to the vertices, and the oracle's image of the same mesh
    (img[..., 3] * w).sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0
    r = O._Rasterize.apply(fv, nb, S, float(O.BLUR_RADIUS), K, True, True, True)
    ora = O.sigmoid_alpha_blend(r[3], r[0])
    assert torch.allclose(img[0].detach().cpu(), ora, atol=1e-5)


def test_tiled_kbuffer_is_bit_identical_to_the_naive_one(teapot):
    """occ_rasterize_meshes_tiled (one wave per tile, faces pre-filtered per tile, (depth, face) lists in LDS; 4x4 tiles
    with four faces in flight when there are no clipped pairs, 8x8 tiles in face order when there are) against
    occ_rasterize_meshes_naive (one thread per pixel over all faces): all four outputs bit for bit - two meshes, image
    sides that are no multiple of 8 (or 4), K = 1 / 8 / 100, the z-clipped scene with its clipped pairs, a 5 120-face
    mesh - and faster, also on sub-pixel faces."""
    import time

    from occlusionenv_amd.meshes import SyntheticShapeNet
    from occlusionenv_amd.ops import rasterize_meshes

    def both(fv, first, num, size, blur, K, nb=None, covered=True, reps=1):
        args = (fv.cuda(), first, num, size, blur, K, True, blur > 0, True, None if nb is None else nb.cuda())
        out = []
        for naive in (False, True):
            best = float("inf")
            for _ in range(reps):  # (timed calls: the best of a few - a call may have to wait for the allocator)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = rasterize_meshes(*args, naive=naive)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            out.append((r, best))
        for a, b, name in zip(out[0][0], out[1][0], ("pix_to_face", "zbuf", "bary", "dists")):
            assert torch.equal(a, b), (name, size, K)
        assert not covered or int((out[0][0][0] >= 0).sum()) > 0
        return out[0][1], out[1][1]

    fv, _ = _teapot_scene_face_verts(teapot)
    F1 = fv.shape[0]
    fv2 = torch.cat([fv, fv * torch.tensor([0.7, 0.7, 1.0])])
    for K in (1, 8, 100):
        both(fv2, torch.tensor([0, F1]), torch.tensor([F1, F1]), (44, 52), O.BLUR_RADIUS if K > 1 else 0.0, K)
    fvc, nb = _teapot_scene_face_verts(teapot, radius=1.2)
    assert nb is not None and (nb >= 0).any()
    # (this camera sits inside the teapot: most of what survives the clip is off screen at any size - the lists that do
    # form are compared all the same; the pair rule on covered pixels is what test_kbuffer_matches_oracle's r = 1.2 case checks)
    both(fvc, torch.tensor([0]), torch.tensor([fvc.shape[0]]), 64, O.BLUR_RADIUS, 20, nb, covered=False)
    both(fvc * torch.tensor([0.25, 0.25, 1.0]), torch.tensor([0]), torch.tensor([fvc.shape[0]]), 64, O.BLUR_RADIUS, 20, nb)  # squeezed into view
    v, f = SyntheticShapeNet(n_models=1, seed=5).models[0]
    R, T = O.look_at_view_transform(torch.tensor([4.0]), torch.tensor([0.0]), torch.tensor([0.3]))
    big = O.world_to_ndc(v, R[0], T[0])[f].contiguous()
    bargs = (big, torch.tensor([0]), torch.tensor([big.shape[0]]), 128, O.BLUR_RADIUS, 100)
    both(*bargs)
    # a neighbour array without pairs (what clip_faces hands over for a scene nothing was clipped in): still the order-free
    # kernel, found out per mesh on the device; and a call with one mesh of either kind
    both(big, torch.tensor([0]), torch.tensor([big.shape[0]]), 96, O.BLUR_RADIUS, 100, torch.full((big.shape[0],), -1, dtype=torch.int64))
    sq = fvc * torch.tensor([0.25, 0.25, 1.0])
    both(torch.cat([sq, big]), torch.tensor([0, sq.shape[0]]), torch.tensor([sq.shape[0], big.shape[0]]), 64, O.BLUR_RADIUS, 20,
         torch.cat([nb, torch.full((big.shape[0],), -1, dtype=torch.int64)]))
    both(big, torch.tensor([0]), torch.tensor([big.shape[0]]), (50, 38), O.BLUR_RADIUS, 100)  # sides that are no multiple of 4
    both(torch.cat([big, big * torch.tensor([0.5, 0.5, 1.0])]), torch.tensor([0, big.shape[0]]), torch.tensor([big.shape[0]] * 2), 64, O.BLUR_RADIUS, 3)  # full lists everywhere
    # speed (scripts/dbg/kbuf_time.py, round 4): teapot at 256x256 0.22 vs 2.8 ms; the 5 120-face mesh at 128x128 - a few
    # hundred covered pixels with hundreds of candidates each - 1.2 vs 10.6 ms (round 3's 8x8 tiles in face order: 9-11 ms)
    args = (fv, torch.tensor([0]), torch.tensor([F1]), 256, O.BLUR_RADIUS, 100)
    both(*args)  # warm-up of both kernels
    t_tiled, t_naive = both(*args, reps=3)
    print("teapot, 256x256, K=100: tiled %.2f ms, naive %.2f ms" % (t_tiled * 1e3, t_naive * 1e3))
    assert t_tiled < 2.0 * t_naive  # (a generous ratio: wall-clock on a shared box; the measured ratio is 0.08 - 0.5)
    t_tiled, t_naive = both(*bargs, reps=3)
    print("5 120 faces, 128x128, K=100: tiled %.2f ms, naive %.2f ms" % (t_tiled * 1e3, t_naive * 1e3))
    assert t_tiled < t_naive  # (measured 0.11; generous for a shared box)


def test_zbuf_and_bary_gradients_match_torch_autograd(teapot):
    """The zbuf / bary part of rasterize_meshes' backward (occ_rasterize_meshes_backward; zero on the OcclusionEnv path)
    against torch autograd (f64) of the same formulas - area-normalised edge functions, perspective correction,
    lower-bound clip + renormalisation, interpolated depth - on the K-buffers the forward produced; and
    the three gradient sources add up (dists alone + zbuf / bary alone = all three)."""
    from occlusionenv_amd.ops import rasterize_meshes

    S, K = 40, 6
    fv, _ = _teapot_scene_face_verts(teapot)
    F_ = fv.shape[0]
    g = torch.Generator().manual_seed(3)
    wz, wb, wd = torch.rand(1, S, S, K, generator=g), torch.rand(1, S, S, K, 3, generator=g) - 0.5, torch.rand(1, S, S, K, generator=g)

    def run(use_z, use_b, use_d, blur, clipb):
        x = fv.cuda().requires_grad_(True)
        p2f, zbuf, bary, dists = rasterize_meshes(x, torch.tensor([0]), torch.tensor([F_]), S, blur, K, True, clipb, True)
        m = (p2f >= 0)
        loss = 0.0
        if use_z:
            loss = loss + (zbuf * wz.cuda() * m).sum()
        if use_b:
            loss = loss + (bary * wb.cuda() * m[..., None]).sum()
        if use_d:
            loss = loss + (dists * wd.cuda() * m).sum()
        loss.backward()
        return x.grad.cpu(), p2f.cpu(), zbuf.detach().cpu(), bary.detach().cpu()

    for blur, clipb in ((O.BLUR_RADIUS, True), (0.0, False)):
        g_zb, p2f, zbuf, bary = run(True, True, False, blur, clipb)
        # torch f64 reference on the same (pixel, k) -> face assignment
        y = fv.double().clone().requires_grad_(True)
        idx = torch.nonzero(p2f[0] >= 0)  # (M, 3): yi, xi, k
        f = p2f[0][idx[:, 0], idx[:, 1], idx[:, 2]]
        v = y[f]
        yf = -1.0 + (2.0 * (S - 1 - idx[:, 0]).double() + 1.0) / S
        xf = -1.0 + (2.0 * (S - 1 - idx[:, 1]).double() + 1.0) / S

        def edge(px, py, ax, ay, bx, by):
            return (px - ax) * (by - ay) - (py - ay) * (bx - ax)

        x0, y0, z0, x1, y1, z1, x2, y2, z2 = [v[:, i, j] for i in range(3) for j in range(3)]
        ar = edge(x2, y2, x0, y0, x1, y1) + 1e-8
        p = [edge(xf, yf, x1, y1, x2, y2) / ar, edge(xf, yf, x2, y2, x0, y0) / ar, edge(xf, yf, x0, y0, x1, y1) / ar]
        w = [p[0] * z1 * z2, z0 * p[1] * z2, z0 * z1 * p[2]]
        den = (w[0] + w[1] + w[2]).clamp(min=1e-8)
        p = [wi / den for wi in w]
        if clipb:
            p = [pi.clamp(min=0.0) for pi in p]
            sm = (p[0] + p[1] + p[2]).clamp(min=1e-5)
            p = [pi / sm for pi in p]
        pz = p[0] * z0 + p[1] * z1 + p[2] * z2
        got_z = zbuf[0][idx[:, 0], idx[:, 1], idx[:, 2]].double()
        got_b = bary[0][idx[:, 0], idx[:, 1], idx[:, 2]].double()
        assert torch.allclose(pz.detach(), got_z, atol=1e-5) and torch.allclose(torch.stack(p, 1).detach(), got_b, atol=1e-4)
        sel = (idx[:, 0], idx[:, 1], idx[:, 2])
        loss = (pz * wz[0][sel].double()).sum() + (torch.stack(p, 1) * wb[0][sel].double()).sum()
        loss.backward()
        ref = y.grad
        scale = float(ref.abs().max())
        assert scale > 0
        err = float((g_zb.double() - ref).abs().max())
        assert err <= 2e-4 * scale, (blur, err, scale)
        assert float(g_zb[..., 2].abs().max()) > 0  # depth and perspective correction do depend on z
        # linearity: the gradient of all three outputs = dists alone + (zbuf, bary) alone
        g_d = run(False, False, True, blur, clipb)[0]
        g_all = run(True, True, True, blur, clipb)[0]
        assert torch.allclose(g_all, g_zb + g_d, atol=1e-4 * max(scale, float(g_d.abs().max())), rtol=1e-4)


def test_order_free_kbuffer_equals_the_naive_one_on_random_scenes():
    """occ_rast_quad_fwd_kernel (4x4 tiles, four faces in flight, four partial lists per pixel merged by rank: the kernel
    behind occ_rasterize_meshes_tiled when there are no clipped pairs) against the naive kernel, bit for bit on all four
    outputs, over random scenes: procedural meshes of three densities, cameras near and far, one to three meshes per call,
    sides that are multiples of neither 4 nor 8, K from 1 to 128 (lists that never fill up and lists that overflow at
    every covered pixel), culling on and off, barycentric clipping on and off."""
    from occlusionenv_amd.meshes import SyntheticShapeNet
    from occlusionenv_amd.ops import rasterize_meshes

    g = torch.Generator().manual_seed(2024)
    models = SyntheticShapeNet(n_models=6, seed=9, mixed=True).models
    checked = full = 0
    for case in range(14):
        n_m = 1 + case % 3
        fvs, first, num = [], [], []
        for _ in range(n_m):
            v, f = models[int(torch.randint(0, len(models), (1,), generator=g))]
            radius = float(3.0 + 5.0 * torch.rand(1, generator=g))
            az = float(6.28 * torch.rand(1, generator=g))
            el = float(0.8 * torch.rand(1, generator=g) - 0.4)
            R, T = O.look_at_view_transform(torch.tensor([radius]), torch.tensor([el]), torch.tensor([az]))
            ndc = O.world_to_ndc(2.5 * v + 0.2 * torch.randn(3, generator=g), R[0], T[0])
            assert float(ndc[:, 2].min()) > 0.6  # nothing near the clip plane: no clipped pairs
            first.append(sum(num))
            num.append(int(f.shape[0]))
            fvs.append(ndc[f])
        fv = torch.cat(fvs).contiguous().cuda()
        K = (1, 3, 8, 30, 100, 128)[case % 6]
        size = [(64, 64), (50, 38), (97, 61), (128, 128), (33, 70)][case % 5]
        blur = O.BLUR_RADIUS * (1.0, 4.0, 0.0)[case % 3]
        cull, clipb = case % 4 != 3, blur > 0 and case % 5 != 4
        nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64, device="cuda") if case % 2 else None  # clip_faces' array for an unclipped scene
        args = (fv, torch.tensor(first), torch.tensor(num), size, blur, K, True, clipb, cull, nbr)
        a = rasterize_meshes(*args, naive=False)
        b = rasterize_meshes(*args, naive=True)
        for x, y, name in zip(a, b, ("pix_to_face", "zbuf", "bary", "dists")):
            assert torch.equal(x, y), (case, name, size, K)
        checked += int((a[0][..., 0] >= 0).sum())
        full += int((a[0][..., K - 1] >= 0).sum())
    assert checked > 2000 and full > 200  # covered pixels; pixels whose K-th slot is taken (lists that filled up)
