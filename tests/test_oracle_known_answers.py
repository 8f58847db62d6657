"""Pins the CPU oracle (oracle/) with closed forms, hand-computed single-triangle cases and fp64 finite
differences -- the reference holds no golden vectors for this path (parity unpinned)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import p3d_restate as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constants():
    assert abs(O.BLUR_RADIUS - 9.21024036697585e-4) < 1e-15  # environment.py:251
    assert abs(float(O.proj_scale()) - 1.0 / math.tan(math.radians(30))) < 1e-6
    assert O.K_SOFT == 100 and O.K_HARD == 1 and O.Z_CLIP == 0.5


def test_look_at_closed_form():
    # C = (0,0,r) => x=(-1,0,0), y=(0,1,0), z=(0,0,-1), T=(0,0,r)
    C = torch.tensor([[0.0, 0.0, 4.0]])
    R = O.look_at_rotation(C)
    assert torch.allclose(R[0], torch.diag(torch.tensor([-1.0, 1.0, -1.0])), atol=1e-7)
    T = O.translation_from(R, C)
    assert torch.allclose(T, torch.tensor([[0.0, 0.0, 4.0]]), atol=1e-6)
    # world origin lands at view-space (0,0,r); R is orthonormal for a generic camera
    C = torch.tensor([[1.0, 2.0, -3.0]])
    R = O.look_at_rotation(C)
    T = O.translation_from(R, C)
    assert torch.allclose(R[0] @ R[0].T, torch.eye(3), atol=1e-6)
    assert torch.allclose(torch.zeros(1, 3) @ R[0] + T, torch.tensor([[0.0, 0.0, C.norm()]]), atol=1e-5)


def test_look_at_view_transform_matches_step_convention_at_zero_elevation():
    # reset()'s and step()'s angle conventions agree only when elevation == 0
    r, az = torch.tensor([4.0]), torch.tensor([0.37])
    R1, T1 = O.look_at_view_transform(r, torch.tensor([0.0]), az)
    C = torch.stack([r * torch.sin(az) * 1.0, r * torch.sin(az) * 0.0, r * torch.cos(az)], 1)
    R2 = O.look_at_rotation(C)
    assert torch.allclose(R1, R2, atol=1e-6) and torch.allclose(T1, O.translation_from(R2, C), atol=1e-6)


def test_look_at_degenerate_up_branch():
    C = torch.tensor([[0.0, 4.0, 0.0]])  # camera on the up axis: x ~ 0 -> replacement branch
    R = O.look_at_rotation(C)
    assert torch.isfinite(R).all()


def _tri(z=2.0):
    # front-facing in PyTorch3D's NDC convention (area > 0): E(v0; v1, v2) > 0
    fv = torch.tensor([[[-0.5, -0.5, z], [0.0, 0.5, z], [0.5, -0.5, z]]])
    x, y = fv[0, :, 0], fv[0, :, 1]
    area = (x[0] - x[1]) * (y[2] - y[1]) - (y[0] - y[1]) * (x[2] - x[1])
    if area < 0:
        fv = fv[:, [0, 2, 1]]
    return fv


def test_single_triangle_inside_outside_and_cull():
    S = 8
    fv = _tri()
    p2f, zbuf, bary, dists = O.rasterize_meshes(fv, S, O.BLUR_RADIUS, 4)
    # pixel centres: xf = -1 + (2*(S-1-xi)+1)/S.  Pixel (yi=4, xi=4) -> (xf, yf) = (-0.125, -0.125): inside
    assert p2f[4, 4, 0] == 0 and p2f[4, 4, 1] == -1
    assert abs(float(zbuf[4, 4, 0]) - 2.0) < 1e-6
    assert float(dists[4, 4, 0]) < 0  # inside => negative squared distance
    b = bary[4, 4, 0]
    assert abs(float(b.sum()) - 1.0) < 1e-6 and (b > 0).all()
    # hand value: distance of (-0.125,-0.125) to the closest edge
    v = fv[0, :, :2].double()
    p = torch.tensor([-0.125, -0.125], dtype=torch.float64)

    def seg(a, bb):
        t = ((bb - a) @ (p - a) / ((bb - a) @ (bb - a))).clamp(0, 1)
        return float(((a + t * (bb - a) - p) ** 2).sum())

    d = min(seg(v[0], v[1]), seg(v[0], v[2]), seg(v[1], v[2]))
    assert abs(float(dists[4, 4, 0]) + d) < 1e-6
    # far corner pixel: outside and beyond the blur radius -> empty
    assert p2f[0, 0, 0] == -1 and float(zbuf[0, 0, 0]) == -1 and float(dists[0, 0, 0]) == -1
    # reversed winding is a back face: culled everywhere
    p2f_b, *_ = O.rasterize_meshes(fv[:, [0, 2, 1]], S, O.BLUR_RADIUS, 4)
    assert (p2f_b == -1).all()


def test_blur_band_and_soft_alpha():
    S = 64
    fv = _tri()
    p2f, zbuf, bary, dists = O.rasterize_meshes(fv, S, O.BLUR_RADIUS, O.K_SOFT)
    outside_hit = (p2f[..., 0] == 0) & (dists[..., 0] > 0)
    assert outside_hit.any()  # pixels within sqrt(blur) of an edge are matched from outside
    assert float(dists[..., 0][outside_hit].max()) < O.BLUR_RADIUS
    img = O.sigmoid_alpha_blend(dists, p2f)
    a = img[..., 3]
    assert float(a.min()) >= 0 and float(a.max()) <= 1 and (img[..., :3] == 1).all()
    # A.6 closed form for one face: alpha = sigmoid(-d/sigma)
    m = p2f[..., 0] == 0
    assert torch.allclose(a[m], torch.sigmoid(-dists[..., 0][m] / O.SIGMA), atol=1e-6)
    assert (a[~m] == 0).all()


def test_topk_keeps_nearest_in_z():
    S = 8
    tris = torch.cat([_tri(z) for z in (3.0, 1.5, 2.5, 2.0)])
    p2f, zbuf, _, _ = O.rasterize_meshes(tris, S, 0.0, 2)
    assert p2f[4, 4].tolist() == [1, 3]  # the two nearest, ascending z
    assert torch.allclose(zbuf[4, 4], torch.tensor([1.5, 2.0]))
    # equal depth: the smaller face index wins ((pz, f) lexicographic order, A.4)
    tris = torch.cat([_tri(2.0), _tri(2.0), _tri(2.0)])
    p2f, *_ = O.rasterize_meshes(tris, S, 0.0, 2)
    assert p2f[4, 4].tolist() == [0, 1]


def test_clip_faces_cases():
    # one vertex behind z = 0.5 -> two triangles that know each other; two behind -> one; all behind -> none
    f_one = torch.tensor([[[-0.2, -0.2, 0.2], [0.0, 0.3, 2.0], [0.3, -0.2, 2.0]]])
    out, c2u, nb, conv, cidx = O.clip_faces(f_one)
    assert out.shape[0] == 2 and nb.tolist() == [1, 0] and c2u.tolist() == [0, 0]
    assert float(out[..., 2].min()) >= 0.5 - 1e-6
    f_two = torch.tensor([[[-0.2, -0.2, 0.2], [0.0, 0.3, 0.3], [0.3, -0.2, 2.0]]])
    out, c2u, nb, conv, cidx = O.clip_faces(f_two)
    assert out.shape[0] == 1 and nb.tolist() == [-1] and abs(float(out[0, :, 2].min()) - 0.5) < 1e-6
    f_all = torch.tensor([[[-0.2, -0.2, 0.2], [0.0, 0.3, 0.3], [0.3, -0.2, 0.1]]])
    out, *_ = O.clip_faces(f_all)
    assert out.shape[0] == 0
    f_none = _tri()
    out, c2u, *_ = O.clip_faces(f_none)
    assert out is f_none and c2u is None
    # barycentric conversion rows are convex combinations of the original corners
    out, c2u, nb, conv, cidx = O.clip_faces(f_one)
    assert torch.allclose(conv.sum(1), torch.ones(conv.shape[0], 3), atol=1e-6)


def test_teapot_stats(teapot):
    v, f = teapot
    assert v.shape == (1292, 3) and f.shape == (2464, 3)
    assert abs(float(v[:, 0].min()) + 0.957) < 2e-3 and abs(float(v[:, 0].max()) - 1.094) < 2e-3
    assert abs(float(v[:, 1].min())) < 1e-6 and abs(float(v[:, 1].max()) - 1.005) < 2e-3
    assert abs(float(v.norm(dim=1).max()) - 1.3495) < 2e-3


def _scene(v, f, x2=0.5):
    return [(v, f), (v + torch.tensor([x2, 0, 1.0]), f), (v + torch.tensor([-x2, 0, 2.0]), f)]


def test_env_step_invariants_and_reward_rule(teapot):
    v, f = teapot
    env = O.OracleEnv(_scene(v, f), 32)
    obs0 = env.reset()
    assert obs0.shape == (1, 4, 32, 32)
    loss0 = float(env.fullReward)
    assert abs(float(env.objectMass) - (loss0 + 1)) < 1e-6
    a = torch.tensor([0.3, -0.2], requires_grad=True)
    obs, r, done, info = env.step(a)
    assert obs.shape == (1, 4, 32, 32) and info["full_state"].shape == (1, 32, 32, 4)
    bg = obs[0, 3] == -1
    assert bg.any() and (obs[0, :3][:, bg] == 1).all()  # background: white, depth -1
    assert float(obs[0, :3].max()) <= 1.0 + 1e-6 and float(obs[0, 3][~bg].min()) > 0.5
    assert (info["full_state"][..., :3] == 3).all()  # i1*i2 + i2*i3 + i1*i3 with RGB == 1
    loss1 = float(info["full_reward"])
    expect = (loss0 - loss1) / (loss0 + 1) + (5 if loss1 < 0.1 else -0.2)
    assert abs(float(r) - expect) < 1e-5 and bool(done) == (loss1 < 0.1)
    # el/az update: 0.05 * normalised action (environment.py:356-361)
    n = a.detach() / a.detach().norm()
    assert abs(float(env.elevation) - 0.05 * float(n[0])) < 1e-7 and abs(float(env.azimuth) - 0.05 * float(n[1])) < 1e-7


def test_zero_action_branch(teapot):
    v, f = teapot
    env = O.OracleEnv(_scene(v, f), 16)
    env.reset(azimuth=0.2)
    a = torch.zeros(2, requires_grad=True)  # demo.py:80: nn.Parameter(zeros(2)) -> un-normalised pass-through
    _, r, _, _ = env.step(a)
    r.backward()
    assert torch.isfinite(a.grad).all() and abs(float(env.azimuth) - 0.2) < 1e-7


@pytest.mark.parametrize("az0", [0.0, 0.3])
def test_analytic_backward_matches_fp64_finite_differences(teapot, az0):
    v, f = teapot

    def run(a):
        env = O.OracleEnv(_scene(v.double(), f), 24, dtype=torch.float64)
        env.reset(azimuth=az0)
        return env.step(a)[1]

    a = torch.tensor([0.3, -0.2], dtype=torch.float64, requires_grad=True)
    run(a).backward()
    eps = 1e-6
    for i in range(2):
        ap, am = a.detach().clone(), a.detach().clone()
        ap[i] += eps
        am[i] -= eps
        fd = float((run(ap) - run(am)) / (2 * eps))
        assert abs(fd - float(a.grad[i])) < 1e-4 * max(1.0, abs(fd)), (i, fd, float(a.grad[i]))


def test_clipped_scene_renders(teapot):
    # camera 1.2 from the origin: the near teapot straddles z = 0.5 -> clip_faces cases 3 and 4 are exercised
    v, f = teapot
    env = O.OracleEnv(_scene(v, f, 0.3), 24)
    obs = env.reset(radius=1.2, azimuth=0.1)
    assert torch.isfinite(obs).all() and float(env.fullReward) >= 0
    R, T = O.look_at_view_transform(torch.tensor([1.2]), torch.tensor([0.0]), torch.tensor([0.1]))
    ndc = O.world_to_ndc(env.scene[0], R[0], T[0])
    behind = (ndc[env.scene[1]][:, :, 2] < 0.5).sum(1)
    assert (behind == 1).any() and (behind == 2).any()
