"""oracle/p3d_restate.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement (torch-CPU tensors + the C rasteriser in raster_naive.c) of the arithmetic
``OcclusionEnv.reset/step/render`` reaches through PyTorch3D.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product path (``occlusionenv_amd``) never does.

PARITY UNPINNED: the reference holds no tests or golden vectors for this path and PyTorch3D
(pytorch3d==0.6.2, /root/reference/requirements.txt:50) is neither vendored in the reference nor
installed here, so every function below restates the *published* upstream algorithm as recorded in
SURVEY.md Appendix A, and is pinned only by this repo's closed-form and finite-difference tests.

Each function cites the reference call site it serves (file:line in /root/reference) and the
SURVEY appendix section it follows.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))

# ---- constants (environment.py:219,242,249-255,267-275,286) -----------------
SIGMA = 1e-4
BLUR_RADIUS = float(np.log(1.0 / 1e-4 - 1.0) * SIGMA)  # environment.py:251
K_SOFT = 100
K_HARD = 1
Z_CLIP = 0.5  # znear / 2
STEP_SIZE = 0.05
LIGHT_LOCATION = (2.0, 2.0, -2.0)
AMBIENT, DIFFUSE, SPECULAR, SHININESS = 0.5, 0.3, 0.2, 64.0


def proj_scale(dtype=torch.float32) -> torch.Tensor:
    """FoVPerspectiveCameras() defaults: K[0][0] = 2*znear/(max_x-min_x) with fov=60deg (A.2)."""
    fov = torch.tensor(60.0, dtype=dtype) * (math.pi / 180.0)
    max_y = torch.tan(fov / 2) * 1.0
    return 2.0 * 1.0 / (max_y - (-max_y))


_LIB = None


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        so = os.environ.get("ORC_LIB") or os.path.join(_HERE, "liborc.so")  # ORC_LIB: the sanitizer build (tests)
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", _HERE] + (["asan"] if so.endswith("_asan.so") else []))
        _LIB = ctypes.CDLL(so)
    return _LIB


def _ptr(t: torch.Tensor):
    return ctypes.c_void_p(t.data_ptr())


# ---- camera (A.1) ---------------------------------------------------------------------------
def look_at_rotation(C: torch.Tensor) -> torch.Tensor:
    """environment.py:334,367 -> pytorch3d look_at_rotation(at=0, up=+Y).  C: (N,3) -> (N,3,3)."""
    at = torch.zeros_like(C)
    up = torch.zeros_like(C)
    up[:, 1] = 1.0
    z_axis = F.normalize(at - C, eps=1e-5)
    x_axis = F.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    is_close = torch.isclose(x_axis, torch.tensor(0.0, dtype=C.dtype), atol=5e-3).all(dim=1, keepdim=True)
    if is_close.any():
        replacement = F.normalize(torch.cross(y_axis, z_axis, dim=1), eps=1e-5)
        x_axis = torch.where(is_close, replacement, x_axis)
    R = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1)
    return R.transpose(1, 2)


def translation_from(R: torch.Tensor, C: torch.Tensor) -> torch.Tensor:
    """environment.py:335,368:  T = -bmm(R^T, C)."""
    return -torch.bmm(R.transpose(1, 2), C[:, :, None])[:, :, 0]


def look_at_view_transform(dist, elev, azim):
    """environment.py:308 (degrees=False).  Inputs are (N,) tensors."""
    x = dist * torch.cos(elev) * torch.sin(azim)
    y = dist * torch.sin(elev)
    z = dist * torch.cos(elev) * torch.cos(azim)
    C = torch.stack([x, y, z], dim=1).view(-1, 3)
    R = look_at_rotation(C)
    return R, translation_from(R, C)


def world_to_ndc(verts_world: torch.Tensor, R: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    """MeshRasterizer.transform with FoVPerspectiveCameras defaults (A.2).
    verts (V,3), R (3,3), T (3,) -> (V,3) = (x_ndc, y_ndc, z_view)."""
    view = verts_world @ R + T
    s = proj_scale(verts_world.dtype)
    z = view[:, 2]
    return torch.stack([view[:, 0] * s / z, view[:, 1] * s / z, z], dim=1)


# ---- z clipping (A.3) -----------------------------------------------------------------------
def _find_intersections(fv, p1_ind, clip_value, perspective_correct):
    T_ = fv.shape[0]
    p2_ind = torch.remainder(p1_ind + 1, 3)
    p3_ind = torch.remainder(p1_ind + 2, 3)
    ar = torch.arange(T_)
    p1, p2, p3 = fv[ar, p1_ind], fv[ar, p2_ind], fv[ar, p3_ind]

    def cut(pa, pb):
        w = ((pa[:, 2] - clip_value) / (pa[:, 2] - pb[:, 2])).detach()
        pc = pa * (1 - w[:, None]) + pb * w[:, None]
        if perspective_correct:
            pa_w = pa[:, :2] * pa[:, 2:3]
            pb_w = pb[:, :2] * pb[:, 2:3]
            xy = (pa_w * (1 - w[:, None]) + pb_w * w[:, None]) / clip_value
            pc = torch.cat([xy, pc[:, 2:3]], dim=1)
        return pc, w

    p4, w2 = cut(p1, p2)
    p5, w3 = cut(p1, p3)
    pb = [torch.zeros((T_, 3), dtype=fv.dtype) for _ in range(5)]
    pb[0][ar, p1_ind] = 1
    pb[1][ar, p2_ind] = 1
    pb[2][ar, p3_ind] = 1
    pb[3][ar, p1_ind] = 1 - w2
    pb[3][ar, p2_ind] = w2
    pb[4][ar, p1_ind] = 1 - w3
    pb[4][ar, p3_ind] = w3
    return (p1, p2, p3, p4, p5), pb


def clip_faces(fv: torch.Tensor, z_clip=Z_CLIP, perspective_correct=True):
    """pytorch3d clip_faces with cull_to_frustum=False (A.3).
    Returns (face_verts_clipped, clipped_to_unclipped, neighbor, bary_conversion, conv_idx);
    the last four are None when nothing is clipped."""
    behind = fv[:, :, 2] < z_clip
    nbehind = behind.sum(1)
    if int(nbehind.sum().item()) == 0:
        return fv, None, None, None, None
    case2 = nbehind == 3
    case3 = nbehind == 2
    case4 = nbehind == 1
    case1 = nbehind == 0
    delta = 1 + case4.int() - case2.int()
    u2c = (delta.cumsum(0) - delta).long()
    Fc = int(delta.sum().item())
    i1 = case1.nonzero(as_tuple=True)[0]
    i3 = case3.nonzero(as_tuple=True)[0]
    i4 = case4.nonzero(as_tuple=True)[0]
    pieces = []  # (dest index tensor, (n,3,3) verts)
    c2u = torch.full((Fc,), -1, dtype=torch.int64)
    nb = torch.full((Fc,), -1, dtype=torch.int64)
    conv_idx = torch.full((Fc,), -1, dtype=torch.int64)
    pieces.append((u2c[i1], fv[i1]))
    c2u[u2c[i1]] = i1
    convs = []
    n3 = i3.numel()
    if n3:
        p1_ind = torch.where(~behind[i3])[1]
        (p1, _, _, p4, p5), pb = _find_intersections(fv[i3], p1_ind, z_clip, perspective_correct)
        pieces.append((u2c[i3], torch.stack((p4, p5, p1), 1)))
        c2u[u2c[i3]] = i3
        conv_idx[u2c[i3]] = torch.arange(n3)
        convs.append(torch.stack((pb[3], pb[4], pb[0]), 2))
    n4 = i4.numel()
    if n4:
        p1_ind = torch.where(behind[i4])[1]
        (_, p2, p3, p4, p5), pb = _find_intersections(fv[i4], p1_ind, z_clip, perspective_correct)
        c = u2c[i4]
        pieces.append((c, torch.stack((p4, p2, p5), 1)))
        pieces.append((c + 1, torch.stack((p5, p2, p3), 1)))
        c2u[c] = i4
        c2u[c + 1] = i4
        nb[c] = c + 1
        nb[c + 1] = c
        conv_idx[c] = n3 + torch.arange(n4)
        conv_idx[c + 1] = n3 + n4 + torch.arange(n4)
        convs.append(torch.stack((pb[3], pb[1], pb[4]), 2))
        convs.append(torch.stack((pb[4], pb[1], pb[2]), 2))
    dest = torch.cat([p[0] for p in pieces])
    vals = torch.cat([p[1] for p in pieces])
    order = torch.argsort(dest)
    out = vals[order]  # differentiable gather; dest is a permutation of 0..Fc-1
    bary_conv = torch.cat(convs) if convs else None
    return out, c2u, nb, bary_conv, conv_idx


# ---- rasteriser (A.4, A.5) ------------------------------------------------------------------
class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, face_verts, neighbor, S, blur, K, persp, clipb, cull):
        fv = face_verts.detach().contiguous()
        Fn = fv.shape[0]
        dt = fv.dtype
        suffix = "f32" if dt == torch.float32 else "f64"
        creal = ctypes.c_float if dt == torch.float32 else ctypes.c_double
        p2f = torch.empty((S, S, K), dtype=torch.int64)
        zbuf = torch.empty((S, S, K), dtype=dt)
        bary = torch.empty((S, S, K, 3), dtype=dt)
        dists = torch.empty((S, S, K), dtype=dt)
        nbp = _ptr(neighbor.contiguous()) if neighbor is not None else None
        fn = getattr(lib(), f"orc_rasterize_naive_{suffix}")
        rc = fn(_ptr(fv), nbp, ctypes.c_int64(Fn), S, S, creal(blur), K, int(persp), int(clipb), int(cull),
                _ptr(p2f), _ptr(zbuf), _ptr(bary), _ptr(dists))
        assert rc == 0
        ctx.save_for_backward(fv, p2f)
        ctx.cfg = (S, K, persp, clipb, suffix)
        ctx.mark_non_differentiable(p2f, zbuf, bary)
        ctx.set_materialize_grads(False)
        return p2f, zbuf, bary, dists

    @staticmethod
    def backward(ctx, g_p2f, g_z, g_bary, g_dists):
        fv, p2f = ctx.saved_tensors
        S, K, persp, clipb, suffix = ctx.cfg
        if g_dists is None:
            return (torch.zeros_like(fv),) + (None,) * 7
        gfv = torch.empty_like(fv)
        fn = getattr(lib(), f"orc_rasterize_backward_dists_{suffix}")
        rc = fn(_ptr(fv), _ptr(p2f), _ptr(g_dists.contiguous()), ctypes.c_int64(fv.shape[0]), S, S, K,
                int(persp), int(clipb), _ptr(gfv))
        assert rc == 0
        return (gfv,) + (None,) * 7


def rasterize_meshes(face_verts, S, blur_radius, K, cull_backfaces=True, z_clip=Z_CLIP):
    """pytorch3d rasterize_meshes for one mesh, perspective camera, bin_size=0 (naive).
    face_verts (F,3,3) in (x_ndc,y_ndc,z_view).  Returns pix_to_face (orig ids), zbuf, bary (orig), dists."""
    persp = True
    clipb = blur_radius > 0.0
    fvc, c2u, nb, bary_conv, conv_idx = clip_faces(face_verts, z_clip, persp)
    if fvc.shape[0] == 0:
        dt = face_verts.dtype
        return (torch.full((S, S, K), -1, dtype=torch.int64), torch.full((S, S, K), -1.0, dtype=dt),
                torch.full((S, S, K, 3), -1.0, dtype=dt), torch.full((S, S, K), -1.0, dtype=dt) + 0 * face_verts.sum())
    p2f, zbuf, bary, dists = _Rasterize.apply(fvc, nb, S, float(blur_radius), K, persp, clipb, cull_backfaces)
    if c2u is not None:
        valid = p2f != -1
        safe = p2f.clamp(min=0)
        p2f_u = torch.where(valid, c2u[safe], torch.full_like(p2f, -1))
        if bary_conv is not None:
            cidx = torch.where(valid, conv_idx[safe], torch.full_like(p2f, -1))
            m = cidx != -1
            if m.any():
                sub = bary_conv[cidx[m]].bmm(bary[m].unsqueeze(-1)).squeeze(-1)
                sub = sub / sub.sum(dim=1, keepdim=True)
                bary = bary.clone()
                bary[m] = sub
        p2f = p2f_u
    return p2f, zbuf, bary, dists


def pixel_candidates(face_verts, S, yi, xi, blur_radius, band=1e-3, cull_backfaces=True, max_out=4096, area_band=2e-9,
                     vert_band=0.0):
    """Tie classifier support (tests/parity_utils.py): what the naive rasteriser computes at ONE pixel for every face
    that is a candidate there or misses by a hair (raster_naive.c: orc_pixel_candidates).  face_verts (F,3,3) AFTER
    clip_faces.  Returns dict of numpy arrays f, z, dist, minb, flags (1 inside, 2 candidate, 4 pz < 0, 8 = face area
    within area_band + vert_band * perimeter of the kEpsilon visibility threshold)."""
    fv = face_verts.detach().to(torch.float32).contiguous()
    of = torch.empty(max_out, dtype=torch.int64)
    oz, od, ob = (torch.empty(max_out, dtype=torch.float32) for _ in range(3))
    ofl = torch.empty(max_out, dtype=torch.int32)
    fn = lib().orc_pixel_candidates_f32
    fn.restype = ctypes.c_int
    n = fn(_ptr(fv), ctypes.c_int64(fv.shape[0]), S, S, int(yi), int(xi), ctypes.c_float(blur_radius), 1,
           int(blur_radius > 0.0), int(cull_backfaces), ctypes.c_float(band), ctypes.c_float(area_band),
           ctypes.c_float(vert_band), _ptr(of), _ptr(oz), _ptr(od), _ptr(ob), _ptr(ofl), max_out)
    assert n >= 0
    return dict(f=of[:n].numpy(), z=oz[:n].numpy(), dist=od[:n].numpy(), minb=ob[:n].numpy(), flags=ofl[:n].numpy())


# ---- shaders (A.6, A.7) ---------------------------------------------------------------------
def sigmoid_alpha_blend(dists, pix_to_face, sigma=SIGMA):
    """SoftSilhouetteShader (environment.py:263): (S,S,K) -> (S,S,4), RGB = 1."""
    mask = (pix_to_face >= 0).to(dists.dtype)
    prob = torch.sigmoid(-dists / sigma) * mask
    alpha = 1.0 - torch.prod(1.0 - prob, dim=-1)
    ones = torch.ones(dists.shape[:2] + (3,), dtype=dists.dtype)
    return torch.cat([ones, alpha[..., None]], dim=-1)


def soft_silhouette(verts_world, faces, R, T, S, K=K_SOFT):
    """silhouette_renderer(meshes_world=mesh, R=R, T=T) (environment.py:316-318,370-372).  K = faces_per_pixel: the
    reference's 100 (environment.py:79-83); the engine takes it as a parameter and the tests vary it."""
    ndc = world_to_ndc(verts_world, R, T)
    p2f, _, _, dists = rasterize_meshes(ndc[faces], S, BLUR_RADIUS, K)
    return sigmoid_alpha_blend(dists, p2f)


def sample_atlas(atlas, p2f, bary):
    """[P3D] TexturesAtlas.sample_textures (environment.py:127,152,175 wrap ShapeNet's per-face (R,R,3) atlases):
    the barycentric pair (w0, w1) picks a texel of the face's R x R grid; the upper triangle of each cell is
    mirrored onto the lower one.  atlas (F,R,R,3), p2f (...,K), bary (...,K,3) -> texels (...,K,3)."""
    Rr = atlas.shape[1]
    mask = (p2f < 0)[..., None]
    w01 = torch.where(mask, torch.zeros_like(bary[..., :2]), bary[..., :2])
    w_xy = (w01 * Rr).to(torch.int64).clamp(max=Rr - 1)
    below_diag = (w01.sum(dim=-1) * Rr - w_xy.to(bary.dtype).sum(dim=-1)) <= 1.0
    w_x, w_y = w_xy.unbind(-1)
    w_x = torch.where(below_diag, w_x, Rr - 1 - w_x)
    w_y = torch.where(below_diag, w_y, Rr - 1 - w_y)
    texels = atlas[p2f.clamp(min=0), w_y, w_x]
    return texels * (p2f >= 0)[..., None].to(bary.dtype)


def hard_flat_rgbd(verts_world, faces, R, T, S, verts_rgb=None, atlas=None):
    """phong_renderer(meshes_world=mesh, R=R, T=T) with HardFlatShader (environment.py:310,336,375;
    wrapper :42-51).  Returns (image (S,S,4), zbuf (S,S,1)).  White TexturesVertex unless verts_rgb."""
    dt = verts_world.dtype
    ndc = world_to_ndc(verts_world, R, T)
    p2f, zbuf, bary, _ = rasterize_meshes(ndc[faces], S, 0.0, K_HARD)
    fverts = verts_world[faces]  # (F,3,3)
    n = torch.cross(fverts[:, 1] - fverts[:, 0], fverts[:, 2] - fverts[:, 0], dim=1)
    nn = n.norm(dim=1, keepdim=True).clamp(min=1e-6)
    fnormals = n / nn
    fcenters = fverts.mean(dim=-2)
    mask = p2f == -1
    idx = p2f.clamp(min=0)
    pcoords = fcenters[idx]  # (S,S,1,3)
    pnormals = fnormals[idx]
    pcoords = torch.where(mask[..., None], torch.zeros_like(pcoords), pcoords)
    pnormals = torch.where(mask[..., None], torch.zeros_like(pnormals), pnormals)
    rgb = torch.ones_like(verts_world) if verts_rgb is None else verts_rgb
    frgb = rgb[faces]  # (F,3,3)
    texels = (bary[..., None] * frgb[idx]).sum(dim=-2)  # (S,S,1,3)
    texels = torch.where(mask[..., None], torch.zeros_like(texels), texels)
    if atlas is not None:
        texels = sample_atlas(atlas.to(dt), p2f, bary)
    # lighting (PointLights / Materials defaults, A.7)
    L = torch.tensor(LIGHT_LOCATION, dtype=dt)
    # camera centre as get_camera_center() returns it: the translation row of inv([R|T])
    M = torch.eye(4, dtype=dt)
    M[:3, :3] = R
    M[3, :3] = T
    C = torch.linalg.inv(M)[3, :3]
    nrm = F.normalize(pnormals, p=2, dim=-1, eps=1e-6)
    direction = F.normalize(L - pcoords, p=2, dim=-1, eps=1e-6)
    cos_angle = torch.sum(nrm * direction, dim=-1)
    diffuse = DIFFUSE * F.relu(cos_angle)[..., None]
    smask = (cos_angle > 0).to(dt)
    view_dir = F.normalize(C - pcoords, p=2, dim=-1, eps=1e-6)
    reflect = -direction + 2 * (cos_angle[..., None] * nrm)
    alpha = F.relu(torch.sum(view_dir * reflect, dim=-1)) * smask
    specular = SPECULAR * torch.pow(alpha, SHININESS)[..., None]
    colors = (AMBIENT + diffuse) * texels + specular  # (S,S,1,3)
    is_bg = p2f[..., 0] < 0
    pix = torch.where(is_bg[..., None], torch.ones_like(colors[..., 0, :]), colors[..., 0, :])
    a = (~is_bg).to(dt)[..., None]
    return torch.cat([pix, a], dim=-1), zbuf


def vertex_normals(verts, faces):
    """[P3D] Meshes.verts_normals_packed(): every face adds its (area-weighted) normal to its three corners, the sums
    are normalised with eps 1e-6."""
    vf = verts[faces]
    n = torch.zeros_like(verts)
    n = n.index_add(0, faces[:, 1], torch.cross(vf[:, 2] - vf[:, 1], vf[:, 0] - vf[:, 1], dim=1))
    n = n.index_add(0, faces[:, 2], torch.cross(vf[:, 0] - vf[:, 2], vf[:, 1] - vf[:, 2], dim=1))
    n = n.index_add(0, faces[:, 0], torch.cross(vf[:, 1] - vf[:, 0], vf[:, 2] - vf[:, 0], dim=1))
    return F.normalize(n, eps=1e-6, dim=1)


def phong_rgbd(verts_world, faces, R, T, S, soft=False, atlas=None, vnormals=None):
    """phong_renderer with the two shaders the reference keeps commented out next to HardFlatShader
    (environment.py:281-282): HardPhongShader (soft=False) / SoftPhongShader (soft=True), both with the renderer's
    K = 1, blur 0 rasterisation (environment.py:267-273) and default BlendParams (sigma = gamma = 1e-4, white
    background), znear 1 / zfar 100 from FoVPerspectiveCameras.  [P3D] phong_shading: pixel position and normal are
    barycentric interpolations of the face's vertex positions / vertex normals; lighting as in A.7;
    softmax_rgb_blend as published.  Returns (image (S,S,4), zbuf (S,S,1))."""
    dt = verts_world.dtype
    ndc = world_to_ndc(verts_world, R, T)
    p2f, zbuf, bary, dists = rasterize_meshes(ndc[faces], S, 0.0, K_HARD)
    vn = vertex_normals(verts_world, faces) if vnormals is None else vnormals
    mask = p2f == -1
    idx = p2f.clamp(min=0)
    fverts, fnorm = verts_world[faces], vn[faces]
    pcoords = (bary[..., None] * fverts[idx]).sum(dim=-2)
    pnormals = (bary[..., None] * fnorm[idx]).sum(dim=-2)
    pcoords = torch.where(mask[..., None], torch.zeros_like(pcoords), pcoords)
    pnormals = torch.where(mask[..., None], torch.zeros_like(pnormals), pnormals)
    texels = torch.where(mask[..., None], torch.zeros_like(pcoords), bary.sum(-1, keepdim=True).expand_as(pcoords))  # white verts
    if atlas is not None:
        texels = sample_atlas(atlas.to(dt), p2f, bary)
    L = torch.tensor(LIGHT_LOCATION, dtype=dt)
    M = torch.eye(4, dtype=dt)
    M[:3, :3] = R
    M[3, :3] = T
    C = torch.linalg.inv(M)[3, :3]
    nrm = F.normalize(pnormals, p=2, dim=-1, eps=1e-6)
    direction = F.normalize(L - pcoords, p=2, dim=-1, eps=1e-6)
    cos_angle = torch.sum(nrm * direction, dim=-1)
    diffuse = DIFFUSE * F.relu(cos_angle)[..., None]
    smask = (cos_angle > 0).to(dt)
    view_dir = F.normalize(C - pcoords, p=2, dim=-1, eps=1e-6)
    reflect = -direction + 2 * (cos_angle[..., None] * nrm)
    alpha = F.relu(torch.sum(view_dir * reflect, dim=-1)) * smask
    specular = SPECULAR * torch.pow(alpha, SHININESS)[..., None]
    colors = (AMBIENT + diffuse) * texels + specular  # (S,S,1,3)
    if not soft:  # hard_rgb_blend
        is_bg = p2f[..., 0] < 0
        pix = torch.where(is_bg[..., None], torch.ones_like(colors[..., 0, :]), colors[..., 0, :])
        return torch.cat([pix, (~is_bg).to(dt)[..., None]], dim=-1), zbuf
    # softmax_rgb_blend(colors, fragments, BlendParams(), znear=1, zfar=100)
    sigma, gamma, znear, zfar, eps = SIGMA, 1e-4, 1.0, 100.0, 1e-10
    m = (p2f >= 0).to(dt)
    prob = torch.sigmoid(-dists / sigma) * m
    alpha_px = torch.prod(1.0 - prob, dim=-1)
    z_inv = (zfar - zbuf) / (zfar - znear) * m
    z_inv_max = torch.max(z_inv, dim=-1).values[..., None].clamp(min=eps)
    weights_num = prob * torch.exp((z_inv - z_inv_max) / gamma)
    delta = torch.exp((eps - z_inv_max) / gamma).clamp(min=eps)
    denom = weights_num.sum(dim=-1)[..., None] + delta
    wcol = (weights_num[..., None] * colors).sum(dim=-2)
    rgb = (wcol + delta * torch.ones(3, dtype=dt)) / denom
    return torch.cat([rgb, (1.0 - alpha_px)[..., None]], dim=-1), zbuf


# ---- environment (environment.py:286-402) ---------------------------------------------------
class OracleEnv:
    """Restatement of OcclusionEnv.reset/step/render for ONE env on the CPU.

    ``objects`` is a list of three (verts (V,3), faces (F,3) int64) in WORLD space (offsets already
    applied, environment.py:148,171); the joined scene is their concatenation
    (join_meshes_as_scene, environment.py:191).
    """

    def __init__(self, objects, img_size, dtype=torch.float32, atlases=None):
        self.dtype = dtype
        self.S = img_size
        self.objs = [(v.to(dtype), f.long()) for v, f in objects]
        # per-object (F,R,R,3) texture atlases (TexturesAtlas) or None = white TexturesVertex for the whole scene
        self.atlas = None if atlases is None else torch.cat([a.to(dtype) for a in atlases])
        vs, fs, off = [], [], 0
        for v, f in self.objs:
            vs.append(v)
            fs.append(f + off)
            off += v.shape[0]
        self.scene = (torch.cat(vs), torch.cat(fs))
        # optional (S,S) weight of every pixel's term of the loss (tests: 0 on pixels classified as exact ties, so
        # that loss / reward / gradient are compared over the remaining pixels; the product's OccScene.pix_weight)
        self.pixel_weight = None
        # shader of the observation renderer: "flat" = HardFlatShader (environment.py:283, what the reference runs);
        # "hard_phong" / "soft_phong" = the alternatives it keeps commented out (environment.py:281-282)
        self.shader = "flat"
        # environment.py:208: False = objectMass is the initial loss + 1; True = sum((image1 + image2 + image3).alpha ** 2) + 1
        self.normWithObjectSize = False

    def _loss(self):
        sq = self.image[..., 3] ** 2
        return torch.sum(sq if self.pixel_weight is None else sq * self.pixel_weight.to(sq.dtype))

    def _observe(self, R, T):
        if self.shader == "flat":
            return hard_flat_rgbd(self.scene[0], self.scene[1], R[0], T[0], self.S, atlas=self.atlas)
        return phong_rgbd(self.scene[0], self.scene[1], R[0], T[0], self.S, soft=(self.shader == "soft_phong"), atlas=self.atlas)

    def _render_all(self, R, T):
        S = self.S
        obs_img, depth = self._observe(R, T)
        observation = obs_img[None].permute(0, 3, 1, 2).clone()
        observation[:, 3] = depth[None].permute(0, 3, 1, 2)[:, 0]
        imgs = [soft_silhouette(v, f, R[0], T[0], S, getattr(self, "faces_per_pixel", K_SOFT))[None] for v, f in self.objs]
        image = imgs[0] * imgs[1] + imgs[1] * imgs[2] + imgs[0] * imgs[2]
        return observation, image, imgs

    def reset(self, radius=4.0, azimuth=0.0, elevation=0.0):
        dt = self.dtype
        self.camera_position = torch.zeros(3, dtype=dt)
        self.radius = torch.tensor([radius], dtype=dt)
        self.elevation = torch.tensor([elevation], dtype=dt)
        self.azimuth = torch.tensor([azimuth], dtype=dt)
        R, T = look_at_view_transform(self.radius, self.elevation, self.azimuth)
        self.R, self.T = R, T
        observation, self.image, self.alphas = self._render_all(R, T)
        loss = self._loss()
        self.fullReward = loss.detach()
        # environment.py:320,324
        objects = self.alphas[0] + self.alphas[1] + self.alphas[2]
        self.objectMass = (torch.sum(objects[..., 3] ** 2).detach() + 1) if self.normWithObjectSize else (loss.detach() + 1)
        return observation

    def step(self, action):
        for t in (self.elevation, self.azimuth, self.radius, self.camera_position):
            t.detach_()
        action_norm = torch.norm(action)
        normalized_action = action / action_norm if action_norm else action
        self.elevation += normalized_action[0] * STEP_SIZE
        self.azimuth += normalized_action[1] * STEP_SIZE
        self.camera_position[0] = self.radius * torch.sin(self.azimuth) * torch.cos(self.elevation)
        self.camera_position[1] = self.radius * torch.sin(self.azimuth) * torch.sin(self.elevation)
        self.camera_position[2] = self.radius * torch.cos(self.azimuth)
        R = look_at_rotation(self.camera_position[None, :])
        T = translation_from(R, self.camera_position[None, :])
        self.R, self.T = R, T
        observation, self.image, self.alphas = self._render_all(R, T)
        loss = self._loss()
        reward = self.fullReward - loss
        self.fullReward = loss.detach()
        finished = self.fullReward < 0.1
        reward = reward / self.objectMass
        reward = reward + 5 if finished else reward - 0.2
        info = {"full_state": self.image, "position": self.camera_position, "full_reward": self.fullReward}
        return observation, reward, finished, info

    def render(self):
        R = look_at_rotation(self.camera_position[None, :])
        T = translation_from(R, self.camera_position[None, :])
        img, depth = self._observe(R, T)
        return img[None], depth[None]
