"""Operator-level drop-in for ``pytorch3d.renderer.mesh.rasterize_meshes`` as the reference reaches it through
``MeshRasterizer`` (/root/reference/environment.py:258-262, :276-280): the naive
(``bin_size=0``) rasteriser with K-buffer outputs in PyTorch3D's layout, differentiable w.r.t. ``face_verts``
through ``dists``, ``zbuf`` and ``bary_coords``.  ``OcclusionEnv.step`` does NOT go through here (it uses the fused ``occ_render``); this is for
callers of the rasteriser itself and for parity tests at that boundary.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _native as nat


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _RasterizeFaceVerts(torch.autograd.Function):
    @staticmethod
    def forward(ctx, face_verts, first_idx, num_faces, neighbor, H, W, blur_radius, K, persp, clipb, cull, naive=False):
        lib = nat.load()
        if not face_verts.is_cuda:
            raise nat.NativeError("rasterize_meshes needs CUDA/ROCm tensors; there is no CPU fallback")
        fv = face_verts.detach().contiguous().float()
        dev = fv.device
        N = int(first_idx.numel())
        first_idx = first_idx.to(dev, torch.int64).contiguous()
        num_faces = num_faces.to(dev, torch.int64).contiguous()
        nb = None if neighbor is None else neighbor.to(dev, torch.int64).contiguous()
        p2f = torch.empty(N, H, W, K, dtype=torch.int64, device=dev)
        zbuf = torch.empty(N, H, W, K, dtype=torch.float32, device=dev)
        bary = torch.empty(N, H, W, K, 3, dtype=torch.float32, device=dev)
        dists = torch.empty(N, H, W, K, dtype=torch.float32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        fn = lib.occ_rasterize_meshes_naive if naive else lib.occ_rasterize_meshes_tiled
        nat.check(fn(_p(fv), _p(first_idx), _p(num_faces), _p(nb), N, H, W, float(blur_radius), K, int(persp), int(clipb),
                     int(cull), _p(p2f), _p(zbuf), _p(bary), _p(dists), st), "occ_rasterize_meshes")
        ctx.save_for_backward(fv, p2f)
        ctx.cfg = (N, H, W, K, int(persp), int(clipb))
        ctx.mark_non_differentiable(p2f)
        ctx.set_materialize_grads(False)
        return p2f, zbuf, bary, dists

    @staticmethod
    def backward(ctx, g_p2f, g_z, g_bary, g_dists):
        fv, p2f = ctx.saved_tensors
        N, H, W, K, persp, clipb = ctx.cfg
        if g_dists is None and g_z is None and g_bary is None:
            return (torch.zeros_like(fv),) + (None,) * 11
        lib = nat.load()
        gfv = torch.empty_like(fv)
        st = C.c_void_p(torch.cuda.current_stream(fv.device).cuda_stream)

        def c32(g):
            return None if g is None else g.contiguous().float()

        gz, gb, gd = c32(g_z), c32(g_bary), c32(g_dists)
        nat.check(lib.occ_rasterize_meshes_backward(_p(fv), _p(p2f), _p(gz), _p(gb), _p(gd), fv.shape[0], N, H, W, K, persp,
                                                    clipb, _p(gfv), st), "occ_rasterize_meshes_backward")
        return (gfv,) + (None,) * 11


def rasterize_meshes(face_verts: torch.Tensor, mesh_to_face_first_idx: torch.Tensor, num_faces_per_mesh: torch.Tensor,
                     image_size: int = 256, blur_radius: float = 0.0, faces_per_pixel: int = 8,
                     perspective_correct: bool = False, clip_barycentric_coords: bool = False,
                     cull_backfaces: bool = False, clipped_faces_neighbor_idx: Optional[torch.Tensor] = None,
                     naive: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """``(pix_to_face, zbuf, bary_coords, dists)`` like PyTorch3D's ``_C.rasterize_meshes`` with ``bin_size=0``.
    ``face_verts`` (F,3,3) packed (x_ndc, y_ndc, z_view) of all meshes (already z-clipped, see
    ``clipped_faces_neighbor_idx``); ``image_size`` int or (H, W).  ``naive=True`` runs the one-thread-per-pixel kernel
    over all faces instead of the tiled one (``occ_rasterize_meshes_tiled``); the outputs are bit-identical."""
    H, W = (image_size, image_size) if isinstance(image_size, int) else image_size
    return _RasterizeFaceVerts.apply(face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx,
                                     int(H), int(W), blur_radius, int(faces_per_pixel), perspective_correct,
                                     clip_barycentric_coords, cull_backfaces, bool(naive))


class _SigmoidAlphaBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dists, pix_to_face, sigma):
        lib = nat.load()
        if not dists.is_cuda:
            raise nat.NativeError("sigmoid_alpha_blend needs CUDA/ROCm tensors; there is no CPU fallback")
        d = dists.detach().contiguous().float()
        p2f = pix_to_face.to(d.device, torch.int64).contiguous()
        K = d.shape[-1]
        n_pix = d.numel() // K
        images = torch.empty(d.shape[:-1] + (4,), dtype=torch.float32, device=d.device)
        st = C.c_void_p(torch.cuda.current_stream(d.device).cuda_stream)
        nat.check(lib.occ_sigmoid_alpha_blend_fwd(_p(d), _p(p2f), n_pix, K, float(sigma), _p(images), st),
                  "occ_sigmoid_alpha_blend_fwd")
        ctx.save_for_backward(d, p2f)
        ctx.sigma = float(sigma)
        return images

    @staticmethod
    def backward(ctx, g_images):
        d, p2f = ctx.saved_tensors
        lib = nat.load()
        K = d.shape[-1]
        gd = torch.empty_like(d)
        st = C.c_void_p(torch.cuda.current_stream(d.device).cuda_stream)
        nat.check(lib.occ_sigmoid_alpha_blend_bwd(_p(d), _p(p2f), _p(g_images.contiguous().float()), d.numel() // K, K,
                                                  ctx.sigma, _p(gd), st), "occ_sigmoid_alpha_blend_bwd")
        return gd, None, None


def sigmoid_alpha_blend(dists: torch.Tensor, pix_to_face: torch.Tensor, sigma: float = 1e-4) -> torch.Tensor:
    """PyTorch3D ``sigmoid_alpha_blend`` as ``SoftSilhouetteShader`` applies it (environment.py:242,263): K-buffers
    ``(N,H,W,K)`` -> RGBA images ``(N,H,W,4)`` with RGB = 1 and alpha = 1 - prod_k (1 - sigmoid(-d_k / sigma) [face >= 0]);
    differentiable w.r.t. ``dists``.  With ``rasterize_meshes`` this is the reference's silhouette renderer at
    operator level."""
    return _SigmoidAlphaBlend.apply(dists, pix_to_face, sigma)
