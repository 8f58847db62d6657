"""Known answers for the oracle's TexturesAtlas.sample_textures restatement (oracle/p3d_restate.py:sample_atlas)."""
import torch

from oracle import p3d_restate as O


def _atlas(R=4):
    a = torch.zeros(2, R, R, 3)
    for y in range(R):
        for x in range(R):
            a[0, y, x] = torch.tensor([x / 10.0, y / 10.0, 0.5])
            a[1, y, x] = torch.tensor([0.9, x / 10.0, y / 10.0])
    return a


def test_texel_grid_and_mirror():
    a = _atlas()
    p2f = torch.tensor([[0], [0], [0], [1], [-1]])
    bary = torch.tensor([[[0.1, 0.1, 0.8]],      # cell (0,0), below the diagonal
                         [[0.6, 0.3, 0.1]],      # w*R = (2.4, 1.2): sum 3.6 - 3 = 0.6 <= 1 -> (x=2, y=1)
                         [[0.45, 0.45, 0.1]],    # w*R = (1.8, 1.8): 3.6 - 2 = 1.6 > 1 -> mirrored (x=2, y=2)
                         [[1.0, 0.0, 0.0]],      # clamps to R-1 on face 1
                         [[0.3, 0.3, 0.4]]])     # background
    t = O.sample_atlas(a, p2f, bary)[:, 0]
    assert torch.allclose(t[0], torch.tensor([0.0, 0.0, 0.5]))
    assert torch.allclose(t[1], torch.tensor([0.2, 0.1, 0.5]))
    assert torch.allclose(t[2], torch.tensor([0.2, 0.2, 0.5]))
    assert torch.allclose(t[3], torch.tensor([0.9, 0.3, 0.0]))
    assert torch.equal(t[4], torch.zeros(3))


def test_textured_render_differs_only_in_rgb():
    from occlusionenv_amd.meshes import icosphere
    v, f = icosphere(1)
    v, f = torch.as_tensor(v, dtype=torch.float32) * 0.4, torch.as_tensor(f, dtype=torch.int64)
    R, T = O.look_at_view_transform(torch.tensor([2.0]), torch.tensor([0.3]), torch.tensor([0.5]))
    R, T = R[0], T[0]
    white, z0 = O.hard_flat_rgbd(v, f, R, T, 32)
    g = torch.Generator().manual_seed(0)
    atlas = torch.rand(f.shape[0], 4, 4, 3, generator=g)
    tex, z1 = O.hard_flat_rgbd(v, f, R, T, 32, atlas=atlas)
    assert torch.equal(z0, z1) and torch.equal(white[..., 3], tex[..., 3])
    hit = z0[..., 0] >= 0
    assert (tex[..., :3][hit] <= white[..., :3][hit] + 1e-6).all() and not torch.allclose(tex, white)
