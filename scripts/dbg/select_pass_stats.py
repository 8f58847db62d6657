"""CPU estimate (oracle only): how many entries are left in the boundary bucket of an overflowing pixel after ONE 32-bucket
histogram pass of the raster kernel's radix select (occ_raster2.hpp: select_topk), with the tile-wide window the kernel uses
today and with a per-pixel window [tile minimum, the pixel's own largest key].    python scripts/dbg/select_pass_stats.py [n_objects]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import p3d_restate as O
from occlusionenv_amd.meshes import SyntheticShapeNet

n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 6
S, K, KALL = 128, 100, 1024
ds = SyntheticShapeNet(n_models=16, seed=1234)
g = torch.Generator().manual_seed(3)


def keys_of(z):
    return (z.astype(np.float32).view(np.uint32) | np.uint32(0x80000000)).astype(np.int64)


def boundary(keys, L, sh, need):
    """One pass: bucket of the need-th smallest key >= L; returns (entries in that bucket, entries needed from it)."""
    b = (keys - L) >> sh
    b = b[(keys >= L) & (b < 32)]
    cnt = np.bincount(b.astype(np.int64), minlength=32)
    cum = np.cumsum(cnt)
    j = int(np.searchsorted(cum, need))
    if j >= 32:
        return 0, 0
    return int(cnt[j]), need - (int(cum[j - 1]) if j else 0)


res = {"tile": [], "pixel": [], "pixel_minmax": []}
tiles = {"tile": [], "pixel": [], "pixel_minmax": []}
for k in range(n_obj):
    v, f = ds.models[k % len(ds.models)]
    az = float((torch.rand(1, generator=g) * 2 - 1) * 0.6)
    off = torch.tensor([float(torch.randn(1, generator=g)) * 0.5, 0.0, float(k % 3)])
    R, T = O.look_at_view_transform(torch.tensor([4.0]), torch.tensor([0.0]), torch.tensor([az]))
    fv = O.world_to_ndc(v + off, R[0], T[0])[f].contiguous()
    p2f, zbuf, _, _ = O.rasterize_meshes(fv, S, O.BLUR_RADIUS, KALL)
    p2f, zbuf = p2f.numpy().reshape(S, S, KALL), zbuf.numpy().reshape(S, S, KALL)
    zmin = fv[:, :, 2].min(1).values.numpy()
    cnt = (p2f >= 0).sum(-1)
    for ty in range(0, S, 8):
        for tx in range(0, S, 8):
            c = cnt[ty:ty + 8, tx:tx + 8]
            if c.max() <= K:
                continue
            allk, allf = [], []
            for yy in range(ty, ty + 8):
                for xx in range(tx, tx + 8):
                    n = int(cnt[yy, xx])
                    allk.append(keys_of(zbuf[yy, xx, :n]))
                    allf.append(p2f[yy, xx, :n])
            kk = np.concatenate(allk)
            kmin_tile = int(keys_of(zmin[np.concatenate(allf)]).min())
            L0 = max(kmin_tile - 4096, 0)
            kmx = int(kk.max())
            rng = max(kmx - L0, 1)
            sh_tile = max(0, int(rng).bit_length() - 5)
            worst = {m: 0 for m in res}
            for keys in allk:
                if keys.size <= K:
                    continue
                m, _ = boundary(keys, L0, sh_tile, K)
                res["tile"].append(m)
                worst["tile"] = max(worst["tile"], m)
                pmax = ((int(keys.max()) >> 16) + 1) << 16
                sh_p = max(0, int(max(pmax - L0, 1)).bit_length() - 5)
                m, _ = boundary(keys, L0, sh_p, K)
                res["pixel"].append(m)
                worst["pixel"] = max(worst["pixel"], m)
                pmin = (int(keys.min()) >> 16) << 16
                sh_q = max(0, int(max(pmax - pmin, 1)).bit_length() - 5)
                m, _ = boundary(keys, pmin, sh_q, K)
                res["pixel_minmax"].append(m)
                worst["pixel_minmax"] = max(worst["pixel_minmax"], m)
            for m in res:
                tiles[m].append(worst[m])
    print("object", k, {m: (len(res[m]), float(np.mean(res[m]))) for m in res}, flush=True)
for m in res:
    a, t = np.array(res[m]), np.array(tiles[m])
    print("%-13s pixels %d: boundary-bucket entries mean %.1f  median %.0f  90%% %.0f  99%% %.0f | tiles %d: all pixels <= 8: %.0f %%  <= 16: %.0f %%  <= 24: %.0f %%  <= 32: %.0f %%" % (
        m, a.size, a.mean(), np.median(a), np.percentile(a, 90), np.percentile(a, 99), t.size, 100 * (t <= 8).mean(), 100 * (t <= 16).mean(),
        100 * (t <= 24).mean(), 100 * (t <= 32).mean()))
