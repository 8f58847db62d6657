"""CPU estimate (oracle only, no GPU) of what a per-tile front-to-back order would prune on the bench's scenes: for every
pixel the candidates arrive by their face's nearest-vertex depth; once the pixel holds K of them, B = the largest stored
depth; a later face with zmin > B can be skipped BEFORE evaluation (pair-level pruning), one with depth >= B after it
(not logged).  Prints, over the pixels of tiles that hold an overflowing pixel: candidates, evaluated, logged.
  python scripts/dbg/prune_potential.py [n_objects]"""
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
tot = dict(cand=0, cand_ovf_tiles=0, evaluated=0, logged=0, ovf_pixels=0, tiles=0, ovf_tiles=0, kept=0)
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
    assert cnt.max() < KALL, "raise KALL"
    for ty in range(0, S, 8):
        for tx in range(0, S, 8):
            c = cnt[ty:ty + 8, tx:tx + 8]
            if c.sum() == 0:
                continue
            tot["tiles"] += 1
            tot["cand"] += int(c.sum())
            if c.max() <= K:
                continue
            tot["ovf_tiles"] += 1
            tot["cand_ovf_tiles"] += int(c.sum())
            for yy in range(ty, ty + 8):
                for xx in range(tx, tx + 8):
                    n = int(cnt[yy, xx])
                    if n == 0:
                        continue
                    fs, zs = p2f[yy, xx, :n], zbuf[yy, xx, :n]
                    if n <= K:  # never reaches K: everything is evaluated and logged
                        tot["evaluated"] += n
                        tot["logged"] += n
                        continue
                    tot["ovf_pixels"] += 1
                    tot["kept"] += K
                    order = np.lexsort((fs, zmin[fs]))
                    stored_max, stored = -np.inf, 0
                    for i in order:
                        if stored >= K and zmin[fs[i]] > stored_max:
                            break  # sorted by zmin: every later face is beyond the bound as well
                        tot["evaluated"] += 1
                        if stored < K or zs[i] < stored_max:
                            tot["logged"] += 1
                            stored += 1
                            # bound = largest depth among the stored (a looser bound than the K-th smallest, as the kernel keeps it)
                            stored_max = max(stored_max, zs[i]) if stored <= K else stored_max
    print("object", k, {a: b for a, b in tot.items()}, flush=True)
c = tot["cand_ovf_tiles"]
print("tiles with an overflowing pixel: %d of %d; their candidates %d of %d (%.0f %%)" % (tot["ovf_tiles"], tot["tiles"], c, tot["cand"], 100.0 * c / max(tot["cand"], 1)))
print("with a per-tile front-to-back order: evaluated %.0f %%, logged %.0f %% of those candidates (kept by the final top-K: %.0f %%)"
      % (100.0 * tot["evaluated"] / max(c, 1), 100.0 * tot["logged"] / max(c, 1), 100.0 * (tot["kept"] + tot["evaluated"] * 0) / max(c, 1)))
