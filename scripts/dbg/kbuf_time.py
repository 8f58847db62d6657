"""Timing of the two operator-level K-buffer kernels (GPU box): python scripts/dbg/kbuf_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import p3d_restate as O
from occlusionenv_amd.meshes import SyntheticShapeNet, load_obj
from occlusionenv_amd.ops import rasterize_meshes

def scene(v, f, az=0.3):
    R, T = O.look_at_view_transform(torch.tensor([4.0]), torch.tensor([0.0]), torch.tensor([az]))
    return O.world_to_ndc(v, R[0], T[0])[f].contiguous()

cases = [("synthetic 5120 faces", scene(*SyntheticShapeNet(n_models=1, seed=5).models[0])),
         ("teapot 2464 faces", scene(*load_obj(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "data", "teapot.obj"))))]
for name, fv in cases:
    for S, K in ((128, 100), (128, 8), (256, 100)):
        x = fv.cuda(); first = torch.tensor([0]).cuda(); num = torch.tensor([fv.shape[0]]).cuda()
        for naive in (False, True):
            for _ in range(2):
                rasterize_meshes(x, first, num, S, O.BLUR_RADIUS, K, True, True, True, naive=naive)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                r = rasterize_meshes(x, first, num, S, O.BLUR_RADIUS, K, True, True, True, naive=naive)
            e1.record(); torch.cuda.synchronize()
            print("%-22s S %3d K %3d %-6s %8.3f ms  covered %d" % (name, S, K, "naive" if naive else "tiled", e0.elapsed_time(e1) / 5, int((r[0][..., 0] >= 0).sum())), flush=True)
