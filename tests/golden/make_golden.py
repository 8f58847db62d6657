"""Generates tests/golden/vecenv_golden.npz by importing the parts of the reference that ARE importable in the
authoring container (baseVecEnv.py imports fine; environment.py / SubProcVecEnv.py do not, gym
and pytorch3d are absent).  Run once, in the authoring container only:

    python tests/golden/make_golden.py

The reference never travels to the GPU box; only the small .npz of inputs/outputs is committed.
"""
import inspect
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
import baseVecEnv as ref  # noqa: E402

rng = np.random.default_rng(20261003)
out = {}
for n, h, w, c in [(1, 3, 4, 3), (4, 2, 2, 1), (5, 3, 2, 3), (7, 4, 4, 4), (9, 2, 3, 3)]:
    x = rng.random((n, h, w, c)).astype(np.float32)
    out[f"tile_in_{n}"] = x
    out[f"tile_out_{n}"] = ref.tile_images(x)
out["vecenv_abstract"] = np.array(sorted(ref.VecEnv.__abstractmethods__))
out["vecenv_methods"] = np.array(sorted(n for n, _ in inspect.getmembers(ref.VecEnv) if not n.startswith("__")))
out["wrapper_methods"] = np.array(sorted(n for n, _ in inspect.getmembers(ref.VecEnvWrapper) if not n.startswith("__")))
out["step_sig"] = np.array(str(inspect.signature(ref.VecEnv.step)))
out["init_sig"] = np.array(str(inspect.signature(ref.VecEnv.__init__)))
out["err_already"] = np.array(str(ref.AlreadySteppingError()))
out["err_not"] = np.array(str(ref.NotSteppingError()))
# ---- H1 pin: what the reference's agent makes of an observation --------------------------
# model.FullNetwork(8, dilation=2, separable=True) is the encoder PPO.ActorCritic builds (PPO.py:47); its pooled
# features are what PPO.select_action stores per env-step (PPO.py:155-162).  Shapes only: weights are random-init.
import torch  # noqa: E402

import model as ref_model  # noqa: E402

torch.manual_seed(0)
net = ref_model.FullNetwork(8, dilation=2, separable=True).eval()
out["h1_param_count"] = np.array(sum(p.numel() for p in net.parameters()))
shapes = []
for shp in [(1, 4, 64, 64), (1, 4, 128, 128), (8, 4, 128, 128)]:
    with torch.no_grad():
        feats, segm, gradp = net(torch.rand(*shp))
        act, val = net.act(feats)
    shapes.append([list(shp), list(feats.shape), list(segm.shape), list(gradp.shape), list(act.shape), list(val.shape)])
out["h1_shapes"] = np.array(repr(shapes))  # [obs, pooled_features, segm, grad_pred, action_scores, value] per case
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vecenv_golden.npz"), **out)
print({k: getattr(v, "shape", None) for k, v in out.items()})
