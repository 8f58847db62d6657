"""Offline dataset format of the reference: writer = datasetGenerator.py:76-124, reader = dataset.py:12-80.

Layout: ``<root>/run_%d/{RGB/%d.jpg, Depth/%d.png, Occl/%d.png, params.pickle}``.  RGB = the observation's colour
channels as 8-bit JPEG *in the channel order cv2.imwrite gives an RGB array* (i.e. stored B<->R swapped, like the
reference's files); Depth = view depth * 51 as 8-bit PNG with background 0 (datasetGenerator.py:110-112); Occl = the
occlusion image (alpha of ``info['full_state']``) as 8-bit PNG; params.pickle = one flat float64 array, 5 values per
frame: ``[j, elevation, azimuth, grad0, grad1]`` (datasetGenerator.py:99-100,121-124).

The generator is batched: N envs produce N runs at once, every frame is one vectorised step with a fresh random
action whose reward gradient is the label (datasetGenerator.py:83-117).
"""
from __future__ import annotations

import glob
import os
import os.path as osp
import pickle

import numpy as np
import torch
from PIL import Image


def _ubyte(a: np.ndarray) -> np.ndarray:
    """skimage.img_as_ubyte for floats in [0, 1]: round(x * 255)."""
    return np.clip(np.rint(np.asarray(a, dtype=np.float64) * 255.0), 0, 255).astype(np.uint8)


class RunWriter:
    """One ``run_%d`` directory."""

    def __init__(self, root: str, run: int):
        self.dir = osp.join(root, "run_%d" % run)
        for sub in ("Depth", "RGB", "Occl"):
            os.makedirs(osp.join(self.dir, sub), exist_ok=True)
        self.params = np.empty((0,), dtype=np.float64)

    def write_frame(self, j: int, obs_chw: np.ndarray, occl_hw: np.ndarray, elevation: float, azimuth: float, grad) -> None:
        img = np.transpose(obs_chw, (1, 2, 0))  # (S,S,4), datasetGenerator.py:102
        rgb = _ubyte(img[..., :3])
        Image.fromarray(rgb[..., ::-1].copy()).save(osp.join(self.dir, "RGB", "%d.jpg" % j))  # cv2 writes BGR
        Image.fromarray(_ubyte(occl_hw)).save(osp.join(self.dir, "Occl", "%d.png" % j))
        depth = img[..., 3].copy()
        depth[depth == -1] = 0
        Image.fromarray((depth * 51).astype(np.uint8)).save(osp.join(self.dir, "Depth", "%d.png" % j))
        self.params = np.append(self.params, np.array([j, elevation, azimuth, grad[0], grad[1]], dtype=np.float64))

    def close(self) -> None:
        with open(osp.join(self.dir, "params.pickle"), "wb") as fh:
            pickle.dump(self.params, fh)


def generate(venv, root: str, num_frames: int = 20, lr: float = 2.5e-2, first_run: int = 0, generator=None,
             max_steps: int = 0) -> int:
    """datasetGenerator.py:76-124 for all envs of ``venv`` at once; returns the number of runs written.

    Like the reference, every run stays on ONE scene for all its frames: the envs are stepped through the engine
    directly (``OcclusionEnv.step`` semantics, datasetGenerator.py:88), not through ``SimpleVecEnv.step`` whose
    auto-reset would swap the scene of an env that reports ``finished`` -- the reference ignores ``finished`` here.
    A frame whose gradient is NaN is retried with the SAME action from the pose the failed step left behind, without
    advancing that run's frame counter (datasetGenerator.py:93-95 ``continue``s before drawing a new action)."""
    os.makedirs(root, exist_ok=True)
    N = venv.num_envs
    eng = venv.engine
    writers = [RunWriter(root, first_run + i) for i in range(N)]
    venv.reset()
    venv._drain()  # no auto-reset report may be pending while the engine is driven directly
    frame = np.zeros(N, dtype=np.int64)
    action = lr * torch.randn(N, 2, device=eng.device, generator=generator)
    steps, max_steps = 0, max_steps or 50 * num_frames
    while int(frame.min()) < num_frames:
        if steps >= max_steps:
            raise RuntimeError("generate(): gradients stayed NaN for %d steps" % steps)
        steps += 1
        a = action.clone().requires_grad_(True)
        obs, rewards, dones, full_state, loss = eng.step(a)
        rewards.sum().backward()
        grad = a.grad.detach().cpu().numpy()
        obs_h = obs.detach().cpu().numpy()
        occl = full_state[..., 3].detach().cpu().numpy()
        el, az = eng.elevation.cpu().numpy(), eng.azimuth.cpu().numpy()
        fresh = lr * torch.randn(N, 2, device=eng.device, generator=generator)
        good = torch.zeros(N, dtype=torch.bool)
        for i in range(N):
            if frame[i] < num_frames and not np.isnan(grad[i]).any():
                writers[i].write_frame(int(frame[i]), obs_h[i], occl[i], float(el[i]), float(az[i]), grad[i])
                frame[i] += 1
                good[i] = True
        good = good.to(eng.device)
        action = torch.where(good[:, None], fresh, action)  # a new action only after a frame was written
    for w in writers:
        w.close()
    return N


class OcclusionDataset(torch.utils.data.Dataset):
    """dataset.py:12-80 without the torchvision colour jitter (torchvision is not part of this image): same file
    discovery (sorted recursive globs), same label parsing, same resizing, flips for split == 'train', same tensors:
    ``(img (4,H,W) in [0,1], label (1,H,W), pos (2,), grad (2,))``."""

    def __init__(self, root: str, split: str = "", size=(256, 256)):
        super().__init__()
        self.root, self.split, self.size = root, split, size
        base = osp.join(root, split)
        self.images = sorted(glob.glob(base + "/**/RGB/*.jpg", recursive=True))
        self.depthImages = sorted(glob.glob(base + "/**/Depth/*.png", recursive=True))
        self.labelImages = sorted(glob.glob(base + "/**/Occl/*.png", recursive=True))
        pos, grad = [], []
        for lfile in sorted(glob.glob(base + "/**/*.pickle", recursive=True)):
            with open(lfile, "rb") as fh:
                arr = np.asarray(pickle.load(fh)).reshape([-1, 5])
            pos.append(arr[:, 1:3])
            grad.append(arr[:, 3:5])
        self.posLabels = np.concatenate(pos, 0) if pos else np.zeros((0, 2))
        self.gradLabels = np.concatenate(grad, 0) if grad else np.zeros((0, 2))

    def __len__(self):
        return self.posLabels.shape[0]

    def __getitem__(self, i):
        img = Image.open(self.images[i]).convert("RGB").resize(self.size)
        depth = Image.open(self.depthImages[i]).convert("I").resize(self.size)
        label = Image.open(self.labelImages[i]).convert("1").resize(self.size)
        if self.split == "train" and torch.rand(1) > 0.5:
            img = img.transpose(method=Image.FLIP_LEFT_RIGHT)
            depth = depth.transpose(method=Image.FLIP_LEFT_RIGHT)
            label = label.transpose(method=Image.FLIP_LEFT_RIGHT)
        img_t = torch.tensor(np.asarray(img)) / 255.0
        depth_t = torch.tensor(np.asarray(depth)).unsqueeze(2) / 255.0
        img_t = torch.cat([img_t, depth_t], 2).permute(2, 0, 1)
        label_t = torch.tensor(np.asarray(label)).unsqueeze(0).float()
        return img_t.float(), label_t, torch.tensor(self.posLabels[i]), torch.tensor(self.gradLabels[i])
