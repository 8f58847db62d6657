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


def generate(venv, root: str, num_frames: int = 20, lr: float = 2.5e-2, first_run: int = 0, generator=None) -> int:
    """datasetGenerator.py:76-124 for all envs of ``venv`` at once; returns the number of runs written."""
    os.makedirs(root, exist_ok=True)
    N = venv.num_envs
    eng = venv.engine
    writers = [RunWriter(root, first_run + i) for i in range(N)]
    venv.reset()
    for j in range(num_frames):
        action = (lr * torch.randn(N, 2, device=eng.device, generator=generator)).requires_grad_(True)
        obs, rewards, dones, infos = venv.step(action)
        rewards.sum().backward()
        grad = action.grad.detach().cpu().numpy()
        obs_h = obs.detach().cpu().numpy()
        occl = torch.cat([infos[i]["full_state"] for i in range(N)])[..., 3].detach().cpu().numpy()
        el, az = eng.elevation.cpu().numpy(), eng.azimuth.cpu().numpy()
        for i in range(N):
            if not np.isnan(grad[i]).any():  # datasetGenerator.py:94-95
                writers[i].write_frame(j, obs_h[i], occl[i], float(el[i]), float(az[i]), grad[i])
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
