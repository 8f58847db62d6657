#!/usr/bin/env python
"""Single-env latency of the trainRL.py path (GPU box): ONE OcclusionEnv (the reference drives a single env at the
default 512x512, /root/reference/trainRL.py:75,191-198; environment.py:202), step + reward.backward(), synchronised
after every step like a caller that feeds the observation to its policy.

  python scripts/single_env_latency.py [out.json]
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from occlusionenv_amd import _native as nat  # noqa: E402
from occlusionenv_amd.environment import OcclusionEnv, seed_scene_rng  # noqa: E402
from occlusionenv_amd.meshes import SyntheticShapeNet  # noqa: E402


def run(ds, img, steps=60, warmup=10, label=""):
    np.random.seed(5)
    seed_scene_rng(5)
    env = OcclusionEnv(ds, img_size=img)
    env.reset()
    lib = nat.load()
    g = torch.Generator().manual_seed(3)
    lat = []
    for t in range(warmup + steps):
        if t == warmup:
            nat.check(lib.occ_profile_enable(1), "profile")
        a = torch.nn.Parameter(torch.randn(2, generator=g).to(env.device))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        obs, reward, done, info = env.step(a)
        reward.backward()
        torch.cuda.synchronize()
        if t >= warmup:
            lat.append((time.perf_counter() - t0) * 1e3)
        if bool(done):
            env.reset()
    ms, n = C.c_double(0.0), C.c_int(0)
    nat.check(lib.occ_profile_read(C.byref(ms), C.byref(n)), "profile_read")
    lib.occ_profile_enable(0)
    lat.sort()
    return dict(case=label, img=img, steps=steps, step_ms_median=lat[len(lat) // 2], step_ms_p90=lat[int(len(lat) * 0.9)],
                step_ms_min=lat[0], raster_kernel_ms_mean=ms.value / max(n.value, 1), raster_launches=n.value)


if __name__ == "__main__":
    torch.cuda.set_device(0)
    ds = SyntheticShapeNet(n_models=64, seed=1234)
    out = []
    for label, data in (("three teapots (default scene)", None), ("3 x 5120-face synthetic ShapeNet-size meshes", ds)):
        for img in (128, 256, 512):
            r = run(data, img, label=label)
            print(json.dumps(r), flush=True)
            out.append(r)
    if len(sys.argv) > 1:
        json.dump(dict(note="one OcclusionEnv, step + reward.backward(), host-synchronised every step (latency, not "
                            "throughput); raster_kernel_ms = HIP events around occ_raster2_kernel", results=out),
                  open(sys.argv[1], "w"), indent=1)
