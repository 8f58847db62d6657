#!/usr/bin/env python
"""A/B timing of library variants on ONE box in ONE process (GPU box): the bench workload is built once, every
variant (another build of the same sources, `name=path[:waves_per_cu]`) then runs the same step loop, interleaved
over several cycles so that clock / box drift hits all variants alike.

  python scripts/ab_bench.py --steps 30 --cycles 2 base=occlusionenv_amd/libocc_hip.so v1=build/ab/libocc_v1.so:16

Prints one line per variant: raster kernel ms (HIP events around the kernel, occ_profile_*) and ms per step.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from occlusionenv_amd import _native as nat  # noqa: E402


def load_variant(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in nat.SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    assert lib.occ_abi_version() == nat.ABI_VERSION, path
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--cycles", type=int, default=2)
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--img", type=int, default=128)
    ap.add_argument("--workload", default="shapenet5k")
    ap.add_argument("--pool-models", type=int, default=1024)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    specs = []
    for v in args.variants:
        name, rest = v.split("=", 1)
        path, _, waves = rest.partition(":")
        specs.append((name, load_variant(path), int(waves) if waves else None))
    torch.cuda.set_device(0)
    venv, _ = bench.build_env(args.workload, args.envs, args.img, seed=42, pool_models=args.pool_models)
    eng = venv.engine
    az0 = (torch.rand(args.envs, generator=torch.Generator().manual_seed(42)) * 2 - 1) * 0.6
    venv._reset_envs(list(range(args.envs)), az0)
    if eng.R:
        venv._warm_reserve()
    gen = torch.Generator(device=eng.device).manual_seed(7)
    default_waves = eng.waves_per_cu

    def one_step():
        a = torch.randn(args.envs, 2, device=eng.device, generator=gen, requires_grad=True)
        obs, rewards, dones, infos = venv.step(a)
        rewards.sum().backward()

    res = {name: dict(raster=[], step=[]) for name, _, _ in specs}
    for cyc in range(args.cycles):
        for name, lib, waves in specs:
            venv._drain()
            torch.cuda.synchronize()
            eng.lib = lib
            w = waves or default_waves
            eng.waves_per_cu = w
            eng._ws_key = None  # the workspace is re-made for every variant: its sizes are the LIBRARY's (occ_workspace_query)
            for _ in range(args.warmup):
                one_step()
            nat.check(lib.occ_profile_enable(1), "profile")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                one_step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ms, n = C.c_double(0.0), C.c_int(0)
            nat.check(lib.occ_profile_read(C.byref(ms), C.byref(n)), "profile_read")
            lib.occ_profile_enable(0)
            eng.check_status()
            res[name]["raster"].append(ms.value / max(n.value, 1))
            res[name]["step"].append(dt / args.steps * 1e3)
            print(f"[cycle {cyc}] {name:>12s} waves/CU {w:2d}  raster {ms.value / max(n.value, 1):.3f} ms  step {dt / args.steps * 1e3:.3f} ms",
                  flush=True)
    print("---- means ----")
    for name, _, waves in specs:
        r = res[name]
        print(f"{name:>12s} raster {sum(r['raster']) / len(r['raster']):.3f} ms (min {min(r['raster']):.3f})  "
              f"step {sum(r['step']) / len(r['step']):.3f} ms (min {min(r['step']):.3f})", flush=True)
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
