#!/usr/bin/env python
"""A/B timing of ENGINE SETTINGS on one box in one process (GPU box): the bench workload is built once, every variant
(`name=attr:value[,attr:value...]`, attributes of OcclusionEngine; `ring:k` switches the output ring, `lib:path` the library build) then runs the same
step loop, interleaved over several cycles.

  python scripts/ab_toggle.py --steps 40 --cycles 3 lds=setup_vertex_lds:1 glob=setup_vertex_lds:0
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from occlusionenv_amd import _native as nat  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cycles", type=int, default=3)
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--img", type=int, default=128)
    ap.add_argument("--workload", default="shapenet5k")
    ap.add_argument("--pool-models", type=int, default=1024)
    args = ap.parse_args()
    specs = []
    for v in args.variants:
        name, rest = v.split("=", 1)
        specs.append((name, [tuple(kv.split(":")) for kv in rest.split(",") if kv]))
    torch.cuda.set_device(0)
    venv, _ = bench.build_env(args.workload, args.envs, args.img, seed=42, pool_models=args.pool_models)
    eng = venv.engine
    az0 = (torch.rand(args.envs, generator=torch.Generator().manual_seed(42)) * 2 - 1) * 0.6
    venv._reset_envs(list(range(args.envs)), az0)
    if eng.R:
        venv._warm_reserve()
    gen = torch.Generator(device=eng.device).manual_seed(7)
    lib = eng.lib
    libs = {}

    def one_step():
        a = torch.randn(args.envs, 2, device=eng.device, generator=gen, requires_grad=True)
        obs, rewards, dones, infos = venv.step(a)
        rewards.sum().backward()

    res = {name: dict(raster=[], step=[]) for name, _ in specs}
    for cyc in range(args.cycles):
        for name, kvs in specs:
            venv._drain()
            torch.cuda.synchronize()
            for k, v in kvs:
                if k == "ring":
                    venv.use_output_ring(int(v))
                elif k == "lib":  # another build of the same sources (same ABI)
                    if v not in libs:
                        libs[v] = C.CDLL(os.path.abspath(v))
                        for name_, (res_, args_) in nat.SYMBOLS.items():
                            fn = getattr(libs[v], name_)
                            fn.restype, fn.argtypes = res_, args_
                        assert libs[v].occ_abi_version() == nat.ABI_VERSION, v
                    eng.lib = lib = libs[v]
                else:
                    setattr(eng, k, type(getattr(eng, k))(int(v)))
            eng._scene_cache = None
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
            print(f"[cycle {cyc}] {name:>10s} raster {ms.value / max(n.value, 1):.3f} ms  step {dt / args.steps * 1e3:.3f} ms  "
                  f"step-raster {dt / args.steps * 1e3 - ms.value / max(n.value, 1):.3f}", flush=True)
    print("---- means ----")
    for name, _ in specs:
        r = res[name]
        st, ra = sum(r["step"]) / len(r["step"]), sum(r["raster"]) / len(r["raster"])
        print(f"{name:>10s} raster {ra:.3f} ms  step {st:.3f} ms (min {min(r['step']):.3f})  step-raster {st - ra:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
