#!/usr/bin/env python
"""The replicated learner's load as the world grows: occ_ppo_update over M samples x 80 epochs on ONE
GPU.  Config 5 gathers 2 048 envs x T = 50 = 102 400 samples on every rank of an 8-GPU node (PPO.py:196-217 runs its 80
epochs over the whole buffer); one rank alone has 12 800.

  python scripts/ppo_learner_scale.py [lib=path[:max_blocks]] ...   -> JSON on stdout
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from occlusionenv_amd import _native as nat  # noqa: E402


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in nat.SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def run(lib, max_blocks, M, epochs=80, reps=5):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    feats = torch.rand(M, 256, device=dev, generator=g)
    actions = torch.randn(M, 2, device=dev, generator=g)
    old_lp = -1.5 + 0.2 * torch.randn(M, device=dev, generator=g)
    returns = torch.randn(M, device=dev, generator=g)
    P = lambda *shape: (torch.randn(*shape, device=dev, generator=g) * 0.05).contiguous()  # noqa: E731
    w_a, b_a, w_v, b_v = P(2, 256), P(2), P(1, 256), P(1)
    m, v, step = torch.zeros(771, device=dev), torch.zeros(771, device=dev), torch.zeros(1, device=dev)
    scratch = torch.empty(max_blocks * 773, device=dev)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    losses = torch.empty(epochs, 2, device=dev)
    ps = nat.OccPpoState()
    for name, t in (("w_a", w_a), ("b_a", b_a), ("w_v", w_v), ("b_v", b_v), ("adam_m", m), ("adam_v", v), ("adam_step", step)):
        setattr(ps, name, t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def update():
        nat.check(lib.occ_ppo_update(C.c_void_p(feats.data_ptr()), C.c_void_p(actions.data_ptr()), C.c_void_p(old_lp.data_ptr()),
                                     C.c_void_p(returns.data_ptr()), M, 0.36, 0.2, 3e-4, 1e-3, 0.9, 0.999, 1e-8, C.byref(ps), epochs,
                                     C.c_void_p(losses.data_ptr()), C.c_void_p(scratch.data_ptr()), C.c_void_p(counter.data_ptr()),
                                     stream), "occ_ppo_update")

    update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        update()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    assert torch.isfinite(losses).all()
    return ms


def main():
    specs = sys.argv[1:] or ["head=occlusionenv_amd/libocc_hip.so:64"]
    out = []
    for spec in specs:
        name, rest = spec.split("=", 1)
        path, _, mb = rest.partition(":")
        lib = load(path)
        for M in (12800, 25600, 51200, 102400):
            ms = run(lib, int(mb or 64), M)
            out.append(dict(variant=name, max_blocks=int(mb or 64), samples=M, epochs=80, update_ms=ms, us_per_epoch=ms / 80 * 1e3,
                            feature_GBps=M * 1024 * 80 / (ms * 1e-3) / 1e9))
            print(out[-1], file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
