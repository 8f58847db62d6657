"""Diagnostic (GPU box): when the waves of occ_raster2_kernel finish - from an OCC_DBG_ENDS build (plain stores, no atomics:
the launch runs at production speed).    OCC_HIP_LIB=build/ab/libocc_ends.so python scripts/dbg/wave_ends.py [envs]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from occlusionenv_amd import _native as nat
lib = nat.load()
lib.occ_debug_ends.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.cuda.set_device(0)
venv, _ = bench.build_env("shapenet5k", N, 128, seed=42, pool_models=1024)
az0 = (torch.rand(N, generator=torch.Generator().manual_seed(42)) * 2 - 1) * 0.6
venv._reset_envs(list(range(N)), az0)
if venv.engine.R:
    venv._warm_reserve()
g = torch.Generator(device="cuda").manual_seed(7)
buf = (ctypes.c_ulonglong * (4 * 4096))()
for it in range(8):
    a = torch.randn(N, 2, device="cuda", generator=g).requires_grad_(True)
    obs, rew, done, info = venv.step(a)
    rew.sum().backward()
    torch.cuda.synchronize()
    if it < 5:
        continue
    lib.occ_debug_ends(buf)
    d = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4).astype(np.int64)
    d = d[d[:, 2] > 0]
    # NOTE: the buffer holds the LAST launch of the kernel in the step (the full soft+hard+grad launch comes last only if no reset render follows)
    t0 = d[:, 0].min()
    start, last, ex = (d[:, 0] - t0) / 100.0, (d[:, 1] - t0) / 100.0, (d[:, 2] - t0) / 100.0
    items = d[:, 3] & 0xFFFFFFFF
    xcc = d[:, 3] >> 32
    span = ex.max()
    print("step %d: waves %d, span %.0f us; wave starts: max %.0f us; end of last item: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; exit: mean %.0f max %.0f; "
          "last item -> exit: mean %.1f p90 %.1f max %.1f us; idle wave-slot time after exit %.1f %%, after the last item %.1f %%; items/wave mean %.1f min %d max %d"
          % (it, len(d), span, start.max(), last.mean(), np.percentile(last, 50), np.percentile(last, 90), np.percentile(last, 99), last.max(), ex.mean(), ex.max(),
             (ex - last).mean(), np.percentile(ex - last, 90), (ex - last).max(), 100 * (1 - ex.mean() / span), 100 * (1 - last.mean() / span), items.mean(), items.min(), items.max()), flush=True)
    print("   per XCD: last item end mean / max, exit max:", " ".join("(%d: %.0f / %.0f, %.0f)" % (x, last[xcc == x].mean(), last[xcc == x].max(), ex[xcc == x].max()) for x in range(8)), flush=True)
    hist, _ = np.histogram(ex, bins=np.arange(0, span + 50, 50))
    print("   exits per 50 us:", " ".join("%d:%d" % (50 * i, h) for i, h in enumerate(hist) if h), flush=True)
