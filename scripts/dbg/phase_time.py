"""Diagnostic (GPU box): shader cycles per phase of occ_raster2_kernel from an OCC_DBG_TIME build.
   OCC_HIP_LIB=build/dbg2/libocc_time.so python scripts/dbg/phase_time.py [envs]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.parity_utils import make_case
from occlusionenv_amd.engine import OcclusionEngine
from occlusionenv_amd import _native as nat
lib = nat.load()
lib.occ_debug_time.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
names = ["dequeue", "decode+init", "scan", "stage+count", "expand", "rounds", "bounds", "compaction(rest)", "final select(rest)", "result stores", "sel:setup", "sel:hist sweeps", "sel:bucket scans", "sel:final sweep", "sel:lists", "-"]
for mesh, img in [("synthetic", 128), ("mixed", 128), ("teapot", 128)]:
    case = make_case(N, 11, mesh)
    eng = OcclusionEngine(case["pool"], N, img)
    eng.set_scene(list(range(N)), case["mesh_ids"], case["offsets"])
    eng.reset_render(None, 4.0, case["az"], 0.0)
    a = case["actions"].cuda().requires_grad_(True)
    eng.step(a)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 112)()
    lib.occ_debug_time(buf)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    eng.step(case["actions"].cuda().requires_grad_(True))
    ev1.record()
    torch.cuda.synchronize()
    lib.occ_debug_time(buf)
    tot = float(sum(buf[:16]))
    print(mesh, img, "N", N, "step %.2f ms; wave-cycles %.3e:" % (ev0.elapsed_time(ev1), tot),
          "  ".join("%s %.1f%%" % (n, 100.0 * buf[i] / tot) for i, n in enumerate(names)), flush=True)
    t0, t1, tsum, nw = buf[16], buf[17], buf[18], max(buf[19], 1)
    span = (t1 - t0) / 100.0  # us
    print("   waves %d: span %.0f us, mean wave end at %.0f us -> %.1f%% of the wave-slot time is after a wave's end;"
          % (nw, span, (tsum / nw - t0) / 100.0, 100.0 * (1.0 - (tsum / nw - t0) / max(t1 - t0, 1))),
          "items with selection: %d, mean %.0f us, max %.0f us; others: %d, mean %.1f us"
          % (buf[21], buf[20] / max(buf[21], 1) / 100.0, buf[22] / 100.0, buf[24], buf[23] / max(buf[24], 1) / 100.0), flush=True)
    print("   wave ends per 100 us (count:mean last item us):", " ".join("[%d]%d:%.0f" % (b, buf[32 + b], buf[64 + b] / max(buf[32 + b], 1) / 100.0) for b in range(32) if buf[32 + b]), flush=True)
    print("   last item end -> exit: mean %.1f us, max %.1f us; waves that started > 50 us late: %d" % (buf[25] / nw / 100.0, buf[26] / 100.0, buf[27]))
    print("   per XCD (first start, last end) us after the earliest start:", " ".join("(%.0f, %.0f)" % ((buf[96 + x] - t0) / 100.0, (buf[104 + x] - t0) / 100.0) for x in range(8)), flush=True)
    del eng
