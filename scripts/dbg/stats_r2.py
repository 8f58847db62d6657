"""Diagnostic (GPU box): loop trip counts of the raster kernel from an OCC_DBG_STATS build.
   OCC_HIP_LIB=build/dbg/libocc_stats.so python scripts/dbg/stats_r2.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.parity_utils import make_case
from occlusionenv_amd.engine import OcclusionEngine
from occlusionenv_amd import _native as nat
lib = nat.load()
lib.occ_debug_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
N = 256
for mesh, img in [("synthetic", 128), ("teapot", 128), ("mixed", 128), ("synthetic", 256)]:
    case = make_case(N, 11, mesh)
    eng = OcclusionEngine(case["pool"], N, img)
    eng.set_scene(list(range(N)), case["mesh_ids"], case["offsets"])
    eng.reset_render(None, 4.0, case["az"], 0.0)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    lib.occ_debug_stats(buf)
    a = case["actions"].cuda().requires_grad_(True)
    eng.step(a)
    torch.cuda.synchronize()
    lib.occ_debug_stats(buf)
    st = [x / N for x in buf]
    al = eng.alphas
    npx = float((al > 0).float().sum()) / N
    print(mesh, img, "per env: items %.1f chunk_rows %.0f stagings %.0f staged_pairs %.0f iters %.0f cands %.0f "
          "cands_in_ovf_px %.0f items_with_ovf %.1f  covered pixel-objects %.0f | hist passes %.1f  swept log entries (per pass) %.0f  "
          "ovf pixels %.0f  re-accumulated entries %.0f  final-sweep sub-passes %.0f" % (st[0], st[5], st[1], st[2], st[3], st[4], st[6], st[7], npx,
                                                                                        st[8], st[9], st[10], st[11], st[12]), flush=True)
    if st[14]:
        print("   rounds with accepted pairs %.0f: sub-passes per such round with 4 / 5 / 6 / 8 copies %.2f / %.2f / %.2f / %.2f; faces per round %.2f; "
              "counting accepted lanes' faces only: 4 copies %.2f, 6 copies %.2f; accepted pairs per round %.1f" % (
                  st[14], st[15] / st[14], st[16] / st[14], st[17] / st[14], st[18] / st[14], st[19] / st[14], st[20] / st[14], st[21] / st[14], st[22] / st[14]), flush=True)
    if st[14]:
        print("   exact schedules, sub-passes per round: copy = face & 3, rank among live / accepted lanes of (pixel, copy) %.2f / %.2f; "
              "copy = rank & 3 among live / accepted lanes of the pixel %.2f / %.2f" % (st[23] / st[14], st[24] / st[14], st[25] / st[14], st[26] / st[14]), flush=True)
    del eng
