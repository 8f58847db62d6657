"""SURVEY.md §5 (race detection / sanitizers): the CPU oracle's C sources under AddressSanitizer + UBSan, and the
C-ABI's host-side argument paths with the sanitizer runtime loaded.  GPU AddressSanitizer is not available on this
pool, so the device code is covered by the OCC_DBG_BOUNDS build instead (tests/test_gpu_env_api.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


CODE = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch
from oracle import p3d_restate as O
from occlusionenv_amd.meshes import load_obj
v, f = load_obj(os.path.join(%(root)r, "data", "teapot.obj"))
objs = [(v, f), (v + torch.tensor([0.5, 0.0, 1.0]), f), (v + torch.tensor([-0.5, 0.0, 2.0]), f)]
for radius, S in ((4.0, 24), (1.2, 16)):          # far view; camera inside the scene: z-clipped faces, pair rule
    for dt in (torch.float32, torch.float64):     # both builds of raster_naive.c
        env = O.OracleEnv([(a.to(dt), b) for a, b in objs], S, dtype=dt)
        env.reset(radius=radius, azimuth=0.3)
        a = torch.tensor([0.3, -0.2], dtype=dt, requires_grad=True)
        obs, r, d, info = env.step(a)              # forward: orc_rasterize_naive x4
        r.backward()                               # backward: orc_rasterize_backward_dists x3
        assert torch.isfinite(a.grad).all()
ndc = O.world_to_ndc(v, env.R[0].float(), env.T[0].float())
c = O.pixel_candidates(ndc[f], 16, 8, 8, O.BLUR_RADIUS)   # the tie classifier's dump
print("SAN-OK", len(c["f"]))
"""


@pytest.mark.skipif(_libasan() is None, reason="libasan.so not found")
def test_oracle_c_sources_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, ORC_LIB=os.path.join(ROOT, "oracle", "liborc_asan.so"), LD_PRELOAD=_libasan(),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-c", CODE % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "SAN-OK" in out.stdout and "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]


@pytest.mark.skipif(_libasan() is None, reason="libasan.so not found")
def test_cabi_argument_validation_with_sanitizer_runtime():
    """Every C-ABI entry point must turn null / nonsensical arguments into OCC_ERR_ARG before touching memory; run with
    ASan's allocator and interceptors in the process (host code of libocc_hip.so is not instrumented: no GPU ASan here)."""
    code = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from occlusionenv_amd import _native as nat
lib = nat.load()
sc, ws, ro, sz = nat.OccScene(), nat.OccWorkspace(), nat.OccRenderOut(), nat.OccWorkspaceSizes()
assert lib.occ_workspace_query(None, 0, C.byref(sz)) == 1 and lib.occ_workspace_query(C.byref(sc), 0, None) == 1
assert lib.occ_workspace_query(C.byref(sc), 16, C.byref(sz)) == 1           # n_env = 0, img = 0
sc.n_env, sc.img, sc.rec_cap = 4, 64, 128
assert lib.occ_workspace_query(C.byref(sc), 16, C.byref(sz)) == 0 and sz.lists_bytes > 0 and sz.n_slots == 16
assert lib.occ_record_sizes(100, 4, C.byref(sz)) == 1 and lib.occ_record_sizes(128, 4, C.byref(sz)) == 0
assert lib.occ_camera(0, None, None, None, None, None, None, 4, None) == 1 and lib.occ_camera(7, None, None, None, None, None, None, 4, None) == 1
assert lib.occ_render(C.byref(sc), None, C.byref(ws), C.byref(ro), 3, 100, None) == 1
assert lib.occ_render(None, None, None, None, 3, 100, None) == 1
assert lib.occ_step_finish(None, None, None, None, None, None, None, None, 4, None) == 1
assert lib.occ_rasterize_meshes_naive(None, None, None, None, 1, 8, 8, 0.0, 1, 1, 0, 1, None, None, None, None, None) == 1
assert lib.occ_sigmoid_alpha_blend_fwd(None, None, 4, 1, 1e-4, None, None) == 1
assert lib.occ_auto_reset(None, None, None, 4, 2, None, None, None, None, None, None, None, 64, None, None, None, None) == 1
assert lib.occ_reserve_refill(None, 3, 4, 2, None, None, None, None, None) == 1 and lib.occ_reserve_refill(None, 0, 4, 2, None, None, None, None, None) == 0
print("ARG-OK")
""" % ROOT
    env = dict(os.environ, LD_PRELOAD=_libasan(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:protect_shadow_gap=0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ARG-OK" in out.stdout and "ERROR: AddressSanitizer" not in out.stderr, out.stderr[-3000:]
