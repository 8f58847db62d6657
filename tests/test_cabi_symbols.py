"""The C-ABI shared library loads on a GPU-less host and exports every symbol include/*.h declares.
No compute call is made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "occlusionenv_amd.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(occ_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_documented_entry_points():
    names = _declared()
    for n in ("occ_abi_version", "occ_camera", "occ_render", "occ_step_finish", "occ_workspace_query"):
        assert n in names


def test_library_exports_every_declared_symbol():
    from occlusionenv_amd import _native

    lib = ctypes.CDLL(_native.LIB_PATH)
    for n in _declared():
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert set(_native.SYMBOLS) == set(_declared())
    assert lib.occ_abi_version() == _native.ABI_VERSION


def test_binding_constants_match_header():
    from occlusionenv_amd import _native as nat

    src = open(HEADER).read()

    def define(name):
        return int(re.search(rf"#define\s+{name}\s+(\d+)", src).group(1))

    assert define("OCC_ABI_VERSION") == nat.ABI_VERSION and define("OCC_CAM_STRIDE") == nat.CAM_STRIDE
    assert define("OCC_REC_STRIDE") == nat.REC_STRIDE and define("OCC_TILE") == nat.TILE
    assert define("OCC_LOG_CAP") == nat.LOG_CAP and define("OCC_LOG_ENTRY_BYTES") == nat.LOG_ENTRY_BYTES and define("OCC_MAX_K") == nat.MAX_K
    assert (define("OCC_RENDER_SOFT"), define("OCC_RENDER_HARD"), define("OCC_RENDER_GRAD")) == (1, 2, 4)
    assert (define("OCC_CAM_STEP"), define("OCC_CAM_LOOKAT"), define("OCC_CAM_POSITION")) == (0, 1, 2)
    assert define("OCC_PPO_FEATURES") == nat.PPO_FEATURES and define("OCC_PPO_MAX_BLOCKS") * (nat.PPO_PARAMS + 2) == nat.PPO_SCRATCH_FLOATS


def test_workspace_query_and_argument_checks_need_no_gpu():
    from occlusionenv_amd import _native as nat

    lib = nat.load()
    sc = nat.OccScene()
    sc.n_env, sc.img, sc.rec_cap = 4, 128, 1000
    sizes = nat.OccWorkspaceSizes()
    assert lib.occ_workspace_query(ctypes.byref(sc), 64, ctypes.byref(sizes)) == 0
    assert sizes.rec_bytes == 4 * 3 * 1000 * nat.REC_STRIDE * 4 and sizes.rec_bbox_bytes == 4 * 3 * 1000 * 16 and sizes.scan_bytes == sizes.rec_bbox_bytes
    assert sizes.partials_bytes == 4 * 64 * 16 and sizes.n_slots == 64
    assert sizes.offsets_bytes == 25 * 4 and sizes.queue_bytes == 1024 and sizes.obj_alpha_bytes == 4 * 3 * 128 * 128 * 4 and sizes.obj_grad_bytes == 2 * sizes.obj_alpha_bytes
    assert sizes.lists_bytes == 64 * nat.LOG_CAP * nat.LOG_ENTRY_BYTES  # sized for the log only
    # work-item order: 512 header words + 32 per (env, object) + 3 per tile (rank|class, and the 8-byte item)
    assert sizes.order_bytes == (512 + 4 * 3 * 32 + 4 * 3 * 16 * 16 * 3) * 4
    # an odd tile-table length (odd env count x odd tiles per side) gets one pad word: the 8-byte items stay aligned
    sc.n_env, sc.img = 1, 72
    assert lib.occ_workspace_query(ctypes.byref(sc), 64, ctypes.byref(sizes)) == 0
    assert sizes.order_bytes == (512 + 3 * 32 + 3 * 81 + 1 + 3 * 81 * 2) * 4
    sc.n_env, sc.img = 4, 128
    sc.img = 100  # not a multiple of the tile size
    assert lib.occ_workspace_query(ctypes.byref(sc), 64, ctypes.byref(sizes)) == 1
    # null pointers are rejected before anything is launched
    assert lib.occ_camera(0, None, None, None, None, None, None, 4, None) == 1
    assert lib.occ_render(None, None, None, None, 3, 100, None) == 1
    assert lib.occ_step_finish(None, None, None, None, None, None, None, None, 4, None) == 1
    assert lib.occ_step_flags(None, None, None, 4, 0, None, None) == 1
    assert lib.occ_reset_commit(None, 2, *([None] * 13), 64, None) == 1
    assert lib.occ_reset_commit(None, 0, *([None] * 13), 64, None) == 0  # nothing to commit
    # the learner's entry points: null pointers, a pooling grid that does not divide the image, a non-positive variance
    assert lib.occ_pool8(None, 4, 128, None, None) == 1
    assert lib.occ_pool8(ctypes.c_void_p(16), 4, 100, ctypes.c_void_p(16), None) == 1
    st = nat.OccPpoState()
    assert lib.occ_ppo_update(None, None, None, None, 8, 0.36, 0.2, 3e-4, 1e-3, 0.9, 0.999, 1e-8, ctypes.byref(st), 1, None, None,
                              None, None) == 1
    p16 = ctypes.c_void_p(16)
    assert lib.occ_ppo_update(p16, p16, p16, p16, 8, 0.0, 0.2, 3e-4, 1e-3, 0.9, 0.999, 1e-8, ctypes.byref(st), 1, p16, p16, p16,
                              None) == 1  # variance 0 / state pointers unset
    assert (nat.PPO_FEATURES, nat.PPO_PARAMS) == (256, 771)
