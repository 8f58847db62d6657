"""ctypes binding of the C ABI declared in include/occlusionenv_amd.h.

The product path has NO fallback: if ``libocc_hip.so`` is missing, or a call returns a nonzero
status, this module raises.  Nothing here imports ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OCC_HIP_LIB selects another build of the SAME sources (tests use a small-OCC_LOG_CAP build); never a fallback
LIB_PATH = os.environ.get("OCC_HIP_LIB") or os.path.join(_HERE, "libocc_hip.so")

# layout constants (must match include/occlusionenv_amd.h)
ABI_VERSION = 9
CAM_STRIDE = 48
REC_STRIDE = 32
TILE = 8
LOG_CAP = 12288
LOG_ENTRY_BYTES = 30
MAX_K = 128
CAM_STEP, CAM_LOOKAT, CAM_POSITION = 0, 1, 2
RENDER_SOFT, RENDER_HARD, RENDER_GRAD = 1, 2, 4
STATUS_LIST_OVERFLOW, STATUS_REC_OVERFLOW = 1, 2
SHADER_FLAT, SHADER_HARD_PHONG, SHADER_SOFT_PHONG = 0, 1, 2

# camera buffer slots
C_R, C_T, C_C, C_J, C_EL, C_AZ = 0, 9, 12, 39, 43, 44


class OccScene(C.Structure):
    _fields_ = [
        ("pool_verts", C.c_void_p),
        ("pool_faces", C.c_void_p),
        ("mesh_vert_off", C.c_void_p),
        ("mesh_face_off", C.c_void_p),
        ("scene_mesh", C.c_void_p),
        ("scene_offset", C.c_void_p),
        ("n_meshes", C.c_int32),
        ("n_env", C.c_int32),
        ("img", C.c_int32),
        ("rec_cap", C.c_int32),
        ("pool_atlas", C.c_void_p),
        ("mesh_atlas_off", C.c_void_p),
        ("atlas_res", C.c_int32),
        ("skip", C.c_void_p),
        ("pix_weight", C.c_void_p),
        ("shader", C.c_int32),
        ("pool_vnormals", C.c_void_p),
        ("max_mesh_verts", C.c_int32),
    ]


class OccWorkspace(C.Structure):
    _fields_ = [
        ("rec", C.c_void_p),
        ("rec_bbox", C.c_void_p),
        ("scan", C.c_void_p),
        ("nrec", C.c_void_p),
        ("objrect", C.c_void_p),
        ("queue", C.c_void_p),
        ("lists", C.c_void_p),
        ("partials", C.c_void_p),
        ("status", C.c_void_p),
        ("offsets", C.c_void_p),
        ("rec_cbox", C.c_void_p),
        ("obj_alpha", C.c_void_p),
        ("obj_grad", C.c_void_p),
        ("obj_hz", C.c_void_p),
        ("obj_hrec", C.c_void_p),
        ("n_slots", C.c_int32),
        ("rec_off", C.c_void_p),
        ("rec_total", C.c_int64),
        ("order", C.c_void_p),
    ]


class OccWorkspaceSizes(C.Structure):
    _fields_ = [
        ("rec_bytes", C.c_size_t),
        ("rec_bbox_bytes", C.c_size_t),
        ("nrec_bytes", C.c_size_t),
        ("objrect_bytes", C.c_size_t),
        ("queue_bytes", C.c_size_t),
        ("lists_bytes", C.c_size_t),
        ("partials_bytes", C.c_size_t),
        ("status_bytes", C.c_size_t),
        ("offsets_bytes", C.c_size_t),
        ("obj_alpha_bytes", C.c_size_t),
        ("obj_grad_bytes", C.c_size_t),
        ("obj_hz_bytes", C.c_size_t),
        ("obj_hrec_bytes", C.c_size_t),
        ("rec_cbox_bytes", C.c_size_t),
        ("scan_bytes", C.c_size_t),
        ("n_slots", C.c_int32),
        ("rec_off_bytes", C.c_size_t),
        ("order_bytes", C.c_size_t),
    ]


class OccStepFinish(C.Structure):
    """The reward bookkeeping of occ_step_finish done by the launch that reduces the loss (OccRenderOut.finish)."""
    _fields_ = [(n, C.c_void_p) for n in ("full_reward", "object_mass", "reward", "done", "grad_action")] + [("n_step", C.c_int32)]


class OccRenderOut(C.Structure):
    _fields_ = [
        ("obs", C.c_void_p),
        ("full_state", C.c_void_p),
        ("alphas", C.c_void_p),
        ("loss", C.c_void_p),
        ("grad_elaz", C.c_void_p),
        # region tracking of persistent outputs (include/occlusionenv_amd.h)
        ("rect_prev", C.c_void_p),
        ("rect_next", C.c_void_p),
        ("arect_prev", C.c_void_p),
        ("arect_next", C.c_void_p),
        ("finish", C.POINTER(OccStepFinish)),
    ]


class OccCameraArgs(C.Structure):
    _fields_ = [("mode", C.c_int32), ("action", C.c_void_p), ("el", C.c_void_p), ("az", C.c_void_p), ("radius", C.c_void_p),
                ("cam_pos_out", C.c_void_p), ("cam_pos_out2", C.c_void_p), ("n", C.c_int32)]


class OccAutoResetOpts(C.Structure):
    _fields_ = [("age", C.c_void_p), ("max_ep_len", C.c_int32), ("rect", C.c_void_p), ("arect", C.c_void_p),
                ("reset_full_state", C.c_void_p), ("norm_flags", C.c_void_p), ("slot_objsum", C.c_void_p),
                ("report_host", C.c_void_p)]


class OccEnvState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("el", "az", "radius", "campos", "cam", "alphas", "full_reward", "object_mass",
                                          "scene_mesh", "scene_offset")]


class OccPpoState(C.Structure):
    """include/occlusionenv_amd.h: OccPpoState (parameters, Adam moments and step count of the heads-only learner)."""
    _fields_ = [(n, C.c_void_p) for n in ("w_a", "b_a", "w_v", "b_v", "adam_m", "adam_v", "adam_step")]


PPO_FEATURES = 256
PPO_PARAMS = 3 * PPO_FEATURES + 3
PPO_SCRATCH_FLOATS = 128 * (PPO_PARAMS + 2)  # for the default OCC_PPO_MAX_BLOCKS; ppo_scratch_floats() asks the library


class OccReserveStore(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("obs", "full_state", "loss", "skip")]


RS_EMPTY, RS_PENDING, RS_READY = 0, 1, 2

#: every symbol include/occlusionenv_amd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "occ_abi_version": (C.c_int, []),
    "occ_device_cu_count": (C.c_int, []),
    "occ_workspace_query": (C.c_int, [C.POINTER(OccScene), C.c_int, C.POINTER(OccWorkspaceSizes)]),
    "occ_record_sizes": (C.c_int, [C.c_int64, C.c_int, C.POINTER(OccWorkspaceSizes)]),
    "occ_camera": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_int, C.c_void_p]),
    "occ_render": (C.c_int, [C.POINTER(OccScene), C.c_void_p, C.POINTER(OccWorkspace), C.POINTER(OccRenderOut),
                             C.c_int, C.c_int, C.c_void_p]),
    "occ_step": (C.c_int, [C.POINTER(OccScene), C.POINTER(OccCameraArgs), C.c_void_p, C.POINTER(OccWorkspace),
                           C.POINTER(OccRenderOut), C.c_int, C.c_int, C.c_void_p]),
    "occ_step_finish": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_void_p]),
    "occ_rasterize_meshes_naive": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                             C.c_int, C.c_int] + [C.c_void_p] * 5),
    "occ_rasterize_meshes_tiled": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                             C.c_int, C.c_int] + [C.c_void_p] * 5),
    "occ_rasterize_meshes_backward_dists": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                                      C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "occ_rasterize_meshes_backward": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_int, C.c_void_p, C.c_void_p]),
    "occ_sigmoid_alpha_blend_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "occ_sigmoid_alpha_blend_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p,
                                              C.c_void_p]),
    "occ_ppo_update": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] + [C.c_float] * 7 + [C.POINTER(OccPpoState), C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "occ_ppo_max_blocks": (C.c_int, []),
    "occ_pool8": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "occ_step_flags": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "occ_reset_commit": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 13 + [C.c_int, C.c_void_p]),
    "occ_auto_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.POINTER(OccEnvState), C.c_void_p, C.c_void_p, C.POINTER(OccReserveStore), C.c_void_p,
                                 C.c_int, C.c_void_p, C.c_void_p, C.POINTER(OccAutoResetOpts), C.c_void_p]),
    "occ_object_mass": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "occ_reserve_refill": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "occ_profile_enable": (C.c_int, [C.c_int]),
    "occ_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int)]),
}


class NativeError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    """Load the HIP extension; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} not found: the HIP extension is not built.  Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` at the repo root (needs hipcc). "
            "There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.occ_abi_version() != ABI_VERSION:
        raise NativeError(f"ABI mismatch: library {lib.occ_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def ppo_scratch_floats() -> int:
    """Scratch floats occ_ppo_update needs, from the block cap the LIBRARY was built with (-DOCC_PPO_MAX_BLOCKS)."""
    return int(load().occ_ppo_max_blocks()) * (PPO_PARAMS + 2)


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise NativeError(f"{what} failed with status {rc} (1 = bad argument, 2 = launch failure)")
