"""ctypes binding of liblbvh.so (include/lbvh.h; DEBUG_SIGNATURES: the test hooks and measurement aids of include/lbvh_debug.h).  No fallback: a missing or stale library is an
ImportError, a failing call is an LbvhError — the product path never computes on the CPU."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LBVH_LIB: an alternative build of the same library (tools/build_variant.sh: A/B measurements of kernel variants)
LIB_PATH = os.environ.get("LBVH_LIB") or os.path.join(_HERE, "liblbvh.so")
ABI_VERSION = 11

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C unitysimpleraytracing_amd/csrc` "
        "(or __graft_entry__.build()); there is no CPU fallback for the hot path")

lib = C.CDLL(LIB_PATH)


class LbvhError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"lbvh status {status}: {message}")
        self.status = status


class Camera(C.Structure):
    _fields_ = [("screen_width", C.c_int32), ("screen_height", C.c_int32),
                ("camera_fov", C.c_float), ("near_plane", C.c_float),
                ("camera_to_world", C.c_float * 16)]

    @classmethod
    def from_dict(cls, d):
        cam = cls()
        cam.screen_width = d["screen_width"]
        cam.screen_height = d["screen_height"]
        cam.camera_fov = d["camera_fov"]
        cam.near_plane = d["near_plane"]
        for i, v in enumerate(np.asarray(d["camera_to_world"], dtype=np.float32).reshape(-1)):
            cam.camera_to_world[i] = float(v)
        return cam


class ProfileRow(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint32), ("total_ms", C.c_float)]


class Scene(C.Structure):
    _fields_ = [("n", C.c_uint32), ("sorted_indices", C.c_void_p), ("triangle_aabb", C.c_void_p),
                ("internal_nodes", C.c_void_p), ("leaf_nodes", C.c_void_p), ("bvh", C.c_void_p),
                ("triangles", C.c_void_p)]


# every entry point include/lbvh.h declares: (name, restype, argtypes)
_P = C.c_void_p
_U32 = C.c_uint32
_I32 = C.c_int32
_SZ = C.c_size_t
_F3 = C.POINTER(C.c_float)
SIGNATURES = {
    "lbvh_abi_version": (_I32, []),
    "lbvh_device_count": (_I32, []),
    "lbvh_create": (_I32, [_I32, C.POINTER(_P)]),
    "lbvh_create_on_stream": (_I32, [_I32, _P, C.POINTER(_P)]),
    "lbvh_destroy": (_I32, [_P]),
    "lbvh_last_error": (C.c_char_p, [_P]),
    "lbvh_sync": (_I32, [_P]),
    "lbvh_buffer_alloc": (_I32, [_P, _SZ, _SZ, C.POINTER(_P)]),
    "lbvh_buffer_free": (_I32, [_P, _P]),
    "lbvh_buffer_fill_u32": (_I32, [_P, _P, _U32, _SZ]),
    "lbvh_buffer_upload": (_I32, [_P, _P, _P, _SZ]),
    "lbvh_buffer_download": (_I32, [_P, _P, _P, _SZ]),
    "lbvh_morton_aabb": (_I32, [_P, _P, _U32, _U32, _F3, _F3, _P, _P, _P]),
    "lbvh_sort_pairs": (_I32, [_P, _P, _P, _U32]),
    "lbvh_distribute_keys": (_I32, [_P, _P, _U32]),
    "lbvh_build_tree": (_I32, [_P, _U32, _P, _P, _P]),
    "lbvh_refit": (_I32, [_P, _U32, _P, _P, _P, _P, _P]),
    "lbvh_build_fast_scene": (_I32, [_P, C.POINTER(Scene), _F3, _F3]),
    "lbvh_trace_primary": (_I32, [_P, C.POINTER(Camera), _I32, _I32, _I32, _I32, C.POINTER(Scene),
                                  _I32, _P, _P]),
    "lbvh_trace_primary_shard": (_I32, [_P, C.POINTER(Camera), _U32, _U32, C.POINTER(Scene), _I32, _P, _P]),
    "lbvh_build_scene": (_I32, [_P, _P, _U32, _U32, _F3, _F3, _P, _P, _P, _P, _P, _P, _U32]),
    "lbvh_key_histogram": (_I32, [_P, _P, _U32, _P, _U32, _U32, _U32, _P]),
    "lbvh_lower_bound": (_I32, [_P, _P, _U32, _P, _U32, _P]),
    "lbvh_trace_costs_export": (_I32, [_P, _P, _U32, _U32]),
    "lbvh_trace_costs_import": (_I32, [_P, _P, _U32, _U32]),
    "lbvh_key_histogram_device": (_I32, [_P, _P, _U32, _P, _U32, _U32, _U32, _P]),
    "lbvh_lower_bound_device": (_I32, [_P, _P, _U32, _P, _U32, _P]),
    "lbvh_animate": (_I32, [_P, _P, _U32, _P, _P, C.c_float, C.c_float, _P]),
    "lbvh_animate_build_scene": (_I32, [_P, _P, _P, _P, C.c_float, C.c_float, _P, _U32, _U32, _F3, _F3, _P, _P, _P, _P, _P, _P, _U32]),
    "lbvh_trace_rays": (_I32, [_P, _P, _SZ, C.c_float, C.POINTER(Scene), _P]),
    "lbvh_path_begin": (_I32, [_P, C.POINTER(Camera), _P]),
    "lbvh_path_scatter": (_I32, [_P, C.POINTER(Scene), _P, _SZ, _U32, _U32, C.c_float, _P]),
    "lbvh_path_bounce": (_I32, [_P, C.POINTER(Scene), _P, _P, _SZ, _U32, _U32, C.c_float, C.c_float]),
    "lbvh_path_first_bounce": (_I32, [_P, C.POINTER(Camera), C.POINTER(Scene), _P, _P, _U32, C.c_float, C.c_float]),
    "lbvh_path_resolve": (_I32, [_P, _P, _SZ, _P]),
    "lbvh_trace_forget": (_I32, [_P]),
    "lbvh_peer_enable": (_I32, [_P, _I32]),
    "lbvh_sync_event_create": (_I32, [_P, C.POINTER(_P)]),
    "lbvh_event_wait": (_I32, [_P, _P]),
    "lbvh_ipc_export": (_I32, [_P, _P, C.POINTER(C.c_uint8)]),
    "lbvh_ipc_import": (_I32, [_P, C.POINTER(C.c_uint8), C.POINTER(_P)]),
    "lbvh_ipc_close": (_I32, [_P, _P]),
    "lbvh_flags_alloc": (_I32, [_P, _SZ, C.POINTER(_P)]),
    "lbvh_frame_signal": (_I32, [_P, _P, _U32, _U32]),
    "lbvh_frame_wait": (_I32, [_P, _P, _U32, _U32]),
    "lbvh_trace_primary_shard_packed": (_I32, [_P, C.POINTER(Camera), _U32, _U32, C.POINTER(Scene), _I32, _P, _P]),
    "lbvh_shard_records": (C.c_uint64, [_I32, _I32, _U32, _U32]),
    "lbvh_frame_unpack": (_I32, [_P, _P, C.c_uint64, _U32, _U32, _U32, _I32, _I32, _P]),
    "lbvh_shade": (_I32, [_P, _P, _SZ, _P, _P, _I32, _I32, _P]),
    "lbvh_compose": (_I32, [_P, _P, _P, _SZ, _P]),
    "lbvh_event_create": (_I32, [_P, C.POINTER(_P)]),
    "lbvh_event_destroy": (_I32, [_P, _P]),
    "lbvh_event_record": (_I32, [_P, _P]),
    "lbvh_event_elapsed_ms": (_I32, [_P, _P, _P, C.POINTER(C.c_float)]),
}

# include/lbvh_debug.h: not part of the drop-in boundary
DEBUG_SWITCH_SORT_QUEUES, DEBUG_SWITCH_COLD_ORDER, DEBUG_SWITCH_BUILD_FORM, DEBUG_SWITCH_FRAME_WAIT_MS, DEBUG_SWITCH_SORT_FORM, DEBUG_SWITCH_FAIL_RESERVE = range(6)
DEBUG_SIGNATURES = {
    "lbvh_debug_switch": (_I32, [_P, _U32, _U32]),
    "lbvh_debug_sort_ticket_tile": (_U32, [_U32, _U32, _U32, _U32]),
    "lbvh_debug_ray_stack_split": (_I32, [_P, _U32]),
    "lbvh_debug_ray_walker": (_I32, [_P, _U32]),
    "lbvh_debug_ray_stack_limit": (_I32, [_P, _U32]),
    "lbvh_ray_stats_target": (_I32, [_P, _P]),
    "lbvh_clock_probe": (_I32, [_P, C.POINTER(C.c_float)]),
    "lbvh_trace_tile_costs": (_I32, [_P, C.POINTER(Camera), C.POINTER(Scene), _P, _P, _P]),
    "lbvh_profile_begin": (_I32, [_P]),
    "lbvh_profile_end": (_I32, [_P, C.POINTER(ProfileRow), _I32, C.POINTER(_I32)]),
    "lbvh_copy_bandwidth_probe": (_I32, [_P, _P, _P, _SZ]),
}

for _name, (_res, _args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
    _fn = getattr(lib, _name)          # AttributeError here = library/header mismatch
    _fn.restype = _res
    _fn.argtypes = _args

if lib.lbvh_abi_version() != ABI_VERSION:
    raise ImportError(f"liblbvh.so ABI {lib.lbvh_abi_version()} != binding ABI {ABI_VERSION}: rebuild")


def check(ctx, status):
    if status != 0:
        msg = lib.lbvh_last_error(ctx)
        raise LbvhError(status, msg.decode() if msg else "")
    return status
