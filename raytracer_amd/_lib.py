"""ctypes binding of librt_mi355x.so (the C ABI declared in include/rt_mi355x.h).

There is no fallback: if the library is missing or no HIP device is usable, the calls raise.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

HERE = Path(__file__).resolve().parent
# RT_MI355X_LIB selects another build of the same library (kernel A/B experiments)
LIB_PATH = Path(os.environ.get("RT_MI355X_LIB", HERE / "librt_mi355x.so"))

# every symbol include/rt_mi355x.h declares
SYMBOLS = (
    "rt_create", "rt_destroy", "rt_last_error", "rt_resize", "rt_upload", "rt_set_scene_bounds", "rt_set_camera",
    "rt_render", "rt_reset_accumulation", "rt_rendered_frames", "rt_read_sample", "rt_read_accumulation",
    "rt_read_aux", "rt_get_stats", "rt_reset_stats", "rt_sync", "rt_render_sum", "rt_sum_buffer", "rt_clear_sum",
    "rt_finalize_sum", "rt_render_noalloc", "rt_comm_unique_id", "rt_comm_init", "rt_comm_reduce_sum",
    "rt_comm_all_reduce_f64", "rt_comm_barrier", "rt_set_fast_mode", "rt_fast_bvh_size", "rt_fast_bvh_export",
    "rt_traversal_variant",
    "rt_map_image", "rt_set_rendered_frames", "rt_set_traversal_variant", "rt_device_info", "rt_shader_clock", "rt_selftest",
    "rt_comm_info",
)
RT_COMM_ID_BYTES = 128
RT_COMM_SUM, RT_COMM_MAX = 0, 1

RT_FLAG_AUX = 1
RT_FLAG_STATS = 2


class RtStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "pixel_samples", "nodes", "tri_tests", "sphere_tests", "box_tests",
                                           "max_stack", "stack_overflows", "hits")] + [("last_render_ms", C.c_double),
                                                                               ("last_trace_ms", C.c_double)]


class RtSceneView(C.Structure):
    """rt_scene_view: device pointers to reference-layout buffers plus the host uniform values."""
    _fields_ = [("spheres", C.c_void_p), ("n_spheres", C.c_uint32), ("planes", C.c_void_p), ("n_planes", C.c_uint32),
                ("boxes", C.c_void_p), ("n_boxes", C.c_uint32), ("materials", C.c_void_p), ("n_materials", C.c_uint32),
                ("bvh", C.c_void_p), ("n_nodes", C.c_uint32), ("triangles", C.c_void_p), ("n_triangles", C.c_uint32),
                ("scene_min", C.c_float * 3), ("scene_extent", C.c_float * 3), ("inv_view_proj", C.c_float * 16),
                ("eye", C.c_float * 3)]


# capacity of rt_render_noalloc's fixed device array (RT_NOALLOC_MAX_* in the header)
NOALLOC_MAX = {"nodes": 32768, "triangles": 65536, "spheres": 4096, "boxes": 4096, "planes": 256, "materials": 4096}

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` (hipcc, gfx950). "
            "There is no CPU fallback for the render path."
        )
    L = C.CDLL(str(LIB_PATH))
    vp = C.c_void_p
    L.rt_last_error.restype = C.c_char_p
    L.rt_last_error.argtypes = []
    sig = {
        "rt_create": [C.c_int, C.POINTER(vp)],
        "rt_destroy": [vp],
        "rt_resize": [vp, C.c_int, C.c_int],
        "rt_upload": [vp, C.c_int, vp, C.c_size_t],
        "rt_set_scene_bounds": [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)],
        "rt_set_camera": [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)],
        "rt_render": [vp, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32],
        "rt_reset_accumulation": [vp],
        "rt_rendered_frames": [vp, C.POINTER(C.c_uint32)],
        "rt_read_sample": [vp, vp],
        "rt_read_accumulation": [vp, vp],
        "rt_read_aux": [vp, vp],
        "rt_get_stats": [vp, C.POINTER(RtStats)],
        "rt_reset_stats": [vp],
        "rt_sync": [vp],
        "rt_render_sum": [vp, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp],
        "rt_sum_buffer": [vp, C.POINTER(vp), C.POINTER(C.c_size_t)],
        "rt_clear_sum": [vp, vp],
        "rt_finalize_sum": [vp, vp, C.c_uint32],
        "rt_set_fast_mode": [vp, C.c_int],
        "rt_fast_bvh_size": [vp, C.POINTER(C.c_uint32)],
        "rt_fast_bvh_export": [vp, vp, C.c_size_t],
        "rt_traversal_variant": [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "rt_map_image": [vp, C.c_int, C.POINTER(C.POINTER(C.c_float))],
        "rt_comm_unique_id": [C.c_char_p],
        "rt_comm_init": [vp, C.c_int, C.c_int, C.c_char_p],
        "rt_comm_reduce_sum": [vp, vp, C.c_int],
        "rt_comm_all_reduce_f64": [vp, C.POINTER(C.c_double), C.c_int, C.c_int],
        "rt_comm_barrier": [vp],
        "rt_comm_info": [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "rt_set_rendered_frames": [vp, C.c_uint32],
        "rt_set_traversal_variant": [vp, C.c_int],
        "rt_device_info": [vp, C.POINTER(C.c_int), C.c_char_p, C.c_size_t],
        "rt_shader_clock": [vp, C.POINTER(C.c_double)],
        "rt_selftest": [vp, C.POINTER(C.c_ulonglong)],
        "rt_render_noalloc": [C.c_int, C.POINTER(RtSceneView), C.c_int, C.c_int, C.c_uint32, C.c_float, vp, vp,
                              C.c_uint32, vp],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = args
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(load().rt_last_error().decode(errors="replace"))
