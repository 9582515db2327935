"""rt_render_noalloc (the entry point that allocates nothing) against rt_render, through the C ABI.

Two calls in a row on the same GPU share the library's fixed device array; the second renders a
smaller scene at a smaller size, so anything stale from the first call would show.  Bit-for-bit.
"""

import ctypes as C

import numpy as np
import pytest

from util import GOLDEN, load_golden_scene, render_frame

pytestmark = pytest.mark.gpu

_FILES = (("spheres", "spheres.bin", 32), ("planes", "planes.bin", 32), ("boxes", "boxes.bin", 32),
          ("materials", "mats.bin", 12), ("bvh", "bvh.bin", 64), ("triangles", "tris.bin", 64))
_COUNT = {"spheres": "n_spheres", "planes": "n_planes", "boxes": "n_boxes", "materials": "n_materials",
          "bvh": "n_nodes", "triangles": "n_triangles"}


class _Hip:
    """The handful of HIP runtime calls the test itself needs (the caller owns inputs and outputs)."""

    def __init__(self):
        self.h = C.CDLL("libamdhip64.so")
        self.h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.h.hipFree.argtypes = [C.c_void_p]
        self.h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.live = []

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.h.hipMalloc(C.byref(p), max(nbytes, 16)) == 0
        self.live.append(p)
        return p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        if arr.nbytes:
            assert self.h.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1) == 0
        return p

    def zeros(self, nbytes):
        p = self.alloc(nbytes)
        assert self.h.hipMemset(p, 0, nbytes) == 0
        return p

    def download(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        assert self.h.hipDeviceSynchronize() == 0
        assert self.h.hipMemcpy(out.ctypes.data, p, out.nbytes, 2) == 0
        return out

    def free_all(self):
        self.h.hipDeviceSynchronize()
        for p in self.live:
            self.h.hipFree(p)
        self.live = []


def _scene_view(hip, scene):
    from raytracer_amd import _lib

    d = GOLDEN / scene
    v = _lib.RtSceneView()
    for field, name, stride in _FILES:
        data = np.fromfile(d / name, dtype=np.uint8) if (d / name).exists() else np.zeros(0, np.uint8)
        setattr(v, field, hip.upload(data))
        setattr(v, _COUNT[field], data.size // stride)
    u = np.fromfile(d / "uniforms.bin", dtype=np.float32)
    v.inv_view_proj[:] = u[0:16].tolist()
    v.eye[:] = u[16:19].tolist()
    v.scene_min[:] = u[19:22].tolist()
    v.scene_extent[:] = u[22:25].tolist()
    return v


def test_noalloc_matches_rt_render_twice_in_a_row():
    from raytracer_amd import _lib
    from raytracer_amd.renderer import Renderer

    L = _lib.load()
    hip = _Hip()
    r = Renderer(device=0, singleton=False)
    try:
        # first call: the larger scene (830 nodes, 1608 triangles, 30 spheres), two frames with the
        # running mean; second call: a two-triangle scene at a smaller size, right after
        for scene, w, h, bounces, frames in (("gally", 320, 180, 4, (0, 1)), ("debug", 160, 88, 3, (5,))):
            load_golden_scene(r, scene)
            r.update_size(w, h)
            r.reset_accumulation()
            view = _scene_view(hip, scene)
            sample = hip.zeros(w * h * 16)
            acc = hip.zeros(w * h * 16)
            for k, f in enumerate(frames):
                want_sample = render_frame(r, w, h, f, bounces)
                want_acc = r.read_accumulation()
                _lib.check(L.rt_render_noalloc(0, C.byref(view), w, h, f, float(bounces), sample, acc, k, None))
                got_sample = hip.download(sample, (h, w, 4), np.float32)
                got_acc = hip.download(acc, (h, w, 4), np.float32)
                assert np.array_equal(got_sample.view(np.uint32), want_sample.view(np.uint32)), (scene, f, "sample")
                assert np.array_equal(got_acc.view(np.uint32), want_acc.view(np.uint32)), (scene, f, "running mean")
                assert got_sample[..., :3].max() > 0
        # beyond the fixed capacity: refused with a message, nothing launched
        view.n_nodes = _lib.NOALLOC_MAX["nodes"] + 1
        assert L.rt_render_noalloc(0, C.byref(view), 160, 88, 0, 3.0, sample, None, 0, None) != 0
        msg = L.rt_last_error().decode()
        assert "bvh" in msg and "capacity" in msg
        assert L.rt_render_noalloc(0, C.byref(view), 0, 88, 0, 3.0, sample, None, 0, None) != 0
        assert L.rt_render_noalloc(0, None, 160, 88, 0, 3.0, sample, None, 0, None) != 0
    finally:
        hip.free_all()
        r.close()
