"""Renderer: drop-in for the reference ``Renderer`` (reference src/renderer/renderer.py).

Same public surface — ``get_instance()``, ``render()``, ``update_ssbo(ssbo, elements, index)``,
``reset_accumulation()``, ``update_size(w, h)``, the ``bounces`` property, the buffer-handle
attributes (``spheres``, ``planes``, ``boxes``, ``materials``, ``bvh``, ``triangles``),
``scene_min_coords`` / ``scene_extent``, ``render_time`` and ``rendered_frames`` — but the work of
``render()`` is done by hand-written HIP kernels on an MI355X through the C ABI in
``include/rt_mi355x.h`` instead of an OpenGL compute dispatch.  The images live in GPU memory
and are read back with ``read_sample()`` / ``read_accumulation()`` (what the reference's screen
and upscale passes would sample) or dumped with ``save_png()``.
"""

from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import _lib
from .camera import Camera


def is_positive_int(value) -> int:
    """reference src/utils/typecheck.py:100-144 (floats are truncated, negatives rejected)."""
    if isinstance(value, bool) or not isinstance(value, (int, float)):
        raise TypeError("Value must be int")
    v = int(value)
    if v < 0:
        raise TypeError("Value must be a positive int")
    return v


class RendererInitializedError(Exception):
    """reference renderer.py:467-489"""


# binding index of each buffer-handle attribute (reference scene.py:62,97,122,193)
_BINDINGS = (5, 6, 7, 9, 11, 12)


class Renderer:
    __instance: "Renderer | None" = None

    def __init__(self, device: int = 0, camera: Camera | None = None, singleton: bool = True) -> None:
        if singleton:
            if Renderer.__instance is not None:
                raise RendererInitializedError("Renderer already exists!")
            Renderer.__instance = self
        self._L = _lib.load()
        self._ctx = C.c_void_p()
        _lib.check(self._L.rt_create(int(device), C.byref(self._ctx)))
        self._width = 1280   # renderer.py:75-76
        self._height = 720
        self._frames = 0
        self.scene_min_coords = []
        self.scene_extent = []
        # opaque handles, as glGenBuffers would hand out (renderer.py:99-104)
        self.spheres, self.planes, self.boxes, self.materials, self.bvh, self.triangles = 1, 2, 3, 4, 5, 6
        self._camera = camera if camera is not None else Camera.get_instance()
        self.render_time = 0.0
        self.rendered_frames = 0
        self._bounces = 10   # renderer.py:112
        self._flags = 0

    # -- reference surface -----------------------------------------------------------------
    @staticmethod
    def get_instance() -> "Renderer":
        if Renderer.__instance is None:
            Renderer()
        return Renderer.__instance

    @property
    def camera(self) -> Camera:
        return self._camera

    @property
    def width(self) -> int:
        return self._width

    @property
    def height(self) -> int:
        return self._height

    @property
    def bounces(self) -> int:
        return self._bounces

    @bounces.setter
    def bounces(self, value) -> None:
        self._bounces = is_positive_int(value)

    def update_ssbo(self, ssbo: int, elements, index: int, unbind: bool = True) -> None:
        """Copies `elements` (reference std430 layout) into the buffer bound at `index`."""
        if index not in _BINDINGS:
            raise ValueError(f"unknown SSBO binding {index}")
        arr = np.ascontiguousarray(elements)
        _lib.check(self._L.rt_upload(self._ctx, int(index), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def update_size(self, width: int, height: int) -> None:
        """reference renderer.py:347-382: the three images are recreated (black); `rendered_frames`
        is left alone, so the next frame is blended into black with weight 1/(rendered_frames+1)."""
        self._width = is_positive_int(width)
        self._height = is_positive_int(height)
        _lib.check(self._L.rt_resize(self._ctx, self._width, self._height))

    def reset_accumulation(self) -> None:
        _lib.check(self._L.rt_reset_accumulation(self._ctx))
        self.rendered_frames = 0

    def render(self, frames: int = 1) -> None:
        """One reference frame (renderer.py:164-306): sample at time=_frames, then accumulate.

        `frames` > 1 queues that many consecutive frames in one call."""
        start = time.time()
        if self._camera.changed:
            self._camera.changed = False
            self.reset_accumulation()
        self._push_uniforms()
        # `rendered_frames` is a plain attribute in the reference and goes into the accumulate pass as
        # it stands (renderer.py:255-257): whatever the host assigned is what the library starts from
        _lib.check(self._L.rt_set_rendered_frames(self._ctx, is_positive_int(self.rendered_frames)))
        _lib.check(self._L.rt_render(self._ctx, self._frames, int(frames), float(self._bounces), self._flags))
        self._frames += int(frames)
        self.rendered_frames += int(frames)
        self.render_time = time.time() - start

    # -- additions (device-side images are not GL textures) ---------------------------------
    def set_frame_index(self, frame: int) -> None:
        """The value the next render() passes as the `time` uniform."""
        self._frames = is_positive_int(frame)

    def set_uniforms(self, inv_view_proj, eye) -> None:
        """Bypass the camera object and send these uniform values on the next render()."""
        self._override = (np.asarray(inv_view_proj, dtype=np.float32).reshape(16).copy(),
                          np.asarray(eye, dtype=np.float32).reshape(3).copy())

    _override = None

    def _push_uniforms(self) -> None:
        if self._override is not None:
            m, eye = self._override
        else:
            m, eye = self._camera.uniforms_f32()
        fp = C.POINTER(C.c_float)
        _lib.check(self._L.rt_set_camera(self._ctx, m.ctypes.data_as(fp), eye.ctypes.data_as(fp)))
        if len(self.scene_min_coords) == 3 and len(self.scene_extent) == 3:
            mn = np.asarray(self.scene_min_coords, dtype=np.float32).reshape(3).copy()
            ex = np.asarray(self.scene_extent, dtype=np.float32).reshape(3).copy()
            _lib.check(self._L.rt_set_scene_bounds(self._ctx, mn.ctypes.data_as(fp), ex.ctypes.data_as(fp)))

    @property
    def fast_mode(self) -> bool:
        """Opt-in, default False: traverse the library's own tree (all-axis SAH, conservative 8-bit
        boxes in 64-byte nodes) instead of the reference's.  Not the reference's data: results
        differ where the reference's tree decides which hit is found."""
        return self._fast

    @fast_mode.setter
    def fast_mode(self, on: bool) -> None:
        _lib.check(self._L.rt_set_fast_mode(self._ctx, 1 if on else 0))
        self._fast = bool(on)

    _fast = False

    def traversal_variant(self) -> tuple[str, bool]:
        """Which traversal variant the default pipeline uses for this scene ("unified" or
        "leaf steps") and whether the timed frames that decide it have run."""
        ls, dec = C.c_int(), C.c_int()
        _lib.check(self._L.rt_traversal_variant(self._ctx, C.byref(ls), C.byref(dec)))
        return ("leaf steps" if ls.value else "unified"), bool(dec.value)

    def set_traversal_variant(self, variant: str | None) -> None:
        """Fixes the traversal variant ("unified" or "leaf steps") for this renderer, which also keeps
        the library from timing the two with four extra launches at the 65th frame of a scene; None
        returns to the automatic choice."""
        v = {None: -1, "unified": 0, "leaf steps": 1}[variant]
        _lib.check(self._L.rt_set_traversal_variant(self._ctx, v))

    def device_info(self) -> tuple[int, str]:
        """(HIP device ordinal, PCI bus id) of the GPU this renderer runs on."""
        dev = C.c_int()
        buf = C.create_string_buffer(32)
        _lib.check(self._L.rt_device_info(self._ctx, C.byref(dev), buf, 32))
        return dev.value, buf.value.decode()

    def shader_clock_ghz(self) -> float:
        """Shader clock the trace kernel's wavefronts ran at since the last reset_stats(), as the
        kernel measured it (0.0 if no launch sampled it)."""
        g = C.c_double()
        _lib.check(self._L.rt_shader_clock(self._ctx, C.byref(g)))
        return g.value

    def selftest(self) -> int:
        """Number of floats for which the kernels' short reciprocal differs from the division it
        stands for, checked on the device over every float; must be 0."""
        n = C.c_ulonglong()
        _lib.check(self._L.rt_selftest(self._ctx, C.byref(n)))
        return int(n.value)

    def fast_bvh(self) -> np.ndarray:
        """The fast-mode tree as built for the current scene: (n + 1, 16) uint32, the last node a sentinel."""
        nn = C.c_uint32()
        _lib.check(self._L.rt_fast_bvh_size(self._ctx, C.byref(nn)))
        nodes = np.zeros((nn.value + 1, 16), np.uint32)
        _lib.check(self._L.rt_fast_bvh_export(self._ctx, nodes.ctypes.data_as(C.c_void_p), nodes.nbytes))
        return nodes

    def set_flags(self, aux: bool = False, stats: bool = False) -> None:
        self._flags = (_lib.RT_FLAG_AUX if aux else 0) | (_lib.RT_FLAG_STATS if stats else 0)

    def sync(self) -> None:
        _lib.check(self._L.rt_sync(self._ctx))

    def _read(self, fn, dtype, chans) -> np.ndarray:
        out = np.empty((self._height, self._width, chans), dtype=dtype)
        _lib.check(fn(self._ctx, out.ctypes.data_as(C.c_void_p)))
        return out

    def read_sample(self) -> np.ndarray:
        """(H, W, 4) float32, row 0 = bottom: what the compute pass stored for the last frame."""
        return self._read(self._L.rt_read_sample, np.float32, 4)

    def read_accumulation(self) -> np.ndarray:
        """(H, W, 4) float32, row 0 = bottom: the running mean, sRGB-encoded."""
        return self._read(self._L.rt_read_accumulation, np.float32, 4)

    def map_image(self, accumulation: bool = True) -> np.ndarray:
        """(H, W, 4) float32 read-only view of the accumulation image (or the last sample) in the
        library's own page-locked buffer: no copy into Python memory, transfer at the PCIe rate.
        Valid, and unchanged, until the next map_image(), update_size() or close()."""
        p = C.POINTER(C.c_float)()
        _lib.check(self._L.rt_map_image(self._ctx, 1 if accumulation else 0, C.byref(p)))
        a = np.ctypeslib.as_array(p, shape=(self._height, self._width, 4))
        a.flags.writeable = False
        return a

    def read_aux(self) -> np.ndarray:
        return self._read(self._L.rt_read_aux, np.uint32, 16)

    def stats(self) -> dict:
        s = _lib.RtStats()
        _lib.check(self._L.rt_get_stats(self._ctx, C.byref(s)))
        return {n: getattr(s, n) for n, _ in _lib.RtStats._fields_}

    def reset_stats(self) -> None:
        _lib.check(self._L.rt_reset_stats(self._ctx))

    # sharded linear-sum rendering (multi-GPU)
    def render_sum(self, first_frame, nframes, shard=0, nshards=1, frame_stride=1, frame_phase=0, sum_ptr=None):
        self._push_uniforms()
        _lib.check(self._L.rt_render_sum(self._ctx, int(first_frame), int(nframes), float(self._bounces), int(shard),
                                         int(nshards), int(frame_stride), int(frame_phase),
                                         C.c_void_p(sum_ptr) if sum_ptr else None))

    def clear_sum(self, sum_ptr=None) -> None:
        _lib.check(self._L.rt_clear_sum(self._ctx, C.c_void_p(sum_ptr) if sum_ptr else None))

    def finalize_sum(self, spp: int, sum_ptr=None) -> None:
        _lib.check(self._L.rt_finalize_sum(self._ctx, C.c_void_p(sum_ptr) if sum_ptr else None, int(spp)))

    def sum_buffer(self) -> tuple[int, int]:
        p = C.c_void_p()
        n = C.c_size_t()
        _lib.check(self._L.rt_sum_buffer(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def _hip_copy(self, dst, src, nbytes: int, kind: int) -> None:
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        if hip.hipMemcpy(dst, src, nbytes, kind) != 0:
            raise RuntimeError("hipMemcpy of the sum image failed")

    def read_sum(self) -> np.ndarray:
        """(H, W, 4) float32 copy of the context's linear sum image (rgb sums, alpha = sample count)."""
        self.sync()
        p, n = self.sum_buffer()
        out = np.empty((self._height, self._width, 4), np.float32)
        self._hip_copy(out.ctypes.data, C.c_void_p(p), n, 2)
        return out

    def write_sum(self, img) -> None:
        self.sync()
        p, n = self.sum_buffer()
        a = np.ascontiguousarray(img, dtype=np.float32)
        assert a.nbytes == n
        self._hip_copy(C.c_void_p(p), a.ctypes.data, n, 1)

    def close(self) -> None:
        if self._ctx:
            self._L.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()
        if Renderer.__instance is self:
            Renderer.__instance = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
