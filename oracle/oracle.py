"""TEST INFRASTRUCTURE — ctypes wrapper around oracle/librt_oracle.so (the CPU restatement).

Imported only by tests/, ``__graft_entry__.smoke()`` and the benchmark's cpu_baseline leg.
"""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_LIB = None


class RtoScene(C.Structure):
    _fields_ = [
        ("nodes", C.c_void_p), ("n_nodes", C.c_uint32),
        ("tris", C.c_void_p), ("n_tris", C.c_uint32),
        ("spheres", C.c_void_p), ("n_spheres", C.c_uint32),
        ("boxes", C.c_void_p), ("n_boxes", C.c_uint32),
        ("planes", C.c_void_p), ("n_planes", C.c_uint32),
        ("mats", C.c_void_p), ("n_mats", C.c_uint32),
        ("inv_vp", C.c_float * 16),
        ("eye", C.c_float * 3), ("scene_min", C.c_float * 3), ("scene_extent", C.c_float * 3),
    ]


class RtoCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("rays", "nodes", "tri_tests", "sphere_tests", "box_tests", "max_stack", "stack_over_32", "pad")]


def build() -> Path:
    so = HERE / "librt_oracle.so"
    src = HERE / "rt_oracle.c"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "port"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(str(build()))
        L.rto_render.restype = C.c_int
        L.rto_render.argtypes = [C.POINTER(RtoScene), C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p,
                                 C.c_void_p, C.POINTER(RtoCounters), C.c_int]
        L.rto_render_window.restype = C.c_int
        L.rto_render_window.argtypes = [C.POINTER(RtoScene), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_float, C.c_float, C.c_void_p, C.c_int]
        L.rto_accumulate.restype = None
        L.rto_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]
        for name in ("rto_sin", "rto_cos", "rto_exp2", "rto_log2", "rto_srgb_to_linear", "rto_linear_to_srgb"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [C.c_float]
        L.rto_pow.restype = C.c_float
        L.rto_pow.argtypes = [C.c_float, C.c_float]
        L.rto_wang_hash.restype = C.c_uint32
        L.rto_wang_hash.argtypes = [C.c_uint32]
        _LIB = L
    return _LIB


class OracleScene:
    """Holds the byte buffers (reference layouts) and the 25 uniform floats."""

    STRIDE = {"bvh": 64, "tris": 64, "spheres": 32, "boxes": 32, "planes": 32, "mats": 12}

    def __init__(self, bvh=b"", tris=b"", spheres=b"", mats=b"", uniforms=None, boxes=b"", planes=b""):
        self._keep = {}
        self.s = RtoScene()
        for field, key, data in (("nodes", "bvh", bvh), ("tris", "tris", tris), ("spheres", "spheres", spheres),
                                 ("boxes", "boxes", boxes), ("planes", "planes", planes), ("mats", "mats", mats)):
            arr = np.frombuffer(bytes(data), dtype=np.uint8).copy()
            self._keep[key] = arr
            n = arr.size // self.STRIDE[key]
            setattr(self.s, field, arr.ctypes.data if arr.size else None)
            setattr(self.s, "n_" + field, n)
        u = np.asarray(uniforms, dtype=np.float32).reshape(25)
        self.uniforms = u
        self.s.inv_vp[:] = u[0:16].tolist()
        self.s.eye[:] = u[16:19].tolist()
        self.s.scene_min[:] = u[19:22].tolist()
        self.s.scene_extent[:] = u[22:25].tolist()

    @classmethod
    def from_dir(cls, d: Path) -> "OracleScene":
        d = Path(d)

        def rd(n):
            p = d / n
            return p.read_bytes() if p.exists() else b""

        return cls(bvh=rd("bvh.bin"), tris=rd("tris.bin"), spheres=rd("spheres.bin"), mats=rd("mats.bin"),
                   boxes=rd("boxes.bin"), planes=rd("planes.bin"),
                   uniforms=np.fromfile(d / "uniforms.bin", dtype=np.float32))

    def render(self, w, h, frame, bounces, want_primary=False, want_counters=False, threads=0):
        img = np.zeros((h, w, 4), dtype=np.float32)
        prim = np.zeros((h, w, 16), dtype=np.uint32) if want_primary else None
        cnt = RtoCounters() if want_counters else None
        rc = lib().rto_render(C.byref(self.s), w, h, float(frame), float(bounces), img.ctypes.data,
                              prim.ctypes.data if prim is not None else None,
                              C.byref(cnt) if cnt is not None else None, threads)
        if rc != 0:
            raise RuntimeError(f"rto_render failed rc={rc}")
        out = [img]
        if want_primary:
            out.append(prim)
        if want_counters:
            out.append({n: getattr(cnt, n) for n, _ in RtoCounters._fields_ if n != "pad"})
        return out[0] if len(out) == 1 else tuple(out)


def set_fast_bvh(nodes=None):
    """Switches the oracle to the fast-mode tree (the (n + 1, 16) array from
    raytracer_amd.scene.build_fast_bvh or Renderer.fast_bvh, sentinel included), or back to the
    reference's tree with None.  The array is kept alive here while set."""
    global _FAST_KEEP
    L = lib()
    L.rto_set_fast_bvh.restype = None
    L.rto_set_fast_bvh.argtypes = [C.c_void_p, C.c_uint32]
    if nodes is None:
        L.rto_set_fast_bvh(None, 0)
        _FAST_KEEP = None
        return
    n = np.ascontiguousarray(nodes, dtype=np.uint32).reshape(-1, 16)
    _FAST_KEEP = n
    L.rto_set_fast_bvh(n.ctypes.data, n.shape[0] - 1)


_FAST_KEEP = None


def _render_window(self, w, h, x0, y0, ww, wh, frame, bounces, threads=0):
    """Pixels [x0, x0+ww) x [y0, y0+wh) of the w x h frame, as a dense (wh, ww, 4) image."""
    img = np.zeros((wh, ww, 4), dtype=np.float32)
    rc = lib().rto_render_window(C.byref(self.s), w, h, x0, y0, ww, wh, float(frame), float(bounces),
                                 img.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError(f"rto_render_window failed rc={rc}")
    return img


OracleScene.render_window = _render_window


def accumulate(sample: np.ndarray, old: np.ndarray, frames: float) -> np.ndarray:
    sample = np.ascontiguousarray(sample, dtype=np.float32)
    old = np.ascontiguousarray(old, dtype=np.float32)
    out = np.empty_like(sample)
    lib().rto_accumulate(sample.ctypes.data, old.ctypes.data, out.ctypes.data, sample.size // 4, float(frames))
    return out


def primary_from_golden(g: np.ndarray):
    """Decode the instrumented shader's image (oracle/refscene.py 'primary') into (code, t_bits, exists)."""
    y = g[..., 1].astype(np.uint32)
    z = g[..., 2].astype(np.uint32)
    typ = y >> 16
    idx = (y & 0xFFFF) | (z << 16)
    code = (typ << 30) | (idx & 0x3FFFFFFF)
    return code, g[..., 0].view(np.uint32), g[..., 3].astype(np.uint32)
