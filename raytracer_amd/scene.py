"""Scene loading: TOML + OBJ -> the byte buffers the render path consumes.

Mirrors the reference loader (reference src/scene/scene.py:16-71) and the packers it calls
(src/material/material.py:197-230, src/primitive/sphere/sphere.py:132-158,
src/primitive/triangle/triangle.py:152-188, src/utils/utils.py:113-127,164-194), vectorised with
numpy where the reference loops over Python objects, and with the BVH built by the native
builder (csrc/scene_builder.cpp) instead of the reference's CPython recursion.  The outputs are
byte-identical to the reference's for the same inputs (tests/test_scene_pipeline.py checks this
against buffers produced by the reference's own classes).

Not reproduced: the reference reads OBJ files through trimesh (src/model/model.py:40-57), which
is not importable here.  ``read_obj`` keeps faces in file order and fan-triangulates polygons;
whether trimesh orders or merges faces differently for a given file is unpinned.
"""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

try:  # Python >= 3.11
    import tomllib as _toml
except ModuleNotFoundError:  # the same parser under its pre-3.11 name
    import tomli as _toml

HERE = Path(__file__).resolve().parent
_SCENE_LIB = None

TRI_DTYPE = np.dtype({"names": ["v0", "material", "v1", "_p1", "v2", "_p2", "normal", "_p3"],
                      "formats": [("<f4", 3), "<u4", ("<f4", 3), "<f4", ("<f4", 3), "<f4", ("<f4", 3), "<f4"],
                      "offsets": [0, 12, 16, 28, 32, 44, 48, 60], "itemsize": 64})
SPHERE_DTYPE = np.dtype({"names": ["center", "radius", "radius2", "material", "_pad"],
                         "formats": [("<f4", 3), "<f4", "<f4", "<u4", ("<f4", 2)],
                         "offsets": [0, 12, 16, 20, 24], "itemsize": 32})
BOX_DTYPE = np.dtype({"names": ["p0", "material", "p1", "_pad"], "formats": [("<f4", 3), "<u4", ("<f4", 3), "<f4"],
                      "offsets": [0, 12, 16, 28], "itemsize": 32})
PLANE_DTYPE = np.dtype({"names": ["point", "material", "normal", "_pad"],
                        "formats": [("<f4", 3), "<u4", ("<f4", 3), "<i4"], "offsets": [0, 12, 16, 28], "itemsize": 32})
MATERIAL_DTYPE = np.dtype({"names": ["color", "emission", "properties"], "formats": ["<u4", "<u4", "<u4"],
                           "offsets": [0, 4, 8], "itemsize": 12})


def _scene_lib():
    global _SCENE_LIB
    if _SCENE_LIB is None:
        so = HERE / "librt_scene.so"
        if not so.exists():
            raise RuntimeError(f"{so} is missing: build it with `make -C raytracer_amd/csrc`")
        L = C.CDLL(str(so))
        L.rts_build_qbvh.restype = C.c_int
        L.rts_build_qbvh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                     C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.rts_build_fast_bvh.restype = C.c_int
        L.rts_build_fast_bvh.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                         C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32)]
        L.rts_free.restype = None
        L.rts_free.argtypes = [C.c_void_p]
        L.rts_last_error.restype = C.c_char_p
        _SCENE_LIB = L
    return _SCENE_LIB


# ---------------------------------------------------------------------------------------------
# packers
# ---------------------------------------------------------------------------------------------
def _f32(value, n):
    """typecheck.is_3d_array / is_4d_array: lists become float32, float32 arrays pass through."""
    if isinstance(value, list):
        a = np.array(value, dtype=np.float32)
    elif isinstance(value, np.ndarray) and value.dtype == np.float32:
        a = value
    else:
        raise TypeError("Value must be a list of float or an numpy array of float 32")
    if a.shape != (n,):
        raise TypeError(f"Value must contain {n} values")
    return a


def _float(value):
    if isinstance(value, bool) or not isinstance(value, (int, float)):
        raise TypeError("Value must be float")
    return float(value)


def pack_material(color=None, emission=None, smoothness=0.0, metallic=0.0, transmission=0.0, ior=1.0) -> np.ndarray:
    """12-byte material record (material.py:197-230)."""
    color = _f32([1.0, 1.0, 1.0] if color is None else color, 3)
    emission = _f32([0.0, 0.0, 0.0, 0.0] if emission is None else emission, 4)
    smoothness, metallic, transmission, ior = map(_float, (smoothness, metallic, transmission, ior))
    out = np.zeros(1, dtype=MATERIAL_DTYPE)
    props = np.round(np.clip(np.array([smoothness, metallic, transmission, ior]), 0.0, 1.0) * 255).astype(np.uint8)
    props[3] = np.uint8(np.round(np.clip(ior / 3.0, 0.0, 1.0) * 255))
    out[0]["properties"] = props.view(np.uint32)[0]
    c = np.clip(color, 0.0, 1.0)
    r = np.round(c[0] * 2047).astype(np.uint32)
    g = np.round(c[1] * 2047).astype(np.uint32)
    b = np.round(c[2] * 1023).astype(np.uint32)
    out[0]["color"] = (r << 21) | (g << 10) | b
    em = np.round(np.clip(emission, 0.0, 1.0) * 255).astype(np.uint8)
    with np.errstate(invalid="ignore"):
        em[3] = emission[3].astype(np.uint8)  # the strength is truncated, not scaled
    out[0]["emission"] = em.view(np.uint32)[0]
    return out


def pack_spheres(centers: np.ndarray, radii, materials) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(records, aabb_min, aabb_max) for n spheres (sphere.py:132-158)."""
    centers = np.asarray(centers, dtype=np.float32).reshape(-1, 3)
    n = centers.shape[0]
    out = np.zeros(n, dtype=SPHERE_DTYPE)
    radii = [float(r) for r in radii]
    r32 = np.array(radii, dtype=np.float32)
    out["center"] = centers
    out["radius"] = r32
    out["radius2"] = np.array([r * r for r in radii], dtype=np.float32)  # squared in double, then stored
    out["material"] = np.asarray(materials, dtype=np.uint32)
    return out, centers - r32[:, None], centers + r32[:, None]


def pack_triangles(verts: np.ndarray, materials) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(records, aabb_min, aabb_max) for (n,3,3) float32 vertices (triangle.py:152-188)."""
    v = np.asarray(verts, dtype=np.float32).reshape(-1, 3, 3)
    n = v.shape[0]
    out = np.zeros(n, dtype=TRI_DTYPE)
    e1 = v[:, 1] - v[:, 0]
    e2 = v[:, 2] - v[:, 0]
    cr = np.empty((n, 3), dtype=np.float32)
    cr[:, 0] = e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1]
    cr[:, 1] = e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2]
    cr[:, 2] = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    with np.errstate(invalid="ignore", divide="ignore"):
        # numpy.linalg.norm of a float32 3-vector is sqrt(x.dot(x)); the BLAS dot rounds each
        # product to float32 and sums them in double before rounding the sum once
        sq = cr * cr
        sqnorm = (sq[:, 0].astype(np.float64) + sq[:, 1].astype(np.float64) + sq[:, 2].astype(np.float64))
        norm = np.sqrt(sqnorm.astype(np.float32))
        out["normal"] = cr / norm[:, None]
    out["v0"], out["v1"], out["v2"] = v[:, 0], v[:, 1], v[:, 2]
    out["material"] = np.asarray(materials, dtype=np.uint32)
    return out, v.min(axis=1), v.max(axis=1)


def pack_boxes(p0, p1, materials) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(records, aabb_min, aabb_max) for n axis-aligned boxes given by two opposite corners
    (src/primitive/box/box.py:40-47,109-133: corners stored as given, AABB = their min / max)."""
    p0 = np.asarray(p0, dtype=np.float32).reshape(-1, 3)
    p1 = np.asarray(p1, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(p0.shape[0], dtype=BOX_DTYPE)
    out["p0"], out["p1"] = p0, p1
    out["material"] = np.asarray(materials, dtype=np.uint32)
    return out, np.minimum(p0, p1), np.maximum(p0, p1)


def pack_planes(points, normals, materials) -> np.ndarray:
    """Records of n unbounded planes (src/primitive/plane/plane.py:37-47,95-107); planes have no
    bounding box and are not part of the BVH."""
    points = np.asarray(points, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(points.shape[0], dtype=PLANE_DTYPE)
    out["point"] = points
    out["normal"] = np.asarray(normals, dtype=np.float32).reshape(-1, 3)
    out["material"] = np.asarray(materials, dtype=np.uint32)
    return out


def quad_to_triangles(center, width, height, normal):
    """Two triangles of a rectangle (utils.py:164-194); double precision, as lists of floats."""
    center = np.array(center, dtype=float)
    normal = np.array(normal, dtype=float)
    normal /= np.linalg.norm(normal)
    reference = np.array([1.0, 0.0, 0.0])
    if np.allclose(normal, reference) or np.allclose(normal, -reference):
        reference = np.array([0.0, 1.0, 0.0])
    tangent = np.cross(normal, reference)
    tangent /= np.linalg.norm(tangent)
    bitangent = np.cross(normal, tangent)
    tangent *= width / 2.0
    bitangent *= height / 2.0
    p0 = center - tangent - bitangent
    p1 = center + tangent - bitangent
    p2 = center + tangent + bitangent
    p3 = center - tangent + bitangent
    return [[p0.tolist(), p1.tolist(), p2.tolist()], [p0.tolist(), p2.tolist(), p3.tolist()]]


def read_obj(path) -> np.ndarray:
    """(n,3,3) float32 triangles of a Wavefront OBJ: faces in file order, polygons as fans from
    their first vertex, `v/vt/vn` tokens reduced to the position index, negative indices relative
    to the vertices read so far; every other statement (vt, vn, g, o, s, usemtl, mtllib) is skipped
    -- the reference gives a mesh ONE material from its scene file and ignores MTL files
    (src/scene/scene.py:170,181).  The reference reads meshes with trimesh
    (src/model/model.py:40-57), which is not available here: for a file of triangles with one
    material trimesh keeps this order too; for a file with several `usemtl` groups it returns one
    geometry per material, so the reference's triangle order is then grouped by material, which
    this reader does not imitate (ids differ, images do not)."""
    verts: list[tuple[float, float, float]] = []
    faces: list[tuple[int, int, int]] = []
    with Path(path).open() as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
    va = np.array(verts, dtype=np.float32).reshape(-1, 3)
    fa = np.array(faces, dtype=np.int64).reshape(-1, 3)
    if fa.size and (fa.min() < 0 or fa.max() >= va.shape[0]):
        raise ValueError(f"{path}: a face refers to vertex {int(fa.max()) + 1} of {va.shape[0]}")
    return va[fa]


def write_obj(path, meshes, styles=("plain", "full", "negative", "normal_only")) -> int:
    """Writes triangle soups as a Wavefront OBJ that read_obj turns back into exactly the same
    triangles in the same order.  `meshes`: list of ((n,3,3) float32 triangles, material index).
    Every mesh becomes a group (`g`, `usemtl`) with its vertices deduplicated; consecutive triangles
    that form a fan (same first vertex, the next one starting where the last ended) are merged
    into one polygon face, so meshes stored quad by quad come out as quads.  The face syntax
    cycles through `styles` mesh by mesh: "plain" `f a b c`, "full" `f a/t/n ...` with vt and vn
    lines, "negative" relative indices, "normal_only" `f a//n`.  Coordinates are written with nine
    significant digits, which a float32 survives.  Returns the number of faces written."""
    nv = 0      # vertices written so far
    nfaces = 0
    with Path(path).open("w") as f:
        f.write("# written by raytracer_amd.scene.write_obj\nmtllib scene.mtl\n")
        for k, (tris, material) in enumerate(meshes):
            tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 3, 3)
            style = styles[k % len(styles)]
            flat = tris.reshape(-1, 3)
            uniq, inv = np.unique(flat.view(np.uint32), axis=0, return_inverse=True)
            uniq = uniq.view(np.float32)
            inv = inv.reshape(-1, 3)
            f.write(f"g mesh{k}\nusemtl material{int(material)}\n")
            for v in uniq:
                f.write("v %.9g %.9g %.9g\n" % (v[0], v[1], v[2]))
            if style in ("full", "normal_only"):
                f.write("vn 0 1 0\n")
            if style == "full":
                f.write("vt 0.5 0.5\n")
            n_here = uniq.shape[0]

            def tok(local):
                if style == "negative":
                    return str(int(local) - n_here)
                g = nv + int(local) + 1
                return f"{g}/1/1" if style == "full" else f"{g}//1" if style == "normal_only" else str(g)

            i = 0
            while i < inv.shape[0]:
                poly = [inv[i, 0], inv[i, 1], inv[i, 2]]
                j = i + 1
                # a fan continues while the next triangle is (first, last of this one, something new)
                while j < inv.shape[0] and inv[j, 0] == poly[0] and inv[j, 1] == poly[-1] and len(poly) < 8:
                    poly.append(inv[j, 2])
                    j += 1
                f.write("f " + " ".join(tok(p) for p in poly) + "\n")
                nfaces += 1
                i = j
            nv += n_here
    return nfaces


def build_qbvh(aabb_min, aabb_max, types, indices):
    """Native 4-wide BVH build; returns (nodes (n,16) uint32, scene_min f32[3], scene_extent f32[3])."""
    L = _scene_lib()
    mn = np.ascontiguousarray(aabb_min, dtype=np.float32)
    mx = np.ascontiguousarray(aabb_max, dtype=np.float32)
    ty = np.ascontiguousarray(types, dtype=np.uint8)
    ix = np.ascontiguousarray(indices, dtype=np.uint32)
    n = ty.shape[0]
    nodes_p = C.POINTER(C.c_uint32)()
    n_nodes = C.c_uint32()
    smin = (C.c_float * 3)()
    sext = (C.c_float * 3)()
    rc = L.rts_build_qbvh(mn.ctypes.data, mx.ctypes.data, ty.ctypes.data, ix.ctypes.data, n, C.byref(nodes_p),
                          C.byref(n_nodes), smin, sext)
    if rc != 0:
        raise RuntimeError(L.rts_last_error().decode())
    try:
        nodes = np.ctypeslib.as_array(nodes_p, shape=(n_nodes.value, 16)).copy()
    finally:
        L.rts_free(nodes_p)
    return nodes, np.array(list(smin), dtype=np.float32), np.array(list(sext), dtype=np.float32)


# ---------------------------------------------------------------------------------------------
# loader
# ---------------------------------------------------------------------------------------------
def _variable(value, variables):
    """scene.py:257-265: "{name}" looks the name up in [variables]; other strings give None."""
    if not isinstance(value, str):
        return value
    if value.startswith("{") and value.endswith("}"):
        return (variables or {}).get(value[1:-1])
    return None


class SceneBuffers:
    """Everything ``load_scene`` hands to the renderer."""

    def __init__(self, materials, spheres, triangles, bvh, scene_min, scene_extent, boxes=None, planes=None):
        self.materials, self.spheres, self.triangles, self.bvh = materials, spheres, triangles, bvh
        self.scene_min, self.scene_extent = scene_min, scene_extent
        self.boxes = np.zeros(0, dtype=BOX_DTYPE) if boxes is None else boxes
        self.planes = np.zeros(0, dtype=PLANE_DTYPE) if planes is None else planes

    def upload(self, renderer) -> None:
        """The reference's upload order and bindings (scene.py:97,122,193,62,70-71)."""
        renderer.update_ssbo(renderer.materials, self.materials, 9)
        renderer.update_ssbo(renderer.spheres, self.spheres, 5)
        renderer.update_ssbo(renderer.triangles, self.triangles, 12)
        renderer.update_ssbo(renderer.bvh, self.bvh, 11)
        # bindings 7 and 6: the kernel reads them (compute.glsl:77-82), the reference loader
        # never fills them (scene.py:31 is commented out); always sent so that a scene without
        # boxes or planes clears what an earlier scene left
        renderer.update_ssbo(renderer.boxes, self.boxes, 7)
        renderer.update_ssbo(renderer.planes, self.planes, 6)
        renderer.scene_min_coords = self.scene_min
        renderer.scene_extent = self.scene_extent


def build_scene(scene: dict, base_dir=".", mesh_override: dict | None = None,
                extra_meshes: list | None = None, sequential_mesh_indices: bool = False) -> SceneBuffers:
    """TOML dict -> buffers.

    mesh_override maps a [[meshes]] path to another path or to an (n,3,3) float32 array;
    extra_meshes is a list of ((n,3,3) array, material) appended after the scene's own meshes.

    The reference numbers the triangles of EVERY mesh from the end of the quads
    (scene.py:180-187: ``j + triangle_index`` with j restarting per mesh), so in a scene with
    several meshes the BVH leaves of the later meshes point into the first one.  That is
    reproduced by default (byte parity); ``sequential_mesh_indices=True`` numbers the meshes
    consecutively instead, which is what a multi-material scene needs to be visible at all.
    """
    variables = scene.get("variables")
    mats = [pack_material(color=_variable(m.get("color", [1.0, 1.0, 1.0]), variables),
                          emission=_variable(m.get("emission", [0.0, 0.0, 0.0, 0.0]), variables),
                          smoothness=_variable(m.get("smoothness", 0), variables),
                          metallic=_variable(m.get("metallic", 0), variables),
                          transmission=_variable(m.get("transmission", 0), variables),
                          ior=_variable(m.get("ior", 1.0), variables)) for m in scene["materials"]]
    materials = np.concatenate(mats)

    sph = scene.get("spheres", [])
    centers = np.array([_f32(_variable(s.get("center"), variables), 3) for s in sph], dtype=np.float32).reshape(-1, 3)
    radii = [_float(_variable(s.get("radius"), variables)) for s in sph]
    spheres, s_mn, s_mx = pack_spheres(centers, radii, [int(_variable(s.get("material"), variables)) for s in sph])

    tri_v, tri_m = [], []
    for q in scene.get("quads", []):
        t = quad_to_triangles(q.get("center"), q.get("width"), q.get("height"), q.get("normal"))
        tri_v.append(np.array(t, dtype=np.float32))
        tri_m += [q.get("material")] * 2
    n_quad_tris = len(tri_m)
    mesh_index = []  # every mesh restarts its index at the end of the quads, as the reference does
    meshes = []
    for m in scene.get("meshes") or []:
        src = str(m.get("path"))
        if mesh_override and src in mesh_override:
            src = mesh_override[src]
        arr = src if isinstance(src, np.ndarray) else read_obj(Path(base_dir) / src)
        meshes.append((arr, int(m.get("material"))))
    meshes += list(extra_meshes or [])
    for arr, material in meshes:
        arr = np.asarray(arr, dtype=np.float32).reshape(-1, 3, 3)
        tri_v.append(arr)
        tri_m += [material] * arr.shape[0]
        start = len(tri_m) - arr.shape[0] if sequential_mesh_indices else n_quad_tris
        mesh_index.append(np.arange(arr.shape[0], dtype=np.uint32) + start)
    verts = np.concatenate(tri_v) if tri_v else np.zeros((0, 3, 3), np.float32)
    triangles, t_mn, t_mx = pack_triangles(verts, tri_m)
    tri_index = np.concatenate([np.arange(n_quad_tris, dtype=np.uint32)] + mesh_index)

    # [[boxes]] (p0, p1, material) and [[planes]] (point, normal, material): tables the reference's
    # schema reserves (scene.py:31,125-147) and its kernel consumes (compute.glsl:315-323,788-806)
    bxs = scene.get("boxes") or []
    boxes, b_mn, b_mx = pack_boxes([_f32(_variable(b.get("p0"), variables), 3) for b in bxs],
                                   [_f32(_variable(b.get("p1"), variables), 3) for b in bxs],
                                   [int(_variable(b.get("material"), variables)) for b in bxs])
    pls = scene.get("planes") or []
    planes = pack_planes([_f32(_variable(p.get("point"), variables), 3) for p in pls],
                         [_f32(_variable(p.get("normal"), variables), 3) for p in pls],
                         [int(_variable(p.get("material"), variables)) for p in pls])

    aabb_min = np.concatenate([s_mn.reshape(-1, 3), t_mn.reshape(-1, 3), b_mn.reshape(-1, 3)])
    aabb_max = np.concatenate([s_mx.reshape(-1, 3), t_mx.reshape(-1, 3), b_mx.reshape(-1, 3)])
    types = np.concatenate([np.ones(len(sph), np.uint8), np.zeros(len(tri_m), np.uint8),
                            np.full(len(bxs), 2, np.uint8)])
    indices = np.concatenate([np.arange(len(sph), dtype=np.uint32), tri_index, np.arange(len(bxs), dtype=np.uint32)])
    bvh, smin, sext = build_qbvh(aabb_min, aabb_max, types, indices)
    return SceneBuffers(materials, spheres, triangles, bvh, smin, sext, boxes, planes)


def load_scene(filepath: str, renderer=None, mesh_override: dict | None = None) -> SceneBuffers:
    """reference load_scene(filepath): parse, pack, build, upload to the renderer."""
    path = Path(filepath)
    with path.open("rb") as f:
        scene = _toml.load(f)
    # mesh paths in the reference's scenes are relative to the working directory ("./models/...")
    bufs = build_scene(scene, base_dir=".", mesh_override=mesh_override)
    if renderer is None:
        from .renderer import Renderer

        renderer = Renderer.get_instance()
    bufs.upload(renderer)
    return bufs


def build_fast_bvh(triangles=b"", spheres=b"", boxes=b""):
    """The fast-mode tree for reference-layout triangle (64 B), sphere (32 B) and box (32 B) buffers,
    built on the host exactly as the render library builds it when fast mode is switched on: a
    uint32 array of 16 words per node (layout: csrc/rt_fastbvh.cpp), the last one an all-empty sentinel."""
    L = _scene_lib()
    bufs = [np.frombuffer(bytes(np.ascontiguousarray(b).tobytes() if not isinstance(b, (bytes, bytearray)) else b),
                          dtype=np.uint32).copy() for b in (triangles, spheres, boxes)]
    nodes = C.POINTER(C.c_uint32)()
    nn = C.c_uint32()
    rc = L.rts_build_fast_bvh(bufs[0].ctypes.data, bufs[0].size // 16, bufs[1].ctypes.data, bufs[1].size // 8,
                              bufs[2].ctypes.data, bufs[2].size // 8, C.byref(nodes), C.byref(nn))
    if rc != 0:
        raise RuntimeError(L.rts_last_error().decode())
    try:
        return np.ctypeslib.as_array(nodes, shape=((nn.value + 1) * 16,)).copy().reshape(-1, 16)
    finally:
        L.rts_free(nodes)
