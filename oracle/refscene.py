"""TEST INFRASTRUCTURE (oracle side) — builds scene buffers with the REFERENCE's own classes.

Runs only in the build container, where /root/reference exists.  It imports
the numpy-only half of the reference (Material, Sphere, Triangle, BVHNode,
QBVHNode, quad_to_triangles) and replays the reference scene loader's steps
(reference src/scene/scene.py:16-71) without its OpenGL renderer, so the byte
buffers it returns are exactly what ``Renderer.update_ssbo`` would have been
handed for bindings 9, 5, 12 and 11.

Two pieces of the reference loader cannot be imported here and are restated:
* ``tomllib`` (py>=3.11)  -> ``tomli`` (same parser, same API)
* ``trimesh.load`` (model.py:40-57) -> a minimal OBJ reader that keeps faces in
  file order and fan-triangulates polygons.  Face order parity with trimesh is
  unpinned; kernel parity does not depend on it (both sides get the same buffer).
"""

from __future__ import annotations

import contextlib
import io
import sys
import warnings
from pathlib import Path

import numpy as np

REF = Path("/root/reference")


def _import_reference():
    if not REF.exists():
        raise RuntimeError("/root/reference is not present: fixtures can only be generated in the build container")
    src = str(REF / "src")
    if src not in sys.path:
        sys.path.insert(0, src)
    from bvh import BVHNode, QBVHNode  # noqa: PLC0415
    from material import Material  # noqa: PLC0415
    from primitive import Box, Plane, Sphere, Triangle  # noqa: PLC0415
    from utils.utils import quad_to_triangles  # noqa: PLC0415

    return BVHNode, QBVHNode, Material, Sphere, Triangle, quad_to_triangles, Box, Plane


def read_obj_triangles(path: str | Path) -> np.ndarray:
    """(n,3,3) float32 triangle soup, faces in file order, polygons fan-triangulated."""
    verts: list[list[float]] = []
    tris: list[list[list[float]]] = []
    with Path(path).open() as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append([float(p[1]), float(p[2]), float(p[3])])
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    tris.append([verts[idx[0]], verts[idx[k]], verts[idx[k + 1]]])
    return np.array(tris, dtype=np.float32).reshape(-1, 3, 3)


def _variable(value, variables):
    # reference scene.py:257-265
    if not isinstance(value, str):
        return value
    if value.startswith("{") and value.endswith("}"):
        return variables.get(value[1:-1])
    return None


def build_reference_scene(scene: dict, mesh_tris: list[tuple[np.ndarray, int]] | None = None, quiet=True):
    """Replays reference load_scene on a parsed TOML dict.

    mesh_tris: list of ((n,3,3) float32 triangles, material index) replacing the
    [[meshes]] entries (whose OBJ parsing is not importable).
    Returns dict(mats, spheres, tris, bvh, boxes, planes: bytes; scene_min, scene_extent: f32[3]).
    """
    BVHNode, QBVHNode, Material, Sphere, Triangle, quad_to_triangles, Box, Plane = _import_reference()
    variables = scene.get("variables") or {}
    sink = io.StringIO()
    ctx = contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext()
    with ctx, warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # materials: scene.py:74-97
        mats = []
        for m in scene["materials"]:
            mo = Material(
                color=_variable(m.get("color", [1.0, 1.0, 1.0]), variables),
                emission=_variable(m.get("emission", [0.0, 0.0, 0.0, 0.0]), variables),
                smoothness=_variable(m.get("smoothness", 0), variables),
                metallic=_variable(m.get("metallic", 0), variables),
                transmission=_variable(m.get("transmission", 0), variables),
                ior=_variable(m.get("ior", 1.0), variables),
            )
            mats.append(mo.ogl_ssbo_data)
        mats_b = np.concatenate(mats).tobytes()
        # the arrays exactly as the reference loader hands them to Renderer.update_ssbo
        # (structured dtypes; shape (n,) for materials, (n, 1) for spheres and triangles)
        args = {"materials": np.concatenate(mats)}  # scene.py:91-97 concatenates pairwise: same array
        prims = []
        # spheres: scene.py:100-122
        sph = []
        for index, s in enumerate(scene.get("spheres", [])):
            sp = Sphere(
                center=_variable(s.get("center"), variables),
                radius=_variable(s.get("radius"), variables),
                material=_variable(s.get("material"), variables),
            )
            sph.append(sp.ogl_ssbo_data)
            prims.append({"primitive": sp, "index": index, "primitive_type": "sphere"})
        sph_b = np.concatenate(sph).tobytes() if sph else b""
        if sph:
            args["spheres"] = np.concatenate([sph])  # scene.py:121
        # quads: scene.py:196-248
        tri_data = []
        max_index = 0
        for index, q in enumerate(scene.get("quads", [])):
            t = quad_to_triangles(
                center=q.get("center"), width=q.get("width"), height=q.get("height"), normal=q.get("normal")
            )
            t1 = Triangle(v0=t[0][0], v1=t[0][1], v2=t[0][2], material=q.get("material"))
            t2 = Triangle(v0=t[1][0], v1=t[1][1], v2=t[1][2], material=q.get("material"))
            prims.append({"primitive": t1, "index": index * 2, "primitive_type": "triangle"})
            prims.append({"primitive": t2, "index": index * 2 + 1, "primitive_type": "triangle"})
            tri_data.append(t1.ogl_ssbo_data)
            tri_data.append(t2.ogl_ssbo_data)
            max_index = index * 2 + 2
        # meshes: scene.py:150-193 (every mesh restarts its index at max_index, as the reference does)
        for arr, material in mesh_tris or []:
            va = np.array(arr, dtype=np.float32).reshape(-1, 3, 3)
            for j, tri in enumerate(va):
                tr = Triangle(v0=tri[0], v1=tri[1], v2=tri[2], material=int(material))
                prims.append({"primitive": tr, "index": j + max_index, "primitive_type": "triangle"})
                tri_data.append(tr.ogl_ssbo_data)
        tris_b = np.vstack(tri_data).tobytes() if tri_data else b""
        if tri_data:
            args["triangles"] = np.vstack(tri_data)  # scene.py:191
        # boxes: what scene.py:125-147 intends (the call at :31 is commented out and the function
        # appends the growing list instead of the primitive; the primitive itself is used here)
        box_data = []
        for index, bx in enumerate(scene.get("boxes") or []):
            bp = Box(p0=_variable(bx.get("p0"), variables), p1=_variable(bx.get("p1"), variables),
                     material=_variable(bx.get("material"), variables))
            box_data.append(bp.ogl_ssbo_data)
            prims.append({"primitive": bp, "index": index, "primitive_type": "box"})
        boxes_b = np.concatenate(box_data).tobytes() if box_data else b""
        # planes: no loader exists in the reference; records come from its Plane class
        plane_data = [Plane(point=pl.get("point"), normal=pl.get("normal"), material=pl.get("material")).ogl_ssbo_data
                      for pl in scene.get("planes") or []]
        planes_b = np.concatenate(plane_data).tobytes() if plane_data else b""
        root = BVHNode.build_tree(prims)
        q = QBVHNode.from_bvh(root)
        q.flatten_qbvh()
        bvh_b = q.ogl_ssbo_data.tobytes()
        args["bvh"] = q.ogl_ssbo_data  # scene.py:62
        args["scene_min_coords"] = root.aabb.min  # scene.py:70-71
        args["scene_extent"] = root.aabb.max - root.aabb.min
        smin = np.asarray(root.aabb.min, dtype=np.float32)
        sext = np.asarray(root.aabb.max - root.aabb.min, dtype=np.float32)
    return {
        "mats": mats_b,
        "spheres": sph_b,
        "tris": tris_b,
        "bvh": bvh_b,
        "boxes": boxes_b,
        "planes": planes_b,
        "scene_min": smin,
        "scene_extent": sext,
        "update_ssbo_args": args,
    }


# --------------------------------------------------------------------------------------
# shader variants: produced by text substitution on the reference file at run time,
# written to a temporary directory, never into the repository.
# --------------------------------------------------------------------------------------
_F4_OLD = "        mask = mask && lessThan(tNear, vec4(bestHit.t));"
_F4_NEW = (
    "        bvec4 lt_ = lessThan(tNear, vec4(bestHit.t));\n"
    "        mask = bvec4(mask.x && lt_.x, mask.y && lt_.y, mask.z && lt_.z, mask.w && lt_.w);"
)


def _sub(text: str, old: str, new: str, count: int) -> str:
    n = text.count(old)
    if n != count:
        raise RuntimeError(f"expected {count} occurrence(s) of {old!r}, found {n}")
    return text.replace(old, new)


def shader_variant(kind: str) -> str:
    """kind: 'color' (reference output), 'primary' (first-hit ids), 'counters' (work counters)."""
    text = (REF / "shaders" / "compute.glsl").read_text()
    # strict-GLSL fix: `bvec4 && bvec4` does not compile on Mesa; component-wise AND is equivalent
    text = _sub(text, _F4_OLD, _F4_NEW, 1)
    if kind == "color":
        return text
    glob = (
        "uint g_ct = 3u; uint g_ci = 0x3FFFFFFFu; uint g_pt = 3u; uint g_pi = 0x3FFFFFFFu;\n"
        "float g_t = 0.0; bool g_ex = false;\n"
        "uint g_rays = 0u; uint g_nodes = 0u; uint g_tri = 0u; uint g_sph = 0u; uint g_maxsp = 0u;\n"
    )
    text = _sub(text, "// function declarations\n", glob + "// function declarations\n", 1)
    text = _sub(
        text,
        "    stack[stackPtr++] = 0; // root\n",
        "    stack[stackPtr++] = 0; // root\n    g_ct = 3u; g_ci = 0x3FFFFFFFu; g_rays++;\n",
        1,
    )
    text = _sub(
        text,
        "        uint nodeIdx = stack[--stackPtr];\n",
        "        g_maxsp = max(g_maxsp, stackPtr);\n        uint nodeIdx = stack[--stackPtr];\n        g_nodes++;\n",
        1,
    )
    text = _sub(text, "bestHit = prim_hit;", "bestHit = prim_hit; g_ct = prim_type; g_ci = prim_idx;", 3)
    text = _sub(
        text,
        "                if (prim_type == TRIANGLE) {\n",
        "                if (prim_type == TRIANGLE) {\n                    g_tri++;\n",
        1,
    )
    text = _sub(
        text,
        "                } else if (prim_type == SPHERE) {\n",
        "                } else if (prim_type == SPHERE) {\n                    g_sph++;\n",
        1,
    )
    text = _sub(
        text,
        "        hit = calculate_ray_bvh(ray);\n",
        "        hit = calculate_ray_bvh(ray);\n"
        "        if (i == 0u) { g_pt = g_ct; g_pi = g_ci; g_t = hit.t; g_ex = hit.exists; }\n",
        1,
    )
    store_old = "    imageStore(img_output, texel_coord, vec4(rgb_to_srgb(light), 1.0));\n"
    if kind == "primary":
        # x: t of the first ray; y: (type<<16)|(idx & 0xFFFF); z: idx>>16; w: exists
        store_new = (
            "    imageStore(img_output, texel_coord, vec4(g_t, float((g_pt << 16) | (g_pi & 0xFFFFu)),"
            " float(g_pi >> 16), g_ex ? 1.0 : 0.0));\n"
        )
    elif kind == "counters":
        store_new = (
            "    imageStore(img_output, texel_coord, vec4(float(g_nodes), float(g_tri), float(g_sph),"
            " float(g_rays * 64u + g_maxsp)));\n"
        )
    else:
        raise ValueError(kind)
    return _sub(text, store_old, store_new, 1)


def shader_capture_variant(bounce: int, expr: str) -> str:
    """Reference shader whose RGB output is untouched and whose alpha channel carries `expr`
    (a float GLSL expression over main()'s locals) sampled at the end of bounce `bounce`,
    just before `ray.inv_direction` is refreshed (compute.glsl:660).  Pixels whose path ended
    earlier keep alpha 0.  Because RGB is kept, the rest of the program is compiled exactly as in
    the unmodified shader, which makes these captures trustworthy witnesses of its inner state."""
    text = shader_variant("color")
    anchor = "        ray.inv_direction = 1 / ray.direction;\n\n        if (!russianRoulette"
    text = _sub(text, anchor, f"        if (i == {int(bounce)}u) g_cap = {expr};\n" + anchor, 1)
    text = _sub(text, "// function declarations\n", "float g_cap = 0.0;\n// function declarations\n", 1)
    store_old = "    imageStore(img_output, texel_coord, vec4(rgb_to_srgb(light), 1.0));\n"
    return _sub(text, store_old, "    imageStore(img_output, texel_coord, vec4(rgb_to_srgb(light), g_cap));\n", 1)
