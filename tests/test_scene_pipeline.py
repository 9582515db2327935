"""Host scene pipeline (TOML dict + triangle soups -> buffers) against buffers produced by the
reference's own Python classes: byte-identical materials, spheres, triangles and BVH nodes."""

import hashlib
import json

import numpy as np
import pytest

from raytracer_amd import scene as S, synth
from util import GOLDEN

MESHES = {"debug": [], "debug_box": [("box", 1)], "gally": [("gally", 26)], "matlab": [("gally", 38), ("box", 23)],
          "prims": []}


def _mesh(name):
    return np.load(GOLDEN / "meshes" / f"{name}.npz")["triangles"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", sorted(MESHES))
def test_buffers_byte_identical_to_reference_loader(name):
    scene = json.loads((GOLDEN / name / "scene.json").read_text())
    b = S.build_scene(scene, extra_meshes=[(_mesh(m), mat) for m, mat in MESHES[name]])
    for key, arr in (("mats", b.materials), ("spheres", b.spheres), ("tris", b.triangles), ("bvh", b.bvh)):
        assert arr.tobytes() == (GOLDEN / name / f"{key}.bin").read_bytes(), key
    for key, arr in (("boxes", b.boxes), ("planes", b.planes)):
        f = GOLDEN / name / f"{key}.bin"
        assert arr.tobytes() == (f.read_bytes() if f.exists() else b""), key
    u = np.fromfile(GOLDEN / name / "uniforms.bin", dtype=np.float32)
    assert np.array_equal(u[19:22], b.scene_min) and np.array_equal(u[22:25], b.scene_extent)


def test_large_scenes_hash_identical_to_reference_builder():
    """32 k and 265 k triangle scenes: only hashes of the reference's output are kept."""
    big = json.loads((GOLDEN / "big_scene_hashes.json").read_text())
    scene = json.loads((GOLDEN / "gally" / "scene.json").read_text())
    b = S.build_scene(scene, extra_meshes=[(_mesh("sphere"), 26)])
    g = big["test_sphere_obj"]
    assert b.bvh.shape[0] == g["nodes"] and b.triangles.shape[0] == g["triangles"]
    for key, arr in (("mats", b.materials), ("spheres", b.spheres), ("tris", b.triangles), ("bvh", b.bvh)):
        assert sha(arr) == g["sha256"][key], key
    sc, meshes = synth.sponza_synth(1.0)
    b = S.build_scene(sc, extra_meshes=meshes)
    g = big["sponza_synth"]
    assert b.bvh.shape[0] == g["nodes"] and b.triangles.shape[0] == g["triangles"]
    for key, arr in (("mats", b.materials), ("spheres", b.spheres), ("tris", b.triangles), ("bvh", b.bvh)):
        assert sha(arr) == g["sha256"][key], key


def test_camera_known_answer():
    """Default-camera uniform words recorded from a run of the reference (debug scene fixture)."""
    from raytracer_amd.camera import Camera

    m, eye = Camera().uniforms_f32()
    u = np.fromfile(GOLDEN / "debug" / "uniforms.bin", dtype=np.float32)
    assert np.array_equal(m.view(np.uint32), u[:16].view(np.uint32))
    assert np.array_equal(eye, u[16:19])
    words = " ".join("%08x" % w for w in m.view(np.uint32))
    assert words.startswith("a4cda698 a2ff111c bf3c8361 22eba439 00000000 3ed413cd")


def test_material_known_answers():
    a = S.pack_material(color=[1.0, 1.0, 1.0], emission=[1.0, 1.0, 1.0, 12.0])
    assert (int(a["color"][0]), int(a["emission"][0]), int(a["properties"][0])) == (0xFFFFFFFF, 0x0CFFFFFF, 0x55000000)
    b = S.pack_material(color=[1.0, 0.0, 0.0], smoothness=0.5, transmission=1.0, ior=1.5)
    assert (int(b["color"][0]), int(b["emission"][0]), int(b["properties"][0])) == (0xFFE00000, 0, 0x80FF0080)


def test_single_primitive_and_empty_inputs():
    # one primitive: a root with four empty slots (the reference renders nothing for it)
    nodes, smin, sext = S.build_qbvh([[0, 0, 0]], [[1, 1, 1]], [1], [0])
    assert nodes.shape == (1, 16) and (nodes[0, 12:] == 0xFFFFFFFF).all() and (nodes[0, :12] == 0).all()
    with pytest.raises(RuntimeError):
        S.build_qbvh(np.zeros((0, 3)), np.zeros((0, 3)), [], [])
    with pytest.raises(TypeError):
        S.pack_material(color=[1.0, 1.0])


def test_variable_substitution_and_sphere_radius():
    scene = {"variables": {"r": 0.4, "white": [1.0, 1.0, 1.0]},
             "materials": [{"color": "{white}"}],
             "spheres": [{"center": [0.0, 0.0, 0.0], "radius": "{r}", "material": 0},
                         {"center": [2.0, 0.0, 0.0], "radius": 1, "material": 0}]}
    b = S.build_scene(scene)
    assert b.spheres["radius"][0] == np.float32(0.4) and b.spheres["radius2"][0] == np.float32(0.4 * 0.4)
    assert b.spheres["radius"][1] == 1.0


def test_reference_update_ssbo_arguments_are_the_golden_bytes():
    """The structured arrays the reference loader passes to update_ssbo (captured with their dtypes
    and shapes) hold exactly the bytes of the golden buffers, and numpy reports their size in bytes
    the way Renderer.update_ssbo relies on."""
    args = np.load(GOLDEN / "debug" / "update_ssbo_args.npz")
    for key, name in (("materials", "mats.bin"), ("spheres", "spheres.bin"), ("triangles", "tris.bin"), ("bvh", "bvh.bin")):
        a = np.ascontiguousarray(args[key])
        assert a.tobytes() == (GOLDEN / "debug" / name).read_bytes(), key
        assert a.nbytes == (GOLDEN / "debug" / name).stat().st_size
    u = np.fromfile(GOLDEN / "debug" / "uniforms.bin", dtype=np.float32)
    assert np.array_equal(args["scene_min_coords"], u[19:22]) and np.array_equal(args["scene_extent"], u[22:25])
