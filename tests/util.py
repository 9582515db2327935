"""Shared helpers for the tests: loading a golden scene into the HIP renderer."""

import hashlib
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_golden_scene(renderer, scene: str):
    """Uploads the byte buffers of tests/golden/<scene> in the reference loader's order
    (materials 9, spheres 5, triangles 12, BVH 11; reference scene.py:97,122,193,62), then boxes 7
    and planes 6 (empty for the scenes that have none)."""
    d = GOLDEN / scene
    for name, binding in (("mats.bin", 9), ("spheres.bin", 5), ("tris.bin", 12), ("bvh.bin", 11),
                          ("boxes.bin", 7), ("planes.bin", 6)):
        data = np.fromfile(d / name, dtype=np.uint8) if (d / name).exists() else np.zeros(0, np.uint8)
        renderer.update_ssbo(0, data, binding)
    u = np.fromfile(d / "uniforms.bin", dtype=np.float32)
    renderer.set_uniforms(u[0:16], u[16:19])
    renderer.scene_min_coords = u[19:22]
    renderer.scene_extent = u[22:25]
    return u


def render_frame(renderer, w, h, frame, bounces, aux=False, stats=False):
    if (renderer.width, renderer.height) != (w, h):
        renderer.update_size(w, h)
    renderer.bounces = bounces
    renderer.set_frame_index(frame)
    renderer.set_flags(aux=aux, stats=stats)
    renderer.camera.changed = False
    renderer.render()
    return renderer.read_sample()


def fuzz_cases():
    return sorted((GOLDEN / "fuzz").glob("case*.json"))


def fuzz_buffers(rec):
    """Buffers of a random scene rebuilt by the host pipeline, checked against the hashes of the
    buffers the reference's own classes produced; returns (SceneBuffers, 25 uniform floats)."""
    from raytracer_amd import scene as S

    b = S.build_scene(rec["scene"])
    for key, arr in (("mats", b.materials), ("spheres", b.spheres), ("tris", b.triangles), ("bvh", b.bvh),
                     ("boxes", b.boxes), ("planes", b.planes)):
        assert sha(arr) == rec["buffers_sha256"][key], key
    u = np.array([int(x, 16) for x in rec["uniform_words"]], dtype=np.uint32).view(np.float32)
    return b, u


def damaged_gally_bvh():
    """The gally tree with child words that point outside the arrays: the reference's own empty-slot
    word (0xFFFFFFFF = node 0x3FFFFFFF) on slots that keep a real box, node indices just past the
    end, triangle indices past the end.  Returns (nodes as (n, 16) uint32, how many of each kind)."""
    d = GOLDEN / "gally"
    bvh = np.fromfile(d / "bvh.bin", dtype=np.uint32).reshape(-1, 16).copy()
    n_nodes = bvh.shape[0]
    n_tris = (d / "tris.bin").stat().st_size // 64
    rng = np.random.default_rng(7)
    changed = {"empty": 0, "node": 0, "tri": 0}
    for ni in rng.choice(n_nodes, size=min(400, n_nodes), replace=False):
        k = int(rng.integers(4))
        word = int(bvh[ni, 12 + k])
        kind = int(rng.integers(3))
        if word >> 30 == 3 and word != 0xFFFFFFFF and kind == 0:
            bvh[ni, 12 + k] = 0xFFFFFFFF  # the box stays: rays that hit it push node 0x3FFFFFFF
            changed["empty"] += 1
        elif word >> 30 == 3 and word != 0xFFFFFFFF and kind == 1:
            bvh[ni, 12 + k] = 0xC0000000 | (n_nodes + int(rng.integers(1, 9)))
            changed["node"] += 1
        elif word >> 30 == 0:
            bvh[ni, 12 + k] = n_tris + int(rng.integers(0, 1000))
            changed["tri"] += 1
    return bvh, changed


def damaged_gally_materials():
    """The gally triangles and spheres with material indices beyond the material array on every
    seventh triangle and on the first sphere.  Returns (tris as (n, 16) uint32, spheres as (n, 8) uint32)."""
    d = GOLDEN / "gally"
    tris = np.fromfile(d / "tris.bin", dtype=np.uint32).reshape(-1, 16).copy()
    sph = np.fromfile(d / "spheres.bin", dtype=np.uint32).reshape(-1, 8).copy()
    n_mats = (d / "mats.bin").stat().st_size // 12
    tris[::7, 3] = n_mats + (np.arange(tris[::7].shape[0], dtype=np.uint32) % 5)
    if sph.shape[0]:
        sph[0, 5] = 0x7FFFFFFF
    return tris, sph


def spine_tree(depth: int):
    """A hand-made scene whose traversal stack grows by three entries per level: `depth` spine nodes,
    each with the next spine node in slot 0 and three small subtrees (an inner node over four
    spheres) in slots 1..3, every child box being the whole scene box with the camera inside.  All
    entry distances are 0, so the sort keeps the slot order, slot 0 is popped first and the three
    siblings of every level wait on the stack: the deepest pop happens at 3 * depth + 1 entries.
    (With the reference's builder a tree this deep needs millions of primitives; its shader's own
    stack has 32 entries and no guard, so beyond depth 10 the reference itself is undefined.)
    Returns (spheres (n, 8) uint32, nodes (m, 16) uint32, materials (2, 3) uint32, 25 uniform floats)."""
    from raytracer_amd import scene as S
    from raytracer_amd.camera import Camera

    full = np.uint32(0xFFFF0000)
    spheres, nodes = [], []

    def sphere(i):
        # a loose spiral in front of the default camera (which looks along -x from (8, 1.5, 0))
        a = 0.61803 * i
        return [float(2.0 - 0.035 * i), float(1.5 + 2.2 * np.sin(a) * (0.3 + (i % 7) / 9.0)),
                float(2.8 * np.cos(a) * (0.3 + (i % 5) / 7.0))], 0.16 + 0.02 * (i % 3)

    def leaf_node():
        words = np.zeros(16, np.uint32)
        words[0:12] = full
        for k in range(4):
            words[12 + k] = (1 << 30) | len(spheres)
            spheres.append(sphere(len(spheres)))
        return words

    # node numbering: spine node k is 4k, its three subtrees 4k+1 .. 4k+3; the spine ends in a leaf node
    for k in range(depth):
        words = np.zeros(16, np.uint32)
        words[0:12] = full
        words[12] = (3 << 30) | (4 * (k + 1))
        for j in range(1, 4):
            words[12 + j] = (3 << 30) | (4 * k + j)
        nodes.append(words)
        for j in range(1, 4):
            nodes.append(leaf_node())
    nodes.append(leaf_node())
    centers = np.array([c for c, _ in spheres], np.float32)
    radii = [r for _, r in spheres]
    sph, _, _ = S.pack_spheres(centers, radii, [1] * len(radii))
    mats = np.concatenate([S.pack_material(color=[1.0, 1.0, 1.0], emission=[1.0, 1.0, 1.0, 0.0]),
                           S.pack_material(color=[0.9, 0.5, 0.3], emission=[1.0, 0.8, 0.6, 2.0], smoothness=0.4)])
    inv_vp, eye = Camera().uniforms_f32()
    u = np.concatenate([inv_vp, eye, [-6.0, -3.0, -5.0], [16.0, 9.0, 10.0]]).astype(np.float32)
    return (np.frombuffer(sph.tobytes(), np.uint32).reshape(-1, 8), np.array(nodes, np.uint32), 
            np.frombuffer(mats.tobytes(), np.uint32).reshape(-1, 3), u)
