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
