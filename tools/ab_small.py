#!/usr/bin/env python3
"""Steady frame time of the cache-resident configurations (C2 sphere.obj, C3 gally) and the bench
scene for the library named by RT_MI355X_LIB: tools/ab_small.py  (run on the GPU box)"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from raytracer_amd import scene as S, synth  # noqa: E402
from raytracer_amd.camera import Camera  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402

GOLD = ROOT / "tests" / "golden"
gally = json.loads((GOLD / "gally" / "scene.json").read_text())
cases = {
    "C2": S.build_scene(gally, extra_meshes=[(np.load(GOLD / "meshes" / "sphere.npz")["triangles"], 26)]),
    "C3": S.build_scene(gally, extra_meshes=[(np.load(GOLD / "meshes" / "gally.npz")["triangles"], 26)]),
}
sc, meshes = synth.sponza_synth(1.0)
cases["C4"] = S.build_scene(sc, extra_meshes=meshes, sequential_mesh_indices=True)
out = {}
for name, bufs in cases.items():
    r = Renderer(device=0, camera=Camera(), singleton=False)
    r.update_size(1920, 1080)
    bufs.upload(r)
    r.bounces = 4
    r.camera.changed = False
    r.render(frames=70)
    r.sync()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        r.render(frames=64)
        r.sync()
        best = min(best, (time.perf_counter() - t) / 64 * 1e3)
    out[name] = round(best, 4)
    out[name + "_variant"] = r.traversal_variant()[0]
    r.close()
print(json.dumps(out))
