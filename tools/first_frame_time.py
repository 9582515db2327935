#!/usr/bin/env python3
"""Where the time before the first image goes on the bench scene: context, uploads (PCIe, from
pageable numpy arrays), first frame (breadth-first renumbering on the host, repack kernels, first
launch of each kernel), second frame."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from raytracer_amd import scene as S, synth  # noqa: E402
from raytracer_amd.camera import Camera  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402

scene, meshes = synth.sponza_synth(1.0)
bufs = S.build_scene(scene, extra_meshes=meshes, sequential_mesh_indices=True)
for rep in range(2):
    t = [time.perf_counter()]
    r = Renderer(device=0, camera=Camera(), singleton=False)
    r.update_size(1920, 1080)
    r.sync(); t.append(time.perf_counter())
    bufs.upload(r)
    r.sync(); t.append(time.perf_counter())
    r.bounces = 4
    r.set_frame_index(0)
    r.camera.changed = False
    r.render(); r.sync(); t.append(time.perf_counter())
    r.render(); r.sync(); t.append(time.perf_counter())
    r.map_image(); t.append(time.perf_counter())
    r.close()
    names = ["create + resize", "uploads", "first frame", "second frame", "first map_image"]
    print(f"run {rep}: " + ", ".join(f"{n} {(b - a) * 1e3:.2f} ms" for n, a, b in zip(names, t, t[1:])))
