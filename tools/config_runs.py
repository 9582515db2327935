#!/usr/bin/env python3
"""Runs every BASELINE.json configuration that fits one GPU and prints one JSON record each:
wall time with the scene upload (PCIe) and the image read-back included, device-only rate, and,
where the reference's own output is held under tests/golden, whether the result is bit-identical.

  python tools/config_runs.py > gpurun_out/config_runs.jsonl
"""
import hashlib
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


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_mean(name, bufs, w, h, spp, bounces, golden_acc_hash=None):
    """Reference semantics: spp frames, running mean after each (Renderer.render)."""
    r = Renderer(device=0, camera=Camera(), singleton=False)
    r.update_size(w, h)
    r.set_flags(stats=False)
    t0 = time.perf_counter()
    bufs.upload(r)
    r.bounces = bounces
    r.set_frame_index(0)
    r.camera.changed = False
    r.render(frames=1)  # includes the one-off repack of the uploaded buffers
    r.sync()
    t_first = time.perf_counter() - t0
    t1 = time.perf_counter()
    if spp > 1:
        r.render(frames=spp - 1)
    r.sync()
    t_rest = time.perf_counter() - t1
    t2 = time.perf_counter()
    img = r.map_image()  # page-locked view, no copy into Python memory
    t_read = time.perf_counter() - t2
    rays = r.stats()["rays"]
    # steady state: the same frames once more, outside the run's own clock and after a batch long
    # enough for the library to have timed its traversal variants on this scene (64 frames)
    r.render(frames=64)
    r.sync()
    t3 = time.perf_counter()
    if spp > 1:
        r.render(frames=spp - 1)
    r.sync()
    t_steady = (time.perf_counter() - t3) / max(1, spp - 1)
    out = {"config": name, "size": [w, h], "spp": spp, "bounces": bounces, "rays": rays,
           "upload_plus_first_frame_ms": t_first * 1e3, "remaining_frames_ms": t_rest * 1e3,
           "ms_per_frame_steady": t_steady * 1e3, "traversal_variant": r.traversal_variant()[0],
           "readback_ms": t_read * 1e3,
           "Mrays_per_s_device": rays / (t_first + t_rest) / 1e6,
           "Mrays_per_s_with_pcie": rays / (t_first + t_rest + t_read) / 1e6}
    if golden_acc_hash:
        out["bit_identical_to_reference_shaders"] = sha(img) == golden_acc_hash
    r.close()
    return out


def run_sum(name, bufs, w, h, spp, bounces):
    """Offline render: spp frames summed in linear space, mean finalised once."""
    r = Renderer(device=0, camera=Camera(), singleton=False)
    r.update_size(w, h)
    t0 = time.perf_counter()
    bufs.upload(r)
    r.bounces = bounces
    r.clear_sum()
    r.render_sum(0, spp)
    r.finalize_sum(spp)
    r.sync()
    t_dev = time.perf_counter() - t0
    t2 = time.perf_counter()
    r.map_image()
    t_read = time.perf_counter() - t2
    rays = r.stats()["rays"]
    variant = r.traversal_variant()[0]
    r.close()
    return {"config": name, "size": [w, h], "spp": spp, "bounces": bounces, "rays": rays, "traversal_variant": variant,
            "upload_plus_render_ms": t_dev * 1e3, "readback_ms": t_read * 1e3,
            "Mrays_per_s_device": rays / t_dev / 1e6, "Mrays_per_s_with_pcie": rays / (t_dev + t_read) / 1e6}


def main():
    def mesh(n):
        return np.load(GOLD / "meshes" / f"{n}.npz")["triangles"]

    debug = json.loads((GOLD / "debug_box" / "scene.json").read_text())
    test = json.loads((GOLD / "gally" / "scene.json").read_text())
    c2 = json.loads((GOLD / "c2_sphere_obj_1920x1080_b4.json").read_text())
    c3 = json.loads((GOLD / "gally" / "c3full_1920x1080_b4.acc.json").read_text())
    recs = [
        run_mean("C1 debug.toml + box.obj 640x360 1 spp 2 bounces", S.build_scene(debug, extra_meshes=[(mesh("box"), 1)]),
                 640, 360, 1, 2),
        run_mean("C2 test.toml + sphere.obj (32 k triangles) 1080p 16 spp 4 bounces",
                 S.build_scene(test, extra_meshes=[(mesh("sphere"), 26)]), 1920, 1080, 16, 4,
                 c2["acc_sha256_per_frame"][15]),
        run_mean("C3 test.toml + gally.obj 1080p 64 spp 4 bounces", S.build_scene(test, extra_meshes=[(mesh("gally"), 26)]),
                 1920, 1080, 64, 4, c3["acc_sha256_per_frame"][63]),
    ]
    sc, meshes = synth.sponza_synth(1.0)
    sponza = S.build_scene(sc, extra_meshes=meshes, sequential_mesh_indices=True)
    recs.append(run_mean("C4 sponza_synth (265 k triangles) 1080p 16 spp 4 bounces, running mean", sponza, 1920, 1080, 16, 4))
    recs.append(run_sum("C5 sponza_synth 3840x2160 256 spp 4 bounces, all samples on one GPU", sponza, 3840, 2160, 256, 4))
    for r in recs:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
