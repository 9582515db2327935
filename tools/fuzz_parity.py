#!/usr/bin/env python3
"""A long run of the random-scene parity check (the one tests/test_gpu_parity.py makes on 60 scenes):
  tools/fuzz_parity.py [--cases N] [--seed S] [--out FILE]          (run on the GPU box)
Every case draws a scene (sparse: all four primitive kinds, camera anywhere; every sixth dense: hundreds
of small spheres and boxes), builds it with the host pipeline, renders one frame on the GPU with each of
the two traversal variants and compares image bits, first hit, end-of-path state and work counters with
the CPU oracle.  Prints one JSON line with the number of cases and the mismatches found."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))
import gen_fuzz as F  # noqa: E402  scene generator only
import oracle as O  # noqa: E402  the CPU oracle (checker only)
from raytracer_amd import scene as S  # noqa: E402
from raytracer_amd.camera import Camera  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=20261111)
    ap.add_argument("--out", default=None)
    ap.add_argument("--fast", action="store_true",
                    help="also the opt-in fast mode: against the oracle's traversal of the tree the library built, and "
                         "its first hits against the default mode's (none lost, none farther)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    r = Renderer(device=0, camera=Camera(), singleton=False)
    w, h = 96, 64
    r.update_size(w, h)
    bad, t0, rays, nearer = [], time.time(), 0, 0
    for case in range(args.cases):
        if case % 6 == 5:
            scene, cam = F.dense_scene(rng, 100000 + case, int(rng.integers(50, 400)), int(rng.integers(0, 200)), 3)
        else:
            scene, cam = F.random_scene(rng, 100000 + case)
        b = S.build_scene(scene)
        inv_vp, eye = F.camera_uniforms(cam)
        b.upload(r)
        r.set_uniforms(inv_vp, eye)
        u = np.concatenate([inv_vp, eye, b.scene_min, b.scene_extent]).astype(np.float32)
        sc = O.OracleScene(bvh=b.bvh.tobytes(), tris=b.triangles.tobytes(), spheres=b.spheres.tobytes(),
                           mats=b.materials.tobytes(), uniforms=u, boxes=b.boxes.tobytes(), planes=b.planes.tobytes())
        bounces, frame = int(rng.integers(1, 9)), int(rng.integers(0, 100000))
        ref, raux, cnt = sc.render(w, h, frame, bounces, want_primary=True, want_counters=True)
        for variant in ("unified", "leaf steps"):
            r.set_traversal_variant(variant)
            r.reset_stats()
            r.set_flags(aux=True, stats=True)
            r.bounces = bounces
            r.set_frame_index(frame)
            r.camera.changed = False
            r.render()
            img, aux, st = r.read_sample(), r.read_aux(), r.stats()
            if not np.array_equal(img.view(np.uint32), ref.view(np.uint32)):
                bad.append((case, variant, "image"))
            if not np.array_equal(aux, raux):
                bad.append((case, variant, "state"))
            if any(st[k] != cnt[k] for k in ("rays", "nodes", "tri_tests", "sphere_tests", "box_tests", "max_stack")):
                bad.append((case, variant, "counters"))
            if variant == "unified":
                paux = aux
        if args.fast:
            r.set_traversal_variant(None)
            r.fast_mode = True
            try:
                r.reset_stats()
                r.set_flags(aux=True, stats=True)
                r.set_frame_index(frame)
                r.camera.changed = False
                r.render()
                img, aux, st, nodes = r.read_sample(), r.read_aux(), r.stats(), r.fast_bvh()
            finally:
                r.fast_mode = False
            try:
                O.set_fast_bvh(nodes)
                fref, fraux, fcnt = sc.render(w, h, frame, bounces, want_primary=True, want_counters=True)
            finally:
                O.set_fast_bvh(None)
            if not (np.array_equal(img.view(np.uint32), fref.view(np.uint32)) and np.array_equal(aux, fraux)):
                bad.append((case, "fast", "oracle"))
            if any(st[k] != fcnt[k] for k in ("rays", "nodes", "tri_tests", "sphere_tests", "box_tests")):
                bad.append((case, "fast", "counters"))
            tr, tf = paux[..., 1].view(np.float32), aux[..., 1].view(np.float32)
            hp, hf = paux[..., 2] == 1, aux[..., 2] == 1
            both = hp & hf
            if np.any(hp & ~hf) or np.any(tf[both] > tr[both] + 2e-6 * np.maximum(1.0, tr[both])):
                bad.append((case, "fast", "first hit lost or farther"))
            nearer += int(np.count_nonzero(hf & ~hp)) + int(np.count_nonzero(tf[both] < tr[both] - 2e-6 * np.maximum(1.0, tr[both])))
        rays += int(cnt["rays"])
        if case % 100 == 99:
            print(f"{case + 1} cases, {len(bad)} mismatches, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    r.close()
    line = json.dumps({"cases": args.cases, "seed": args.seed, "image": [w, h], "variants": ["unified", "leaf steps"],
                       "rays_checked_per_variant": rays, "mismatches": bad[:50], "n_mismatches": len(bad),
                       "fast_mode_checked": bool(args.fast), "fast_mode_pixels_with_a_nearer_first_hit": nearer,
                       "seconds": round(time.time() - t0, 1)})
    print(line)
    if args.out:
        Path(args.out).write_text(line + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
