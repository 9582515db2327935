#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — random scenes pinned against the reference shader (build container only).

Draws small random scenes in the reference's TOML schema (spheres, quads, boxes, planes, materials
over their whole parameter ranges, cameras placed anywhere including inside objects), builds their
buffers with the reference's own classes, renders them with the reference's compute.glsl on
llvmpipe and stores, per case, the scene description, the 25 uniform words and the image hashes
(plus the small images of the first frame) in tests/golden/fuzz/.  The tests rebuild the buffers
with the host pipeline and compare oracle and HIP output against these.

  python oracle/gen_fuzz.py [--cases N] [--seed S]
"""
from __future__ import annotations

import argparse
import json
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

import gen_fixtures as G  # noqa: E402
import refscene  # noqa: E402
from raytracer_amd.camera import Camera  # noqa: E402

OUT = G.GOLD / "fuzz"
W, H = 96, 64


def r3(rng, lo, hi):
    return [round(float(x), 3) for x in rng.uniform(lo, hi, 3)]


def random_scene(rng: np.random.Generator, case: int) -> tuple[dict, dict]:
    mats = [{"color": [1.0, 1.0, 1.0], "emission": [1.0, 1.0, 1.0, float(rng.integers(2, 20))]}]
    for _ in range(int(rng.integers(3, 9))):
        m = {"color": r3(rng, 0.0, 1.0)}
        kind = rng.integers(0, 4)
        if kind == 0:
            m["emission"] = r3(rng, 0.0, 1.0) + [float(rng.integers(0, 300))]
        # the whole range, and values outside it (the packer clamps): smoothness 1 = mirror branch
        m["smoothness"] = float(rng.choice([0.0, 1.0, round(float(rng.uniform(-0.2, 1.2)), 3)]))
        m["metallic"] = float(rng.choice([0.0, 1.0, round(float(rng.uniform(0, 1)), 3)]))
        m["transmission"] = float(rng.choice([0.0, 1.0, round(float(rng.uniform(0, 1)), 3)]))
        m["ior"] = float(rng.choice([1.0, 1.5, round(float(rng.uniform(0.0, 3.5)), 3)]))
        mats.append(m)
    nm = len(mats)
    scene = {"title": f"fuzz {case}", "materials": mats, "spheres": [], "quads": [], "boxes": [], "planes": []}
    for _ in range(int(rng.integers(1, 10))):
        scene["spheres"].append({"center": r3(rng, -4, 4), "radius": round(float(rng.uniform(0.05, 1.5)), 3),
                                 "material": int(rng.integers(0, nm))})
    for _ in range(int(rng.integers(1, 8))):
        n = rng.normal(size=3)
        if rng.random() < 0.5:  # axis-aligned walls are the common case and the degenerate one for the slab test
            n = np.eye(3)[rng.integers(0, 3)] * rng.choice([-1.0, 1.0])
        scene["quads"].append({"center": r3(rng, -5, 5), "width": round(float(rng.uniform(0.5, 12)), 2),
                               "height": round(float(rng.uniform(0.5, 12)), 2), "normal": [float(x) for x in n],
                               "material": int(rng.integers(0, nm))})
    for _ in range(int(rng.integers(0, 5))):
        p0 = np.array(r3(rng, -4, 4))
        scene["boxes"].append({"p0": [float(x) for x in p0], "p1": [float(x) for x in p0 + rng.uniform(-2, 2, 3).round(3)],
                               "material": int(rng.integers(0, nm))})
    for _ in range(int(rng.integers(0, 3))):
        n = np.eye(3)[rng.integers(0, 3)] * rng.choice([-1.0, 1.0]) if rng.random() < 0.7 else rng.normal(size=3)
        n = n / np.linalg.norm(n)
        scene["planes"].append({"point": r3(rng, -6, 6), "normal": [float(x) for x in n], "material": int(rng.integers(0, nm))})
    if rng.random() < 0.6:  # an emissive shell around everything, so that most paths end lit
        scene["spheres"].append({"center": [0.0, 0.0, 0.0], "radius": 40.0, "material": 0})
    cam = {"position": r3(rng, -6, 6), "yaw": float(rng.choice([0.0, 90.0, 180.0, 270.0, round(float(rng.uniform(0, 360)), 2)])),
           "pitch": float(rng.choice([0.0, round(float(rng.uniform(-80, 80)), 2)]))}
    if rng.random() < 0.25 and scene["spheres"]:  # eye inside a sphere
        cam["position"] = [float(x) for x in scene["spheres"][0]["center"]]
    return scene, cam


def camera_uniforms(cam: dict):
    c = Camera()
    c.position = np.array(cam["position"], dtype=np.float64)
    c.yaw, c.pitch = cam["yaw"], cam["pitch"]
    c._update_camera_vectors()
    return c.uniforms_f32()


def dense_scene(rng: np.random.Generator, case: int, n_spheres: int, n_boxes: int, n_quads: int) -> tuple[dict, dict]:
    """Hundreds of small spheres and boxes (deep trees whose leaves are mostly not triangles: every
    position of a sphere record within its 128-byte line, box leaves at every depth)."""
    scene, cam = random_scene(rng, case)
    nm = len(scene["materials"])
    scene["spheres"] = [s for s in scene["spheres"] if s["radius"] >= 40.0]  # keep the emissive shell if drawn
    scene["boxes"], scene["quads"] = [], scene["quads"][:n_quads]
    for _ in range(n_spheres):
        scene["spheres"].append({"center": r3(rng, -6, 6), "radius": round(float(rng.uniform(0.05, 0.6)), 3),
                                 "material": int(rng.integers(0, nm))})
    for _ in range(n_boxes):
        p0 = np.array(r3(rng, -6, 6))
        scene["boxes"].append({"p0": [float(x) for x in p0], "p1": [float(x) for x in p0 + rng.uniform(0.05, 0.9, 3).round(3)],
                               "material": int(rng.integers(0, nm))})
    return scene, cam


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=24)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--odd-materials", action="store_true",
                    help="add cases whose materials lie far outside their meaningful ranges (ior 0, negative, 1e30, ...)")
    ap.add_argument("--odd-geometry", action="store_true",
                    help="add cases with degenerate primitives (zero-area quads, radius 0 and negative, coincident copies, ...)")
    ap.add_argument("--accumulate", action="store_true",
                    help="running mean over 6 frames of the odd-material cases (infinite and NaN samples in the accumulate pass)")
    ap.add_argument("--tiny", action="store_true",
                    help="add scenes of one, two and three primitives (the one-primitive tree has an all-empty root)")
    ap.add_argument("--dense", action="store_true",
                    help="add the dense sphere / box cases after the existing ones instead of redrawing those")
    args = ap.parse_args()
    OUT.mkdir(parents=True, exist_ok=True)
    if args.tiny:
        first = len(sorted(OUT.glob("case*.json")))
        firsts = dict(np.load(OUT / "first_frames.npz"))
        rng = np.random.default_rng(args.seed + 4)
        for k, n in enumerate((1, 2, 3, 5)):
            case = first + k
            scene, cam = random_scene(rng, case)
            scene["quads"], scene["boxes"] = [], []
            scene["spheres"] = [{"center": r3(rng, -2, 2), "radius": round(float(rng.uniform(0.5, 1.5)), 3), "material": i % len(scene["materials"])}
                                for i in range(n)]
            scene["planes"] = [{"point": [0.0, -2.0, 0.0], "normal": [0.0, 1.0, 0.0], "material": 0}]
            cam = {"position": [0.0, 0.5, 7.0], "yaw": 270.0, "pitch": 0.0}
            write_case(case, scene, cam, 4, firsts, args.threads)
        np.savez_compressed(OUT / "first_frames.npz", **firsts)
        return
    if args.accumulate:
        out = {}
        for path in sorted(OUT.glob("case*.json")):
            rec = json.loads(path.read_text())
            if not any(m.get("ior") in (0.0, -1.5, 1e30, 1e-30) for m in rec["scene"]["materials"]):
                continue
            b = refscene.build_reference_scene(rec["scene"], [])
            uni = np.array([int(x, 16) for x in rec["uniform_words"]], dtype=np.uint32).view(np.float32)
            with tempfile.TemporaryDirectory(prefix="rtfuzz_") as td:
                td = Path(td)
                for key in ("mats", "spheres", "tris", "bvh", "boxes", "planes"):
                    (td / f"{key}.bin").write_bytes(b[key])
                uni.tofile(td / "uniforms.bin")
                img, acc, _ = G.run_shader("color", td, W, H, rec["bounces"], 0, 6, args.threads, accumulate=True)
            out[path.stem] = {"frames": [0, 6], "sample_sha256_per_frame": [G.sha(img[i]) for i in range(6)],
                              "acc_sha256_per_frame": [G.sha(acc[i]) for i in range(6)],
                              "nonfinite_samples": int((~np.isfinite(img)).sum()), "nonfinite_acc": int((~np.isfinite(acc)).sum()),
                              "max_finite_sample": float(np.nanmax(np.where(np.isfinite(img), img, 0)))}
            print(path.stem, out[path.stem]["nonfinite_samples"], out[path.stem]["nonfinite_acc"], out[path.stem]["max_finite_sample"], flush=True)
        (OUT / "odd_accumulate.json").write_text(json.dumps(out, indent=1))
        return
    if args.odd_materials:
        first = len(sorted(OUT.glob("case*.json")))
        firsts = dict(np.load(OUT / "first_frames.npz"))
        rng = np.random.default_rng(args.seed + 2)
        odd_ior = [0.0, -1.5, 1e30, 1e-30, 1.0, 0.5]
        for k in range(4):
            case = first + k
            scene, cam = random_scene(rng, case)
            for i, m in enumerate(scene["materials"][1:]):
                m["ior"] = odd_ior[(i + k) % len(odd_ior)]
                m["transmission"] = float(rng.choice([1.0, 0.5, m.get("transmission", 0.0)]))
                if (i + k) % 3 == 0:
                    m["color"] = [round(float(x), 3) for x in rng.uniform(-0.5, 3.0, 3)]
                if (i + k) % 4 == 1:
                    m["emission"] = [round(float(x), 3) for x in rng.uniform(-1.0, 1.0, 3)] + [float(rng.choice([-5.0, 1e20, 0.0]))]
            write_case(case, scene, cam, int(rng.integers(3, 8)), firsts, args.threads)
        np.savez_compressed(OUT / "first_frames.npz", **firsts)
        return
    if args.odd_geometry:
        first = len(sorted(OUT.glob("case*.json")))
        firsts = dict(np.load(OUT / "first_frames.npz"))
        rng = np.random.default_rng(args.seed + 3)
        for k in range(4):
            case = first + k
            scene, cam = random_scene(rng, case)
            nm = len(scene["materials"])
            # zero-area quads (their triangles get NaN normals), a sphere of radius 0 and one of negative radius
            scene["quads"].append({"center": r3(rng, -3, 3), "width": 0.0, "height": 2.0, "normal": [0.0, 1.0, 0.0],
                                   "material": int(rng.integers(0, nm))})
            scene["quads"].append({"center": r3(rng, -3, 3), "width": 1.5, "height": 0.0, "normal": [1.0, 0.0, 0.0],
                                   "material": int(rng.integers(0, nm))})
            scene["spheres"].append({"center": r3(rng, -3, 3), "radius": 0.0, "material": int(rng.integers(0, nm))})
            scene["spheres"].append({"center": r3(rng, -3, 3), "radius": -0.7, "material": int(rng.integers(0, nm))})
            # coincident copies: every split of them is a tie
            twin = {"center": r3(rng, -3, 3), "radius": 0.8, "material": int(rng.integers(0, nm))}
            scene["spheres"] += [dict(twin) for _ in range(5)]
            q = {"center": r3(rng, -4, 4), "width": 3.0, "height": 2.0, "normal": [0.0, 0.0, 1.0], "material": int(rng.integers(0, nm))}
            scene["quads"] += [dict(q) for _ in range(3)]
            # a box of no volume and a plane through the eye
            p0 = r3(rng, -3, 3)
            scene["boxes"].append({"p0": p0, "p1": list(p0), "material": int(rng.integers(0, nm))})
            scene["planes"].append({"point": list(cam["position"]), "normal": [0.0, 1.0, 0.0], "material": int(rng.integers(0, nm))})
            write_case(case, scene, cam, int(rng.integers(3, 8)), firsts, args.threads)
        np.savez_compressed(OUT / "first_frames.npz", **firsts)
        return
    if args.dense:
        first = len(sorted(OUT.glob("case*.json")))
        firsts = dict(np.load(OUT / "first_frames.npz"))
        rng = np.random.default_rng(args.seed + 1)
        for k, (ns, nb, nq) in enumerate(((300, 0, 2), (0, 250, 2), (200, 150, 3), (500, 300, 0))):
            case = first + k
            scene, cam = dense_scene(rng, case, ns, nb, nq)
            write_case(case, scene, cam, int(rng.integers(3, 7)), firsts, args.threads)
        np.savez_compressed(OUT / "first_frames.npz", **firsts)
        return
    rng = np.random.default_rng(args.seed)
    firsts = {}
    for case in range(args.cases):
        scene, cam = random_scene(rng, case)
        write_case(case, scene, cam, int(rng.integers(2, 9)), firsts, args.threads)
    np.savez_compressed(OUT / "first_frames.npz", **firsts)


def write_case(case, scene, cam, bounces, firsts, threads):
    """Buffers from the reference's classes, three frames and the first-hit record from the reference
    shader on llvmpipe; the record goes to caseNN.json, the first frame's images into `firsts`."""
    b = refscene.build_reference_scene(scene, [])
    inv_vp, eye = camera_uniforms(cam)
    uni = np.concatenate([inv_vp, eye, b["scene_min"], b["scene_extent"]]).astype(np.float32)
    with tempfile.TemporaryDirectory(prefix="rtfuzz_") as td:
        td = Path(td)
        for key in ("mats", "spheres", "tris", "bvh", "boxes", "planes"):
            (td / f"{key}.bin").write_bytes(b[key])
        uni.tofile(td / "uniforms.bin")
        img, _ = G.run_shader("color", td, W, H, bounces, 0, 3, threads)
        prim, _ = G.run_shader("primary", td, W, H, bounces, 0, 1, threads)
    rec = {"scene": scene, "camera": cam, "uniform_words": ["%08x" % x for x in uni.view(np.uint32)],
           "w": W, "h": H, "bounces": bounces, "frames": [0, 3],
           "buffers_sha256": {k: G.sha(b[k]) for k in ("mats", "spheres", "tris", "bvh", "boxes", "planes")},
           "sample_sha256_per_frame": [G.sha(img[i]) for i in range(3)],
           "lit_fraction": float((img[0, :, :, :3].sum(-1) > 0).mean())}
    (OUT / f"case{case:02d}.json").write_text(json.dumps(rec, indent=1))
    firsts[f"color{case:02d}"] = img[0]
    firsts[f"primary{case:02d}"] = prim[0]
    print(case, bounces, "lit", round(rec["lit_fraction"], 3), flush=True)


if __name__ == "__main__":
    main()
