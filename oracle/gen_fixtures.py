#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/ from the reference itself.

Runs only in the build container (needs /root/reference and Mesa llvmpipe):

  1. scene buffers: the reference's own Python classes pack materials, spheres,
     triangles and the flattened 4-wide BVH (oracle/refscene.py);
  2. images: the reference's shaders/compute.glsl (plus the one-line strict-GLSL
     fix, and two instrumented text-substituted variants) is executed on
     llvmpipe by oracle/_ref/llvmpipe_runner.

What is committed are DATA only: input byte buffers, uniform words, output
images (small ones whole, large ones as SHA-256 + checksums).

  python oracle/gen_fixtures.py [--only NAME] [--threads N]
"""

from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np
import tomli

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

import refscene  # noqa: E402
from raytracer_amd.camera import Camera  # noqa: E402

GOLD = ROOT / "tests" / "golden"
RUNNER = ROOT / "oracle" / "_ref" / "llvmpipe_runner"
REF = refscene.REF


def sha(a: np.ndarray | bytes) -> str:
    b = a if isinstance(a, bytes) else np.ascontiguousarray(a).tobytes()
    return hashlib.sha256(b).hexdigest()


def run_shader(kind, sdir: Path, w, h, bounces, f0, f1, threads, reps=1, accumulate=False, recreate_at=None):
    """kind: a refscene.shader_variant name, or ("capture", bounce, expr).
    accumulate: also run the reference's screen shaders after every frame; returns (img, acc, info).
    recreate_at: the textures are recreated before that frame, as Renderer.update_size does."""
    with tempfile.TemporaryDirectory(prefix="rtgold_") as td:
        sp = Path(td) / "variant.glsl"
        if isinstance(kind, tuple):
            sp.write_text(refscene.shader_capture_variant(kind[1], kind[2]))
        else:
            sp.write_text(refscene.shader_variant(kind))
        out = Path(td) / "out.bin"
        cmd = [str(RUNNER), "--shader", str(sp), "--w", str(w), "--h", str(h), "--bounces", str(bounces),
               "--frames", str(f0), str(f1), "--uniforms", str(sdir / "uniforms.bin"), "--out", str(out),
               "--reps", str(reps)]
        acc_out = Path(td) / "acc.bin"
        if accumulate:
            cmd += ["--screen-vs", str(REF / "shaders" / "screen_vertex.glsl"),
                    "--screen-fs", str(REF / "shaders" / "screen_fragment.glsl"), "--acc-out", str(acc_out)]
            if recreate_at is not None:
                cmd += ["--recreate-at", str(recreate_at)]
        for flag, name in (("--bvh", "bvh.bin"), ("--tris", "tris.bin"), ("--spheres", "spheres.bin"),
                           ("--mats", "mats.bin"), ("--boxes", "boxes.bin"), ("--planes", "planes.bin")):
            if (sdir / name).exists() and (sdir / name).stat().st_size:
                cmd += [flag, str(sdir / name)]
        env = dict(os.environ, LP_NUM_THREADS=str(threads))
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, check=False)
        if r.returncode != 0:
            raise RuntimeError(f"runner failed: {r.stderr}")
        info = json.loads(r.stdout.strip().splitlines()[-1])
        img = np.fromfile(out, dtype=np.float32).reshape(f1 - f0, h, w, 4)
        if accumulate:
            return img, np.fromfile(acc_out, dtype=np.float32).reshape(f1 - f0, h, w, 4), info
        return img, info


def write_scene(name: str, toml_text: str, meshes, camera: Camera | None = None, keep_buffers=True) -> Path:
    sdir = GOLD / name
    sdir.mkdir(parents=True, exist_ok=True)
    scene = tomli.loads(toml_text)
    mesh_tris = [(refscene.read_obj_triangles(REF / "models" / m), mat) for m, mat in meshes]
    # inputs as data, so that the scene pipeline can be tested where the reference is absent
    (sdir / "scene.json").write_text(json.dumps(scene, indent=1))
    (GOLD / "meshes").mkdir(exist_ok=True)
    for (m, _), (arr, _) in zip(meshes, mesh_tris):
        np.savez_compressed(GOLD / "meshes" / (Path(m).stem + ".npz"), triangles=arr)
    b = refscene.build_reference_scene(scene, mesh_tris)
    cam = camera or Camera()
    inv_vp, eye = cam.uniforms_f32()
    uni = np.concatenate([inv_vp, eye, b["scene_min"], b["scene_extent"]]).astype(np.float32)
    meta = {"name": name, "meshes": meshes, "sha256": {}, "nbytes": {}}
    for key in ("mats", "spheres", "tris", "bvh", "boxes", "planes"):
        if key in ("boxes", "planes") and not b[key]:
            continue
        meta["sha256"][key] = sha(b[key])
        meta["nbytes"][key] = len(b[key])
        (sdir / f"{key}.bin").write_bytes(b[key])
    (sdir / "uniforms.bin").write_bytes(uni.tobytes())
    meta["uniform_words"] = ["%08x" % x for x in uni.view(np.uint32)]
    meta["keep_buffers"] = keep_buffers
    (sdir / "meta.json").write_text(json.dumps(meta, indent=1))
    return sdir


def golden_frames(sdir: Path, tag, w, h, bounces, f0, f1, threads, store_full: bool, kinds=("color",)):
    """Runs the reference shader and records outputs under sdir/<tag>.*"""
    rec = {"w": w, "h": h, "bounces": bounces, "frames": [f0, f1], "kinds": {}}
    for kind in kinds:
        img, info = run_shader(kind, sdir, w, h, bounces, f0, f1, threads)
        k = {"sha256_per_frame": [sha(img[i]) for i in range(img.shape[0])],
             "sum_rgb_per_frame": [float(img[i, :, :, :3].astype(np.float64).sum()) for i in range(img.shape[0])],
             "llvmpipe_ms": info["ms"], "threads": threads}
        if kind == "counters":
            nodes = img[..., 0].astype(np.float64).sum()
            tri = img[..., 1].astype(np.float64).sum()
            sph = img[..., 2].astype(np.float64).sum()
            wv = img[..., 3].astype(np.int64)
            rays = (wv // 64).sum()
            k.update({"rays": int(rays), "nodes": int(nodes), "tri_tests": int(tri), "sphere_tests": int(sph),
                      "max_stack": int((wv % 64).max())})
        if store_full:
            np.save(sdir / f"{tag}.{kind}.npy", img)
        rec["kinds"][kind] = k
    (sdir / f"{tag}.json").write_text(json.dumps(rec, indent=1))
    return rec


STATE_EXPRS = {"dx": "ray.direction.x", "dy": "ray.direction.y", "dz": "ray.direction.z",
               "ox": "ray.origin.x", "oy": "ray.origin.y", "oz": "ray.origin.z", "thr_x": "color.x"}


def golden_state(sdir: Path, tag, w, h, bounces, frame, threads, capture_bounces=(0, 1, 2)):
    """Per-bounce ray state (direction, origin, throughput.x) witnessed through the alpha channel."""
    arrays = {}
    for bnc in capture_bounces:
        for key, expr in STATE_EXPRS.items():
            img, _ = run_shader(("capture", bnc, expr), sdir, w, h, bounces, frame, frame + 1, threads)
            arrays[f"b{bnc}_{key}"] = img[0, :, :, 3].copy()
            arrays.setdefault("rgb", img[0, :, :, :3].copy())
    np.savez_compressed(sdir / f"{tag}.state.npz", **arrays)
    (sdir / f"{tag}.state.json").write_text(json.dumps(
        {"w": w, "h": h, "bounces": bounces, "frame": frame, "capture_bounces": list(capture_bounces),
         "exprs": STATE_EXPRS}, indent=1))


def golden_accumulation(sdir: Path, tag, w, h, bounces, nframes, threads, keep=(0, 1), recreate_at=None):
    """The reference's running mean (screen_fragment.glsl:94-108 driven as renderer.py:249-288 drives
    it) after each of nframes frames: hashes of all, full images of the frames in `keep` and the last.
    recreate_at: Renderer.update_size (same size) is called before that frame (renderer.py:347-382)."""
    img, acc, _ = run_shader("color", sdir, w, h, bounces, 0, nframes, threads, accumulate=True,
                             recreate_at=recreate_at)
    keep = sorted(set(keep) | {nframes - 1})
    rec = {"w": w, "h": h, "bounces": bounces, "frames": [0, nframes], "kept": keep,
           **({"update_size_before_frame": recreate_at} if recreate_at is not None else {}),
           "sample_sha256_per_frame": [sha(img[i]) for i in range(nframes)],
           "acc_sha256_per_frame": [sha(acc[i]) for i in range(nframes)]}
    np.savez_compressed(sdir / f"{tag}.acc.npz", **{f"acc{i}": acc[i] for i in keep})
    (sdir / f"{tag}.acc.json").write_text(json.dumps(rec, indent=1))
    return rec


def prims_toml() -> str:
    """Boxes and planes: the kernel's other two primitive kinds (compute.glsl:315-323, 788-806), which
    neither reference scene uses.  Same TOML schema; [[boxes]] as scene.py:125-147 reads them
    (p0, p1, material), [[planes]] by analogy (point, normal, material)."""
    out = ['title = "boxes and planes"', ""]
    mats = ["color = [1.0, 1.0, 1.0]\nemission = [1.0, 1.0, 1.0, 9.0]",
            "color = [0.8, 0.8, 0.8]",
            "color = [0.9, 0.3, 0.2]\nsmoothness = 0.4",
            "color = [0.2, 0.5, 0.9]\nsmoothness = 0.9\nmetallic = 0.7",
            "color = [0.9, 0.9, 0.9]\nsmoothness = 1.0\ntransmission = 0.9\nior = 1.5",
            "color = [0.3, 0.8, 0.4]\nsmoothness = 0.2",
            "color = [0.7, 0.7, 0.2]\nsmoothness = 0.6\nmetallic = 0.3",
            "color = [1.0, 0.8, 0.6]\nemission = [1.0, 0.8, 0.6, 3.0]"]
    for m in mats:
        out.append("[[materials]]\n" + m + "\n")
    out.append("[[spheres]]\ncenter = [-1.5, 0.6, 0.3]\nradius = 0.6\nmaterial = 4\n")
    out.append("[[spheres]]\ncenter = [0.5, 0.35, -2.2]\nradius = 0.35\nmaterial = 3\n")
    boxes = [([-3.5, 0.0, -1.6], [-2.3, 1.4, -0.6], 2), ([-0.4, 0.0, 1.2], [0.9, 0.8, 2.6], 3),
             ([-2.8, 0.0, 1.9], [-2.0, 2.2, 2.7], 4), ([1.6, 1.0, -0.5], [0.8, 0.0, 0.4], 5),
             ([-5.0, 0.0, -3.5], [-4.2, 3.0, 3.5], 6), ([-1.0, 2.4, -1.0], [0.2, 2.6, 0.2], 7)]
    for p0, p1, m in boxes:
        out.append(f"[[boxes]]\np0 = {p0}\np1 = {p1}\nmaterial = {m}\n")
    out.append("[[quads]]\ncenter = [-1.0, 5.0, 0.0]\nwidth = 9\nheight = 9\nnormal = [0.0, -1.0, 0.0]\nmaterial = 0\n")
    out.append("[[quads]]\ncenter = [-2.0, 2.0, -6.0]\nwidth = 12\nheight = 6\nnormal = [0.0, 0.0, 1.0]\nmaterial = 5\n")
    out.append("[[planes]]\npoint = [0.0, 0.0, 0.0]\nnormal = [0.0, 1.0, 0.0]\nmaterial = 1\n")
    out.append("[[planes]]\npoint = [-8.0, 0.0, 0.0]\nnormal = [1.0, 0.0, 0.0]\nmaterial = 6\n")
    return "\n".join(out)


def test_toml_with_mesh() -> str:
    """scenes/test.toml minus its [[meshes]] entry (which names a file that exists nowhere)."""
    text = (REF / "scenes" / "test.toml").read_text()
    i = text.index("[[meshes]]")
    return text[:i]


def matlab_toml() -> str:
    """A materials laboratory: same TOML schema as the reference scenes, but with fractional
    metallic / smoothness / transmission / ior / colour values, which the reference's own two
    scenes never use (they only have 0, 0.5 and 1)."""
    out = ['title = "materials lab"', ""]
    mats = []
    # 0: white emitter, 1: white diffuse, 2..4: coloured emitters
    mats.append("color = [1.0, 1.0, 1.0]\nemission = [1.0, 1.0, 1.0, 12.0]")
    mats.append("color = [0.8, 0.8, 0.8]")
    mats.append("color = [1.0, 0.2, 0.1]\nemission = [1.0, 0.3, 0.2, 5.0]")
    mats.append("color = [0.1, 1.0, 0.3]\nemission = [0.2, 1.0, 0.4, 4.0]")
    mats.append("color = [0.2, 0.3, 1.0]\nemission = [0.25, 0.35, 1.0, 6.0]")
    # wall paints
    mats.append("color = [0.75, 0.25, 0.2]\nsmoothness = 0.15")
    mats.append("color = [0.2, 0.7, 0.3]\nsmoothness = 0.35\nmetallic = 0.2")
    mats.append("color = [0.3, 0.35, 0.8]\nsmoothness = 0.6")
    mats.append("color = [0.9, 0.85, 0.7]\nsmoothness = 0.9\nmetallic = 0.5")
    k = 0
    lab = []
    for metallic in (0.0, 0.25, 0.6, 1.0):
        for smooth in (0.0, 0.3, 0.75, 1.0):
            for trans in (0.0, 0.5, 1.0):
                ior = (1.0, 1.33, 1.5, 2.4)[k % 4]
                col = [round(0.15 + 0.8 * ((k * 7) % 11) / 10.0, 3), round(0.15 + 0.8 * ((k * 3) % 13) / 12.0, 3),
                       round(0.15 + 0.8 * ((k * 5) % 7) / 6.0, 3)]
                lab.append(f"color = {col}\nsmoothness = {smooth}\nmetallic = {metallic}\n"
                           f"transmission = {trans}\nior = {ior}")
                k += 1
    first_lab = len(mats)
    mats += lab
    for m in mats:
        out.append("[[materials]]\n" + m + "\n")
    # 8 x 6 grid of spheres facing the camera, one lab material each
    n = 0
    for row in range(6):
        for col in range(8):
            out.append(f"[[spheres]]\ncenter = [-2.0, {-0.4 + row * 0.9:.2f}, {-3.15 + col * 0.9:.2f}]\n"
                       f"radius = 0.4\nmaterial = {first_lab + n}\n")
            n += 1
    quads = [
        ([0.0, -1.0, 0.0], 20, 16, [0.0, 1.0, 0.0], 8),     # floor (semi-metal)
        ([0.0, 8.0, 0.0], 20, 16, [0.0, -1.0, 0.0], 1),     # ceiling
        ([-10.0, 3.5, 0.0], 16, 9, [1.0, 0.0, 0.0], 5),     # far wall
        ([10.0, 3.5, 0.0], 16, 9, [-1.0, 0.0, 0.0], 7),     # wall behind the camera
        ([0.0, 3.5, 8.0], 20, 9, [0.0, 0.0, -1.0], 6),
        ([0.0, 3.5, -8.0], 20, 9, [0.0, 0.0, 1.0], 1),
        ([0.0, 7.9999, 0.0], 4, 4, [0.0, -1.0, 0.0], 0),    # ceiling light
        ([-9.99, 1.0, -5.0], 1.5, 1.5, [1.0, 0.0, 0.0], 2),
        ([-9.99, 1.0, 5.0], 1.5, 1.5, [1.0, 0.0, 0.0], 3),
        ([4.0, 7.999, 5.0], 2, 2, [0.0, -1.0, 0.0], 4),
    ]
    for c, w, h, nrm, m in quads:
        out.append(f"[[quads]]\ncenter = {c}\nwidth = {w}\nheight = {h}\nnormal = {nrm}\nmaterial = {m}\n")
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    if not RUNNER.exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "ref"], check=True)
    T = args.threads

    def want(n):
        return args.only is None or args.only == n

    debug_toml = (REF / "scenes" / "debug.toml").read_text()
    if want("debug"):
        s = write_scene("debug", debug_toml, [])
        golden_frames(s, "c1_640x360_b2", 640, 360, 2, 0, 3, T, store_full=False, kinds=("color", "primary", "counters"))
        golden_frames(s, "small_160x88_b2", 160, 88, 2, 0, 2, T, store_full=True, kinds=("color", "primary"))
        golden_frames(s, "small_160x88_b6", 160, 88, 6, 3, 5, T, store_full=True, kinds=("color",))
    if want("debug_box"):
        s = write_scene("debug_box", debug_toml, [("box.obj", 1)])
        golden_frames(s, "c1_640x360_b2", 640, 360, 2, 0, 2, T, store_full=False, kinds=("color", "primary", "counters"))
        golden_frames(s, "small_160x88_b4", 160, 88, 4, 0, 2, T, store_full=True, kinds=("color", "primary"))
    if want("gally"):
        s = write_scene("gally", test_toml_with_mesh(), [("gally.obj", 26)])
        golden_frames(s, "small_160x88_b4", 160, 88, 4, 0, 2, T, store_full=True, kinds=("color", "primary", "counters"))
        golden_frames(s, "small_160x88_b8", 160, 88, 8, 5, 6, T, store_full=True, kinds=("color",))
        golden_state(s, "state_160x88_b4_f2", 160, 88, 4, 2, T)
        golden_frames(s, "mid_640x360_b4", 640, 360, 4, 0, 4, T, store_full=False, kinds=("color", "primary", "counters"))
        golden_frames(s, "c3_1920x1080_b4", 1920, 1080, 4, 0, 1, T, store_full=False, kinds=("color", "primary", "counters"))
        # 64-spp mean in linear space for the radiance tolerance test
        img, info = run_shader("color", s, 320, 180, 4, 0, 64, T)
        rec = {"w": 320, "h": 180, "bounces": 4, "frames": [0, 64],
               "sha256_per_frame": [sha(img[i]) for i in range(64)], "llvmpipe_ms": info["ms"], "threads": T}
        (s / "spp64_320x180_b4.json").write_text(json.dumps(rec, indent=1))
        np.save(s / "spp64_320x180_b4.first.npy", img[0])
    if want("matlab"):
        toml = matlab_toml()
        s = write_scene("matlab", toml, [("gally.obj", 9 + 29), ("box.obj", 9 + 14)])
        golden_frames(s, "small_160x88_b6", 160, 88, 6, 0, 2, T, store_full=True, kinds=("color", "primary"))
        golden_frames(s, "mid_640x360_b5", 640, 360, 5, 0, 3, T, store_full=False, kinds=("color", "primary", "counters"))
        golden_state(s, "state_160x88_b4_f1", 160, 88, 4, 1, T)
    if want("accum"):
        for name in ("debug", "gally"):
            golden_accumulation(GOLD / name, "accum_160x88_b4", 160, 88, 4, 12, T)
    if want("ssbo_args"):
        # what the reference loader passes to Renderer.update_ssbo, dtype and shape included
        # (structured records; spheres and triangles arrive as (n, 1) arrays: scene.py:121,191)
        b = refscene.build_reference_scene(tomli.loads(debug_toml), [])
        np.savez(GOLD / "debug" / "update_ssbo_args.npz", **b["update_ssbo_args"])
    if want("resize"):
        # Renderer.update_size in the middle of an accumulation: five frames, the textures recreated,
        # three more frames -- frame 5 is blended into the new (black) texture with weight 1/6
        golden_accumulation(GOLD / "gally", "resize_160x88_b4", 160, 88, 4, 8, T, keep=(4, 5), recreate_at=5)
    if want("prims"):
        s = write_scene("prims", prims_toml(), [])
        golden_frames(s, "small_160x88_b5", 160, 88, 5, 0, 2, T, store_full=True, kinds=("color", "primary", "counters"))
        golden_frames(s, "mid_640x360_b4", 640, 360, 4, 0, 3, T, store_full=False, kinds=("color", "primary", "counters"))
        golden_state(s, "state_160x88_b4_f1", 160, 88, 4, 1, T)
        golden_accumulation(s, "accum_160x88_b4", 160, 88, 4, 6, T)
    if want("oob"):
        # gally with damaged indices: some leaves point past the triangle buffer, some inner
        # children past the node buffer.  GL robust buffer access makes such reads return zeros;
        # the product clamps them to an all-zero record.  No cycles are introduced.
        src = GOLD / "gally"
        s = GOLD / "gally_oob"
        s.mkdir(exist_ok=True)
        for f in ("mats.bin", "spheres.bin", "tris.bin", "uniforms.bin"):
            (s / f).write_bytes((src / f).read_bytes())
        bvh = np.frombuffer((src / "bvh.bin").read_bytes(), dtype=np.uint32).reshape(-1, 16).copy()
        n_nodes = bvh.shape[0]
        n_tris = (src / "tris.bin").stat().st_size // 64
        rng = np.random.default_rng(7)
        meta = bvh[:, 12:16]
        typ, idx = meta >> 30, meta & 0x3FFFFFFF
        leaf = np.argwhere(typ == 0)
        for r, c in leaf[rng.choice(len(leaf), size=len(leaf) // 25, replace=False)]:
            meta[r, c] = (0 << 30) | (n_tris + int(rng.integers(0, 1 << 20)))
        inner = np.argwhere((typ == 3) & (idx < n_nodes) & (np.arange(n_nodes)[:, None] > 20))
        for r, c in inner[rng.choice(len(inner), size=len(inner) // 40, replace=False)]:
            meta[r, c] = (3 << 30) | (n_nodes + int(rng.integers(0, 1 << 16)))
        (s / "bvh.bin").write_bytes(bvh.tobytes())
        (s / "meta.json").write_text(json.dumps({"name": "gally_oob", "derived_from": "gally", "damaged_leaves":
                                                 len(leaf) // 25, "damaged_inner": len(inner) // 40}, indent=1))
        golden_frames(s, "small_160x88_b4", 160, 88, 4, 0, 2, T, store_full=True, kinds=("color", "primary"))
        golden_frames(s, "mid_640x360_b4", 640, 360, 4, 0, 2, T, store_full=False, kinds=("color", "primary"))
    if want("configs"):
        # the full-size single-GPU configurations: per-frame hashes of every sample image, and of the
        # running mean after every frame (hash only; nothing of this size is stored whole)
        s = GOLD / "gally"
        img, acc, info = run_shader("color", s, 1920, 1080, 4, 0, 64, T, accumulate=True)
        (s / "c3full_1920x1080_b4.acc.json").write_text(json.dumps(
            {"w": 1920, "h": 1080, "bounces": 4, "frames": [0, 64], "kept": [],
             "sample_sha256_per_frame": [sha(img[i]) for i in range(64)],
             "acc_sha256_per_frame": [sha(acc[i]) for i in range(64)], "llvmpipe_ms": info["ms"], "threads": T},
            indent=1))
        del img, acc
        sph = np.load(GOLD / "meshes" / "sphere.npz")["triangles"]
        with tempfile.TemporaryDirectory(prefix="rtc2_") as td:
            b = refscene.build_reference_scene(tomli.loads(test_toml_with_mesh()), [(sph, 26)])
            for key in ("mats", "spheres", "tris", "bvh"):
                (Path(td) / f"{key}.bin").write_bytes(b[key])
            inv_vp, eye = Camera().uniforms_f32()
            np.concatenate([inv_vp, eye, b["scene_min"], b["scene_extent"]]).astype(np.float32).tofile(
                Path(td) / "uniforms.bin")
            img, acc, info = run_shader("color", Path(td), 1920, 1080, 4, 0, 16, T, accumulate=True)
        (GOLD / "c2_sphere_obj_1920x1080_b4.json").write_text(json.dumps(
            {"scene": "gally/scene.json + meshes/sphere.npz (material 26)", "w": 1920, "h": 1080, "bounces": 4,
             "frames": [0, 16], "sample_sha256_per_frame": [sha(img[i]) for i in range(16)],
             "acc_sha256_per_frame": [sha(acc[i]) for i in range(16)], "llvmpipe_ms": info["ms"], "threads": T},
            indent=1))
    if want("sponza"):
        # the bench workload itself (configuration C4 with the stand-in scene): hashes of the reference
        # shader's 1080p output and its llvmpipe frame time; the 25 MB of buffers are not stored
        # (tests rebuild them with the host pipeline, whose byte parity "big" pins)
        from raytracer_amd import synth
        scene, meshes = synth.sponza_synth(1.0)
        from raytracer_amd import scene as S
        b = S.build_scene(scene, extra_meshes=meshes, sequential_mesh_indices=True)
        with tempfile.TemporaryDirectory(prefix="rtc4_") as td:
            td = Path(td)
            (td / "mats.bin").write_bytes(b.materials.tobytes())
            (td / "spheres.bin").write_bytes(b.spheres.tobytes())
            (td / "tris.bin").write_bytes(b.triangles.tobytes())
            (td / "bvh.bin").write_bytes(b.bvh.tobytes())
            inv_vp, eye = Camera().uniforms_f32()
            np.concatenate([inv_vp, eye, b.scene_min, b.scene_extent]).astype(np.float32).tofile(td / "uniforms.bin")
            rec = {"scene": "sponza_synth(detail=1.0), sequential mesh indices, default camera", "w": 1920, "h": 1080,
                   "bounces": 4, "frames": [0, 3], "buffers_sha256": {"mats": sha(b.materials), "spheres": sha(b.spheres),
                                                                     "tris": sha(b.triangles), "bvh": sha(b.bvh)}}
            img, info = run_shader("color", td, 1920, 1080, 4, 0, 3, T)
            rec["sample_sha256_per_frame"] = [sha(img[i]) for i in range(3)]
            rec["llvmpipe_ms"] = info["ms"]
            rec["threads"] = T
            prim, _ = run_shader("primary", td, 1920, 1080, 4, 0, 1, T)
            code, tbits, exists = __import__("oracle").primary_from_golden(prim[0])
            rec["primary_sha256"] = {"code": sha(code), "t_bits": sha(tbits), "exists": sha(exists)}
            cnt, _ = run_shader("counters", td, 1920, 1080, 4, 0, 1, T)
            rec["rays_frame0"] = int((cnt[0, :, :, 3].astype(np.int64) // 64).sum())
        (GOLD / "c4_sponza_synth_1920x1080_b4.json").write_text(json.dumps(rec, indent=1))
    if args.only == "sponza4k":
        # configuration C5's image size (3840x2160) on the same scene: two whole frames of the reference
        # shader, hashes only (about a minute of llvmpipe per frame; not part of the default run)
        from raytracer_amd import synth
        from raytracer_amd import scene as S
        scene, meshes = synth.sponza_synth(1.0)
        b = S.build_scene(scene, extra_meshes=meshes, sequential_mesh_indices=True)
        with tempfile.TemporaryDirectory(prefix="rtc5_") as td:
            td = Path(td)
            (td / "mats.bin").write_bytes(b.materials.tobytes())
            (td / "spheres.bin").write_bytes(b.spheres.tobytes())
            (td / "tris.bin").write_bytes(b.triangles.tobytes())
            (td / "bvh.bin").write_bytes(b.bvh.tobytes())
            inv_vp, eye = Camera().uniforms_f32()
            np.concatenate([inv_vp, eye, b.scene_min, b.scene_extent]).astype(np.float32).tofile(td / "uniforms.bin")
            rec = {"scene": "sponza_synth(detail=1.0), sequential mesh indices, default camera", "w": 3840, "h": 2160,
                   "bounces": 4, "frames": [0, 2], "buffers_sha256": {"mats": sha(b.materials), "spheres": sha(b.spheres),
                                                                     "tris": sha(b.triangles), "bvh": sha(b.bvh)}}
            img, info = run_shader("color", td, 3840, 2160, 4, 0, 2, T)
            rec["sample_sha256_per_frame"] = [sha(img[i]) for i in range(2)]
            rec["sum_rgb_per_frame"] = [float(img[i, :, :, :3].astype(np.float64).sum()) for i in range(2)]
            rec["llvmpipe_ms"] = info["ms"]
            rec["threads"] = T
        (GOLD / "c5_sponza_synth_3840x2160_b4.json").write_text(json.dumps(rec, indent=1))
    if want("big"):
        # byte-level pins for the native scene pipeline on large inputs: only hashes are kept
        from raytracer_amd import synth
        big = {}
        scene, meshes = synth.sponza_synth(1.0)
        b = refscene.build_reference_scene(scene, meshes)
        big["sponza_synth"] = {"detail": 1.0, "triangles": len(b["tris"]) // 64, "nodes": len(b["bvh"]) // 64,
                               "sha256": {k: sha(b[k]) for k in ("mats", "spheres", "tris", "bvh")},
                               "scene_min": [float(x) for x in b["scene_min"]],
                               "scene_extent": [float(x) for x in b["scene_extent"]]}
        scene = tomli.loads(test_toml_with_mesh())
        sph_tris = refscene.read_obj_triangles(REF / "models" / "sphere.obj")
        np.savez_compressed(GOLD / "meshes" / "sphere.npz", triangles=sph_tris)
        b = refscene.build_reference_scene(scene, [(sph_tris, 26)])
        big["test_sphere_obj"] = {"triangles": len(b["tris"]) // 64, "nodes": len(b["bvh"]) // 64,
                                  "sha256": {k: sha(b[k]) for k in ("mats", "spheres", "tris", "bvh")},
                                  "scene_min": [float(x) for x in b["scene_min"]],
                                  "scene_extent": [float(x) for x in b["scene_extent"]]}
        (GOLD / "big_scene_hashes.json").write_text(json.dumps(big, indent=1))
    print("done")


if __name__ == "__main__":
    main()
