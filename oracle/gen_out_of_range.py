#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- what the reference shader does with child words that point outside the
arrays (build container only).

Takes the gally fixture, damages its tree the way tests/util.py:damaged_gally_bvh does, renders it
with the reference's compute.glsl on llvmpipe (oracle/_ref/llvmpipe_runner) and stores the image
hashes of its images and of its first-hit record in tests/golden/fuzz/damaged_gally.json; the same
for material indices beyond the material array (damaged_materials.json) and for uniforms at the
edges of their ranges (edge_uniforms.json: fractional and very large bounce counts, frame numbers
beyond 2^24), and for axis-parallel rays whose origin lies exactly on child-box planes
(axis_rays.json: NaN plane distances).  Pins the rule the oracle and the
HIP path follow for such reads: zero words, as GL robust buffer access delivers them.

  python oracle/gen_out_of_range.py [--threads N]
"""
from __future__ import annotations

import argparse
import json
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))

import gen_fixtures as G  # noqa: E402
import oracle as O  # noqa: E402
from util import damaged_gally_bvh, damaged_gally_materials  # noqa: E402

W, H, BOUNCES, F0, F1 = 320, 180, 4, 3, 5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    src = G.GOLD / "gally"
    out = G.GOLD / "fuzz" / "damaged_gally.json"
    bvh, changed = damaged_gally_bvh()
    with tempfile.TemporaryDirectory(prefix="rtoor_") as td:
        sdir = Path(td)
        for name in ("tris.bin", "spheres.bin", "mats.bin", "uniforms.bin"):
            shutil.copy(src / name, sdir / name)
        (sdir / "bvh.bin").write_bytes(bvh.tobytes())
        rec = {"w": W, "h": H, "bounces": BOUNCES, "frames": [F0, F1], "changed_words": changed,
               "bvh_sha256": G.sha(bvh), "kinds": {}}
        for kind in ("color", "primary"):
            img, info = G.run_shader(kind, sdir, W, H, BOUNCES, F0, F1, args.threads)
            rec["kinds"][kind] = {"sha256_per_frame": [G.sha(img[i]) for i in range(img.shape[0])],
                                  "llvmpipe_ms": info["ms"], "threads": args.threads}
            if kind == "primary":
                code, tbits, exists = O.primary_from_golden(img[0])
                rec["primary_sha256_first_frame"] = {"code": G.sha(code), "t_bits": G.sha(tbits), "exists": G.sha(exists)}
    out.write_text(json.dumps(rec, indent=1))
    print(json.dumps(rec, indent=1))

    # material indices beyond the material array
    tris, sph = damaged_gally_materials()
    with tempfile.TemporaryDirectory(prefix="rtoor_") as td:
        sdir = Path(td)
        for name in ("bvh.bin", "mats.bin", "uniforms.bin"):
            shutil.copy(src / name, sdir / name)
        (sdir / "tris.bin").write_bytes(tris.tobytes())
        (sdir / "spheres.bin").write_bytes(sph.tobytes())
        img, info = G.run_shader("color", sdir, W, H, BOUNCES, F0, F1, args.threads)
        rec = {"w": W, "h": H, "bounces": BOUNCES, "frames": [F0, F1], "tris_sha256": G.sha(tris),
               "spheres_sha256": G.sha(sph), "sha256_per_frame": [G.sha(img[i]) for i in range(img.shape[0])]}
    (G.GOLD / "fuzz" / "damaged_materials.json").write_text(json.dumps(rec, indent=1))
    print(json.dumps(rec, indent=1))

    # uniforms at the edges: a fractional bounce count, one above the shader's own depth cap, frame
    # numbers where float32 no longer holds every integer
    runs = []
    for scene, bounces, f0, f1 in (("debug_box", 2.5, 0, 2), ("debug_box", 0.5, 4, 5), ("prims", 70.0, 1, 2),
                                   ("debug_box", 3.0, 16777215, 16777219), ("gally", 4.0, 1000000, 1000002)):
        img, info = G.run_shader("color", G.GOLD / scene, 160, 88, bounces, f0, f1, args.threads)
        runs.append({"scene": scene, "w": 160, "h": 88, "bounces": bounces, "frames": [f0, f1],
                     "sha256_per_frame": [G.sha(img[i]) for i in range(img.shape[0])]})
    (G.GOLD / "fuzz" / "edge_uniforms.json").write_text(json.dumps(runs, indent=1))
    print(json.dumps(runs, indent=1))

    # rays along an axis from an origin that lies exactly on child-box planes: (plane - O) * (1 / D)
    # is 0 * inf = NaN there, and what min / max make of a NaN decides whether the child is visited.
    # Identity for the inverse view-projection gives every pixel the direction (0, 0, 1).
    u0 = np.fromfile(src / "uniforms.bin", dtype=np.float32)
    nodes = np.fromfile(src / "bvh.bin", dtype=np.uint32).reshape(-1, 16)
    smin, sext = u0[19:22], u0[22:25]

    def plane(q, a):  # compute.glsl:244-261
        return np.float32(smin[a] + np.float32(np.float32(q) / np.float32(65535.0)) * sext[a])

    cands = [(j, c) for j in range(nodes.shape[0]) for c in range(4)
             if nodes[j][12 + c] != 0xFFFFFFFF and (nodes[j][c] >> 16) != (nodes[j][c] & 0xFFFF)
             and (nodes[j][4 + c] >> 16) != (nodes[j][4 + c] & 0xFFFF)]
    rng = np.random.default_rng(5)
    rays = []
    for trial in range(8):
        j, c = cands[int(rng.integers(0, len(cands)))]
        w = nodes[j]
        ex = plane(int(w[c] & 0xFFFF), 0) if trial % 2 == 0 else plane(int(w[c] >> 16), 0)
        ey = (plane(int(w[4 + c] & 0xFFFF), 1) if trial % 4 >= 2
              else np.float32((plane(int(w[4 + c] & 0xFFFF), 1) + plane(int(w[4 + c] >> 16), 1)) / 2))
        u = u0.copy()
        u[0:16] = np.eye(4, dtype=np.float32).reshape(-1)
        u[16:19] = [ex, ey, np.float32(smin[2] - 3.0)]
        with tempfile.TemporaryDirectory(prefix="rtoor_") as td:
            sdir = Path(td)
            for name in ("bvh.bin", "tris.bin", "spheres.bin", "mats.bin"):
                shutil.copy(src / name, sdir / name)
            u.tofile(sdir / "uniforms.bin")
            img, _ = G.run_shader("color", sdir, 64, 32, 3, 0, 1, args.threads)
            prim, _ = G.run_shader("primary", sdir, 64, 32, 3, 0, 1, args.threads)
        code, tbits, exists = O.primary_from_golden(prim[0])
        rays.append({"node": j, "child": c, "eye_words": ["%08x" % x for x in u[16:19].view(np.uint32)], "w": 64, "h": 32,
                     "bounces": 3, "frame": 0, "sample_sha256": G.sha(img[0]), "first_hits": int(exists.sum()),
                     "primary_sha256": {"code": G.sha(code), "t_bits": G.sha(tbits)}})
    (G.GOLD / "fuzz" / "axis_rays.json").write_text(json.dumps(rays, indent=1))
    print(json.dumps(rays, indent=1)[:600])


if __name__ == "__main__":
    main()
