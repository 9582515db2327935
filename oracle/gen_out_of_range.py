#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- what the reference shader does with child words that point outside the
arrays (build container only).

Takes the gally fixture, damages its tree the way tests/util.py:damaged_gally_bvh does, renders it
with the reference's compute.glsl on llvmpipe (oracle/_ref/llvmpipe_runner) and stores the image
hashes of its images and of its first-hit record in tests/golden/fuzz/damaged_gally.json; the same
for material indices beyond the material array (damaged_materials.json) and for uniforms at the
edges of their ranges (edge_uniforms.json: fractional and very large bounce counts, frame numbers
beyond 2^24).  Pins the rule the oracle and the
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


if __name__ == "__main__":
    main()
