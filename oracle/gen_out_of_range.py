#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- what the reference shader does with child words that point outside the
arrays (build container only).

Takes the gally fixture, damages its tree the way tests/util.py:damaged_gally_bvh does, renders it
with the reference's compute.glsl on llvmpipe (oracle/_ref/llvmpipe_runner) and stores the image
hashes of its images and of its first-hit record in tests/golden/fuzz/damaged_gally.json.  Pins the rule the oracle and the
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
from util import damaged_gally_bvh  # noqa: E402

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


if __name__ == "__main__":
    main()
