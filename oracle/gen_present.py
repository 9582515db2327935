#!/usr/bin/env python3
"""Pins the headless consumer of the accumulation image (raytracer_amd/png.py) on the reference's own
present pass.  TEST INFRASTRUCTURE: runs in the build container only (needs /root/reference and Mesa).

An RGBA32F image of probe values -- every 8-bit code, the rounding boundaries between codes, values
outside [0, 1], infinities, NaN, and random values -- is uploaded the way a host uploads a mapped
accumulation image (glTexSubImage2D) and drawn by the reference's upscale_vertex.glsl /
upscale_fragment.glsl (read where they lie, never copied) into an RGBA8 framebuffer on llvmpipe
(oracle/_ref/llvmpipe_runner --present-*).  Input and resulting bytes go to
tests/golden/present/present_256x16.npz; tests/test_host_utils.py holds png.to_rgb8 to them.

    python3 oracle/gen_present.py
"""
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/shaders")
RUNNER = ROOT / "oracle" / "_ref" / "llvmpipe_runner"
OUT = ROOT / "tests" / "golden" / "present" / "present_256x16.npz"


def probe_image() -> np.ndarray:
    w, h = 256, 16
    rng = np.random.default_rng(20261112)
    img = rng.random((h, w, 4), dtype=np.float32)
    i = np.arange(w, dtype=np.float64)
    rows = [i / 255.0, (i + 0.5) / 255.0, (i + 0.499) / 255.0, (i + 0.501) / 255.0, np.linspace(-0.5, 1.5, w)]
    for r, v in enumerate(rows):
        img[r, :, 0] = v.astype(np.float32)
        img[r, :, 1] = np.nextafter(v.astype(np.float32), np.float32(2.0))
        img[r, :, 2] = np.nextafter(v.astype(np.float32), np.float32(-2.0))
    one = np.float32(1.0)
    special = np.array([np.nan, np.inf, -np.inf, -0.0, 0.0, 1e-45, -1e-45, 1e-38, one, np.nextafter(one, np.float32(2)),
                        np.nextafter(one, np.float32(0)), 2.0, 17.7, 1e30, -1e30, 0.5, 0.5 / 255, 254.5 / 255], np.float32)
    img[5, :, :3] = np.resize(special, (w, 3))
    return img


def main() -> None:
    if not RUNNER.exists() or not REF.exists():
        sys.exit("needs oracle/_ref/llvmpipe_runner (make -C oracle ref) and /root/reference")
    img = probe_image()
    h, w = img.shape[:2]
    with tempfile.TemporaryDirectory() as td:
        src, dst = Path(td) / "in.bin", Path(td) / "out.bin"
        src.write_bytes(img.tobytes())
        r = subprocess.run([str(RUNNER), "--present-vs", str(REF / "upscale_vertex.glsl"), "--present-fs",
                            str(REF / "upscale_fragment.glsl"), "--present-in", str(src), "--present-out", str(dst),
                            "--w", str(w), "--h", str(h)], capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr)
        print(r.stdout.strip())
        out = np.frombuffer(dst.read_bytes(), np.uint8).reshape(h, w, 4).copy()
    OUT.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT, image=img, framebuffer=out)
    print(OUT, "written;", "alpha all 255:", bool((out[..., 3] == 255).all()))


if __name__ == "__main__":
    main()
