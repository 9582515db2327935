"""Headless consumer of the accumulation image: 8-bit PNG dump.

The reference presents the accumulation texture through its upscale pass, which with
sharpness 0 is a plain copy to the default framebuffer (reference
shaders/upscale_fragment.glsl:8-17): values are clamped to [0,1] and quantised by the
framebuffer.  ``save_png`` does the same to a file: clip, scale by 255, round, and flip rows so
that the bottom-up GL image appears upright.
"""

from __future__ import annotations

import struct
import zlib

import numpy as np


def to_rgb8(rgba: np.ndarray) -> np.ndarray:
    a = np.asarray(rgba, dtype=np.float32)[..., :3]
    a = np.nan_to_num(a, nan=0.0, posinf=1.0, neginf=0.0)
    return (np.clip(a, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)[::-1]


def save_png(path, rgba: np.ndarray) -> None:
    img = np.ascontiguousarray(to_rgb8(rgba))
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))
