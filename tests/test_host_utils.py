"""Host-side pieces around the render path that need no GPU: the PNG consumer of the accumulation
image and the camera controls that trigger an accumulation reset."""

import struct
import zlib
from pathlib import Path

import numpy as np
import pytest

from raytracer_amd import png
from raytracer_amd.camera import Camera

ROOT = Path(__file__).resolve().parents[1]


def _decode_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(tag + body) & 0xFFFFFFFF
        chunks.setdefault(tag, b"")
        chunks[tag] += body
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (depth, ctype) == (8, 2)
    raw = zlib.decompress(chunks[b"IDAT"])
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + 3 * w)
    assert np.all(rows[:, 0] == 0)  # filter type none
    return rows[:, 1:].reshape(h, w, 3)


def test_png_is_clipped_quantised_and_upright(tmp_path):
    h, w = 5, 7
    img = np.zeros((h, w, 4), np.float32)
    img[0, :, 0] = 1.0           # bottom row of the GL image: red
    img[h - 1, :, 2] = 2.5       # top row: blue, beyond 1 (the accumulation image is unclamped)
    img[2, 3, :3] = [0.5, -1.0, np.nan]
    img[..., 3] = 1.0
    p = tmp_path / "a.png"
    png.save_png(p, img)
    out = _decode_png(p)
    assert out.shape == (h, w, 3)
    assert np.all(out[h - 1, :, 0] == 255) and np.all(out[h - 1, :, 1:] == 0)  # red row ends up at the bottom of the file
    assert np.all(out[0, :, 2] == 255)
    assert tuple(out[h - 1 - 2, 3]) == (128, 0, 0)


def test_png_quantisation_is_the_reference_present_pass_on_llvmpipe(tmp_path):
    """The reference shows the accumulation texture by drawing it with its upscale shaders into the
    window's 8-bit framebuffer (renderer.py:290-300, upscale_fragment.glsl:8-17).  The fixture holds
    what that pass produces on Mesa llvmpipe for a probe image -- every code, the ties between codes,
    out-of-range values, infinities, NaN, random values (oracle/gen_present.py) -- uploaded the way a
    host uploads a mapped accumulation image; the PNG writer must produce the same bytes."""
    d = np.load(Path(__file__).parent / "golden" / "present" / "present_256x16.npz")
    img, fb = d["image"], d["framebuffer"]
    assert np.all(fb[..., 3] == 255)
    assert np.array_equal(png.to_rgb8(img)[::-1], fb[..., :3])   # to_rgb8 returns rows top-down
    p = tmp_path / "probe.png"
    png.save_png(p, img)
    assert np.array_equal(_decode_png(p)[::-1], fb[..., :3])


def test_camera_controls_flag_a_change_and_move_as_the_reference_does():
    c = Camera()
    c.changed = False
    inv0, eye0 = c.uniforms_f32()
    c.forward(0.5)
    assert c.changed
    _, eye1 = c.uniforms_f32()
    np.testing.assert_allclose(eye1 - eye0, np.float32(0.5) * c.front.astype(np.float32), atol=1e-6)
    c.changed = False
    c.turn(100.0, 2000.0)        # pitch is clamped to +-89 degrees (camera.py:129-141)
    assert c.changed and c.pitch == 89.0 and abs(c.yaw - 190.0) < 1e-9
    c.turn(0.0, -5000.0)
    assert c.pitch == -89.0
    c.changed = False
    c.strafe(1.0)
    c.rise(-0.25)
    assert c.changed
    inv2, _ = c.uniforms_f32()
    assert inv2.dtype == np.float32 and inv2.shape == (16,) and not np.array_equal(inv2, inv0)


def test_host_tree_builders_under_address_and_ub_sanitizers(tmp_path):
    """scene_builder.cpp and rt_fastbvh.cpp (host code only) built with -fsanitize=address,undefined
    and run on random and degenerate primitive sets (tools/sanitize/host_builders.cpp): no report,
    every primitive in exactly one leaf, every child index in range."""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "sanitize_host_builders"
    src = [ROOT / "tools" / "sanitize" / "host_builders.cpp", ROOT / "raytracer_amd" / "csrc" / "scene_builder.cpp",
           ROOT / "raytracer_amd" / "csrc" / "rt_fastbvh.cpp"]
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            *map(str, src), "-o", str(exe)], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    assert "0 failed" in run.stdout


def test_obj_writer_and_reader_round_trip(tmp_path):
    """write_obj / read_obj: every face syntax (plain, v/vt/vn, v//vn, negative indices), polygons as
    fans, groups and usemtl lines skipped; the triangles come back bit for bit in file order."""
    from raytracer_amd import scene as S, synth

    _, meshes = synth.sponza_synth(0.02)
    g = meshes[0][0]
    n = g.shape[0] // 2
    meshes = meshes[:9] + [(np.stack([g[:n], g[n:]], 1).reshape(-1, 3, 3), 2)]
    path = tmp_path / "m.obj"
    nfaces = S.write_obj(path, meshes)
    want = np.concatenate([m for m, _ in meshes])
    got = S.read_obj(path)
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert nfaces < want.shape[0]
    # a pentagon with texture indices and a trailing comment line
    (tmp_path / "p.obj").write_text("o pent\nv 0 0 0\nv 1 0 0\nv 1.5 1 0\nv 0.5 2 0\nv -0.5 1 0\nvt 0 0\n"
                                    "s off\nf 1/1 2/1 3/1 4/1 5/1\n# end\n")
    tri = S.read_obj(tmp_path / "p.obj")
    assert tri.shape == (3, 3, 3) and np.array_equal(tri[:, 0], np.zeros((3, 3), np.float32))
    assert np.array_equal(tri[2, 2], np.array([-0.5, 1, 0], np.float32))
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(ValueError):
        S.read_obj(tmp_path / "bad.obj")


def test_isa_checker_flags_what_it_is_there_for():
    """tools/check_isa.py on fabricated listings: the shape the compiler produces today passes; a copy
    at the control-flow merge, a use after too early a wait, a spill, or arms that load into
    different registers are reported.  (The check of the real code runs at build time.)"""
    import sys

    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    import check_isa as c

    good = """\t;;#ASMSTART
\tglobal_load_dwordx4 v[2:5], v[36:37], off
\tglobal_load_dwordx4 v[6:9], v[36:37], off offset:16
\t;;#ASMEND
\ts_andn2_saveexec_b64 s[0:1], s[4:5]
\tv_mul_lo_u32 v2, v48, s2
\t;;#ASMSTART
\tds_read_b128 v[2:5], v34
\tds_read_b128 v[6:9], v34 offset:16
\t;;#ASMEND
\tv_mov_b64_e32 v[50:51], 0
\ts_or_b64 exec, exec, s[0:1]
\t;;#ASMSTART
\ts_waitcnt vmcnt(1) lgkmcnt(1)
\t;;#ASMEND
\tv_add_f32 v10, v2, v3
\t;;#ASMSTART
\ts_waitcnt vmcnt(0) lgkmcnt(0)
\t;;#ASMEND
\tv_add_f32 v10, v6, v7""".splitlines()

    def run(lines):
        return c.check_function("k", list(enumerate(lines, 1)))

    assert run(good) == []
    bad = list(good)
    bad.insert(12, "\tv_mov_b32_e32 v20, v6")               # a copy at the merge, before any wait
    assert len(run(bad)) == 1 and "v_mov_b32_e32 v20, v6" in run(bad)[0]
    bad = list(good)
    bad[15] = "\tv_add_f32 v10, v6, v3"                     # the second vector used after the first wait only
    assert len(run(bad)) == 1
    bad = list(good)
    bad.insert(11, "\tscratch_store_dword off, v7, s32")    # a spill of a register in flight
    assert len(run(bad)) == 1
    bad = list(good)
    bad[8] = "\tds_read_b128 v[12:15], v34 offset:16"       # the arms disagree: a merge would need copies
    assert any("different registers" in m for m in run(bad))
    assert len(run(good[:12])) == 1                         # loads that are never waited for
