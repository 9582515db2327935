"""The headless front end (python -m raytracer_amd): a scene file of the reference's schema with a
mesh, rendered to a PNG on one GPU and by two rank processes, against the CPU oracle's frames folded
with the reference's running mean."""

import json
import os
import struct
import subprocess
import sys
import zlib
from pathlib import Path

import numpy as np
import pytest

import oracle as O
from raytracer_amd import png, scene as S
from raytracer_amd.camera import Camera
from util import GOLDEN

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

SCENE = """
[variables]
white = [1.0, 1.0, 1.0]
r = 0.4

[[materials]]
color = "{white}"
emission = [1.0, 1.0, 1.0, 12.0]

[[materials]]
color = [0.8, 0.3, 0.2]
smoothness = 0.3

[[materials]]
color = "{white}"
smoothness = 1.0
transmission = 1.0
ior = 1.5

[[spheres]]
center = [0.0, 0.4, 1.5]
radius = "{r}"
material = 2

[[quads]]
center = [0.0, 3.0, 0.0]
normal = [0.0, -1.0, 0.0]
width = 5
height = 5
material = 0

[[quads]]
center = [0.0, 0.0, 0.0]
normal = [0.0, 1.0, 0.0]
width = 12
height = 12
material = 1

[[meshes]]
path = "./models/thing.obj"
material = 1
"""


def _read_png(path):
    data = Path(path).read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h = struct.unpack(">II", body[:8])
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = zlib.decompress(idat)
    rows = [raw[y * (1 + 3 * w) + 1:(y + 1) * (1 + 3 * w)] for y in range(h)]   # filter type 0 on every row
    return np.frombuffer(b"".join(rows), np.uint8).reshape(h, w, 3)


def test_cli_renders_a_scene_file_to_png(tmp_path):
    w, h, spp, b = 320, 180, 6, 4
    (tmp_path / "models").mkdir()
    mesh = np.load(GOLDEN / "meshes" / "box.npz")["triangles"] * np.float32(0.5) + np.float32([0.0, 0.5, -1.0])
    S.write_obj(tmp_path / "models" / "other.obj", [(mesh, 1)])
    (tmp_path / "scene.toml").write_text(SCENE)
    cmd = [sys.executable, "-m", "raytracer_amd", "scene.toml", "--out", "one.png", "--width", str(w), "--height", str(h),
           "--spp", str(spp), "--bounces", str(b), "--mesh", "./models/thing.obj=./models/other.obj"]
    env = dict(os.environ, PYTHONPATH=str(ROOT))
    p = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    info = json.loads(p.stdout.strip().splitlines()[-1])
    assert info["spp"] == spp and info["gpus"] == 1 and info["stack_overflows"] == 0 and info["rays"] > w * h * spp
    # the same scene through the host pipeline and the oracle, frames folded as the reference folds them
    import tomli

    bufs = S.build_scene(tomli.loads(SCENE), mesh_override={"./models/thing.obj": mesh})
    inv_vp, eye = Camera().uniforms_f32()
    u = np.concatenate([inv_vp, eye, bufs.scene_min, bufs.scene_extent]).astype(np.float32)
    sc = O.OracleScene(bvh=bufs.bvh.tobytes(), tris=bufs.triangles.tobytes(), spheres=bufs.spheres.tobytes(),
                       mats=bufs.materials.tobytes(), uniforms=u)
    acc = np.zeros((h, w, 4), np.float32)
    for f in range(spp):
        s = sc.render(w, h, f, b)
        acc = O.accumulate(s, acc, f)
    want = png.to_rgb8(acc)
    got = _read_png(tmp_path / "one.png")
    assert got.shape == (h, w, 3) and np.array_equal(got, want)
    assert want.max() > 200 and (want.sum(-1) > 0).mean() > 0.2      # an actual picture (no sky: the background is black)
    # two rank processes (sharing this GPU: collectives through host files), samples dealt round-robin
    env2 = dict(env, RT_BENCH_REHEARSE="1")
    p = subprocess.run(cmd[:5] + ["two.png"] + cmd[6:] + ["--gpus", "2"], cwd=tmp_path, env=env2, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    info2 = json.loads(p.stdout.strip().splitlines()[-1])
    assert info2["gpus"] == 2 and info2["rays"] == info["rays"]
    two = _read_png(tmp_path / "two.png")
    # the mean of the linear samples against the running mean with its sRGB round trips: a grey level at most
    assert np.abs(two.astype(int) - got.astype(int)).max() <= 1
