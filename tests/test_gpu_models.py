"""Readiness for real models: a Sponza-scale OBJ through the same path a user-supplied
`sponza_full.obj` would take (bench.py --obj), and trees deeper than the reference shader's own
32-entry stack (compute.glsl:230-232,324), where the reference is undefined and this library is not."""

import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle as O
from util import spine_tree

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_obj_file_of_the_stand_in_scene_renders_through_bench(tmp_path):
    """sponza_synth written as OBJ text (groups, usemtl, quads, v/vt/vn tokens, negative indices),
    read back to the same triangles, and rendered by `bench.py --obj`, whose line reports the
    deepest stack and the dropped pushes (none)."""
    from raytracer_amd import scene as S, synth

    _, meshes = synth.sponza_synth(0.25)
    g = meshes[0][0]
    n = g.shape[0] // 2
    meshes = meshes + [(np.stack([g[:n], g[n:]], 1).reshape(-1, 3, 3) + np.float32(0.001), 2)]  # stored quad by quad
    path = tmp_path / "sponza_synth.obj"
    nfaces = S.write_obj(path, meshes)
    want = np.concatenate([m for m, _ in meshes])
    assert nfaces <= want.shape[0] - n           # that mesh's triangle pairs became quads (and a few fans elsewhere)
    text = path.read_text()
    assert "usemtl" in text and "\ng mesh" in text and "//1" in text and "/1/1" in text and " -1" in text
    got = S.read_obj(path)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--obj", str(path), "--steps", "2", "--warmup", "1",
                        "--width", "640", "--height", "360", "--no-cpu-baseline", "--sustain-seconds", "0"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["workload"].startswith("obj:sponza_synth.obj")
    assert out["config"]["triangles"] >= want.shape[0]
    assert 1 <= out["max_stack"] <= 32 and out["stack_overflows"] == 0
    assert out["value"] > 0 and out["roofline"]["per_ray"]["triangle_tests"] > 0


@pytest.mark.parametrize("pipeline", ["mega", "mega-leafsteps", "wave", "regroup"])
def test_trees_deeper_than_the_reference_stack(pipeline):
    """Stack depths 37 and 64 (the reference shader's array has 32 entries and no guard: undefined
    there) are rendered exactly as an unbounded stack would (the oracle's holds 256), with no push
    dropped; at depth 67 the pushes beyond 64 entries are dropped and counted, and the context
    goes on working."""
    import os

    from raytracer_amd.renderer import Renderer

    os.environ["RT_PIPELINE"] = pipeline.split("-")[0]
    if pipeline.startswith("mega"):
        os.environ["RT_LEAF_STEPS"] = "1" if pipeline.endswith("leafsteps") else "0"
    try:
        r = Renderer(device=0, singleton=False)
    finally:
        os.environ.pop("RT_PIPELINE", None)
        os.environ.pop("RT_LEAF_STEPS", None)
    w, h, b = 160, 88, 3
    try:
        r.update_size(w, h)
        r.bounces = b
        for depth, want_stack in ((12, 37), (21, 64), (22, 67)):
            sph, nodes, mats, u = spine_tree(depth)
            r.update_ssbo(r.materials, mats, 9)
            r.update_ssbo(r.spheres, sph, 5)
            r.update_ssbo(r.triangles, np.zeros(0, np.uint8), 12)
            r.update_ssbo(r.bvh, nodes, 11)
            r.set_uniforms(u[0:16], u[16:19])
            r.scene_min_coords, r.scene_extent = u[19:22], u[22:25]
            sc = O.OracleScene(bvh=nodes.tobytes(), spheres=sph.tobytes(), mats=mats.tobytes(), uniforms=u)
            ref, cnt = sc.render(w, h, 0, b, want_counters=True)
            assert cnt["max_stack"] == want_stack and cnt["stack_over_32"] == 1
            r.set_flags(stats=True)
            r.reset_stats()
            r.set_frame_index(0)
            r.camera.changed = False
            r.render()
            img = r.read_sample()
            s = r.stats()
            if want_stack <= 64:
                assert s["stack_overflows"] == 0 and s["max_stack"] == want_stack
                assert s["nodes"] == cnt["nodes"] and s["sphere_tests"] == cnt["sphere_tests"]
                assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
            else:
                assert s["stack_overflows"] > 0 and s["max_stack"] == 64
            r.set_flags()
    finally:
        r.close()
