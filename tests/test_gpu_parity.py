"""HIP path (through the C ABI) against the reference shader's golden images and the CPU oracle.

Bit-for-bit: colour images, first-hit primitive ids and distances, per-pixel path state.
"""

import json

import numpy as np
import pytest

import oracle as O
from util import GOLDEN, damaged_gally_bvh, load_golden_scene, render_frame, sha

pytestmark = pytest.mark.gpu


def _cases():
    out = []
    for sdir in sorted(p for p in GOLDEN.iterdir() if p.is_dir() and p.name != "fuzz"):
        for j in sorted(sdir.glob("*.json")):
            if j.name in ("meta.json", "scene.json") or j.name.endswith((".state.json", ".acc.json")):
                continue
            out.append((sdir.name, j.stem))
    return out


@pytest.fixture(scope="module", params=["wave", "mega", "mega-leafsteps", "regroup"])
def renderer(request):
    """Every scheduling of the same arithmetic: the wavefront pipeline, the megakernel (default) in
    both of its traversal variants (unified node / leaf step, leaf steps of their own) and the
    megakernel that regroups a tile's live rays while they traverse."""
    import os

    from raytracer_amd.renderer import Renderer

    os.environ["RT_PIPELINE"] = request.param.split("-")[0]
    if request.param.startswith("mega"):
        os.environ["RT_LEAF_STEPS"] = "1" if request.param.endswith("leafsteps") else "0"
    try:
        r = Renderer(device=0, singleton=False)
    finally:
        os.environ.pop("RT_PIPELINE", None)
        os.environ.pop("RT_LEAF_STEPS", None)
    r.pipeline_name = request.param.split("-")[0]
    yield r
    r.close()


@pytest.mark.parametrize("scene,tag", _cases())
def test_colour_frames_match_reference_shader(renderer, scene, tag):
    rec = json.loads((GOLDEN / scene / f"{tag}.json").read_text())
    w, h, b = rec["w"], rec["h"], rec["bounces"]
    f0, f1 = rec["frames"]
    hashes = rec["kinds"]["color"]["sha256_per_frame"] if "kinds" in rec else rec["sha256_per_frame"]
    load_golden_scene(renderer, scene)
    bad = []
    for i, f in enumerate(range(f0, f1)):
        img = render_frame(renderer, w, h, f, b)
        if sha(img) != hashes[i]:
            bad.append(f)
    assert not bad, f"{scene}/{tag}: frames {bad} differ from the reference shader's output"


@pytest.mark.parametrize("scene,tag", [c for c in _cases() if (GOLDEN / c[0] / f"{c[1]}.primary.npy").exists()])
def test_primary_hits_match_reference_shader(renderer, scene, tag):
    rec = json.loads((GOLDEN / scene / f"{tag}.json").read_text())
    g = np.load(GOLDEN / scene / f"{tag}.primary.npy")
    w, h, b = rec["w"], rec["h"], rec["bounces"]
    f0, f1 = rec["frames"]
    load_golden_scene(renderer, scene)
    for i, f in enumerate(range(f0, f1)):
        render_frame(renderer, w, h, f, b, aux=True)
        aux = renderer.read_aux()
        code, tbits, exists = O.primary_from_golden(g[i])
        assert np.array_equal(code, aux[..., 0])
        assert np.array_equal(tbits, aux[..., 1])
        assert np.array_equal(exists, aux[..., 2])


@pytest.mark.parametrize("scene,w,h,frame,bounces", [
    ("prims", 320, 180, 2, 6),
    ("debug", 640, 360, 1, 2), ("debug_box", 320, 180, 3, 5), ("gally", 640, 360, 2, 4),
    ("gally", 256, 144, 9, 8), ("matlab", 640, 360, 1, 6), ("matlab", 1920, 1080, 0, 4),
])
def test_full_path_state_matches_oracle(renderer, scene, w, h, frame, bounces):
    """Image, first hit and end-of-path state (direction, origin, throughput, RNG, ray count)."""
    load_golden_scene(renderer, scene)
    img = render_frame(renderer, w, h, frame, bounces, aux=True, stats=True)
    aux = renderer.read_aux()
    sc = O.OracleScene.from_dir(GOLDEN / scene)
    ref, raux, cnt = sc.render(w, h, frame, bounces, want_primary=True, want_counters=True)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    for k in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15):
        assert np.array_equal(aux[..., k], raux[..., k]), f"aux word {k}"


def test_child_words_that_point_outside_the_arrays(renderer):
    """A tree whose child words name nodes and primitives beyond the arrays -- the reference's own
    empty slot (0xFFFFFFFF: node 0x3FFFFFFF) given a real box, node indices just past the end, and
    triangle indices past the end -- reads zero words there (GL robust buffer access); the image must be
    the reference shader's own (llvmpipe, oracle/gen_out_of_range.py), path state and counters the oracle's."""
    d = GOLDEN / "gally"
    bvh, changed = damaged_gally_bvh()
    assert all(v > 10 for v in changed.values()), changed
    u = load_golden_scene(renderer, "gally")
    renderer.update_ssbo(0, bvh.view(np.uint8).reshape(-1), 11)
    sc = O.OracleScene(bvh=bvh.tobytes(), tris=(d / "tris.bin").read_bytes(), spheres=(d / "spheres.bin").read_bytes(),
                       mats=(d / "mats.bin").read_bytes(), uniforms=u)
    w, h, frame, bounces = 320, 180, 3, 4
    renderer.reset_stats()
    img = render_frame(renderer, w, h, frame, bounces, aux=True, stats=True)
    aux = renderer.read_aux()
    s = renderer.stats()
    ref, raux, cnt = sc.render(w, h, frame, bounces, want_primary=True, want_counters=True)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    rec = json.loads((GOLDEN / "fuzz" / "damaged_gally.json").read_text())  # the reference shader on llvmpipe
    assert sha(bvh) == rec["bvh_sha256"] and (w, h, bounces, frame) == (rec["w"], rec["h"], rec["bounces"], rec["frames"][0])
    assert sha(img) == rec["kinds"]["color"]["sha256_per_frame"][0]
    for k in range(16):
        assert np.array_equal(aux[..., k], raux[..., k]), f"aux word {k}"
    for k in ("rays", "nodes", "tri_tests", "sphere_tests", "max_stack"):
        assert s[k] == cnt[k], k
    plain = O.OracleScene.from_dir(d).render(w, h, frame, bounces)
    assert not np.array_equal(plain, ref)  # the damage is visible: the test is not vacuous
    # and without the instrumentation (the kernels the frame loop runs)
    img2 = render_frame(renderer, w, h, frame, bounces)
    assert np.array_equal(img2.view(np.uint32), ref.view(np.uint32))


def test_material_indices_outside_the_array_and_uniforms_at_the_edges(renderer):
    """Reference shader on llvmpipe (oracle/gen_out_of_range.py): material indices beyond the
    material array read a zero material; fractional bounce counts, a count above the shader's own
    depth cap, and frame numbers beyond 2^24 (float32 holds not every integer there, and from frame
    4 294 968 on time * 1000 no longer fits a uint: the conversion saturates and all frames draw the
    same samples)."""
    from util import damaged_gally_materials

    rec = json.loads((GOLDEN / "fuzz" / "damaged_materials.json").read_text())
    tris, sph = damaged_gally_materials()
    assert sha(tris) == rec["tris_sha256"] and sha(sph) == rec["spheres_sha256"]
    load_golden_scene(renderer, "gally")
    renderer.update_ssbo(0, tris.view(np.uint8).reshape(-1), 12)
    renderer.update_ssbo(0, sph.view(np.uint8).reshape(-1), 5)
    f0, f1 = rec["frames"]
    for i, f in enumerate(range(f0, f1)):
        assert sha(render_frame(renderer, rec["w"], rec["h"], f, rec["bounces"])) == rec["sha256_per_frame"][i], f
    # axis-parallel rays from origins exactly on child-box planes (NaN plane distances)
    load_golden_scene(renderer, "gally")
    for run in json.loads((GOLDEN / "fuzz" / "axis_rays.json").read_text()):
        eye = np.array([int(x, 16) for x in run["eye_words"]], dtype=np.uint32).view(np.float32)
        renderer.set_uniforms(np.eye(4, dtype=np.float32).reshape(-1), eye)
        img = render_frame(renderer, run["w"], run["h"], run["frame"], run["bounces"])
        assert sha(img) == run["sample_sha256"], (run["node"], run["child"])
    for run in json.loads((GOLDEN / "fuzz" / "edge_uniforms.json").read_text()):
        load_golden_scene(renderer, run["scene"])
        f0, f1 = run["frames"]
        for i, f in enumerate(range(f0, f1)):
            render_frame(renderer, run["w"], run["h"], f, 1)
            # the uniform is a float and the C ABI takes it as one; the Python mirror, like the
            # reference's property, only lets integers through, so the value goes in underneath it
            renderer._bounces = run["bounces"]
            renderer.set_frame_index(f)
            renderer.render()
            assert sha(renderer.read_sample()) == run["sha256_per_frame"][i], (run["scene"], run["bounces"], f)


def test_counters_match_oracle(renderer):
    load_golden_scene(renderer, "gally")
    renderer.reset_stats()
    render_frame(renderer, 640, 360, 0, 4, stats=True)
    s = renderer.stats()
    sc = O.OracleScene.from_dir(GOLDEN / "gally")
    _, cnt = sc.render(640, 360, 0, 4, want_counters=True)
    for k in ("rays", "nodes", "tri_tests", "sphere_tests", "box_tests", "max_stack"):
        assert s[k] == cnt[k], k
    assert s["pixel_samples"] == 640 * 360
    assert s["stack_overflows"] == 0
    # boxes and planes
    load_golden_scene(renderer, "prims")
    renderer.reset_stats()
    render_frame(renderer, 640, 360, 1, 4, stats=True)
    s = renderer.stats()
    _, cnt = O.OracleScene.from_dir(GOLDEN / "prims").render(640, 360, 1, 4, want_counters=True)
    assert cnt["box_tests"] > 0
    for k in ("rays", "nodes", "tri_tests", "sphere_tests", "box_tests", "max_stack"):
        assert s[k] == cnt[k], k


def test_accumulation_matches_oracle(renderer):
    """Running mean of 6 frames (screen_fragment.glsl:94-108) and reset semantics."""
    load_golden_scene(renderer, "gally")
    w, h, b = 320, 180, 4
    renderer.update_size(w, h)
    renderer.rendered_frames = 0   # update_size leaves the count alone, as the reference's does
    sc = O.OracleScene.from_dir(GOLDEN / "gally")
    acc = np.zeros((h, w, 4), np.float32)
    for k in range(6):
        s_gpu = render_frame(renderer, w, h, k, b)
        s_ref = sc.render(w, h, k, b)
        acc = O.accumulate(s_ref, acc, k)
        a_gpu = renderer.read_accumulation()
        assert np.array_equal(s_gpu.view(np.uint32), s_ref.view(np.uint32))
        assert np.array_equal(a_gpu.view(np.uint32), acc.view(np.uint32)), f"accumulation after frame {k}"
    renderer.reset_accumulation()
    assert renderer.rendered_frames == 0
    assert np.array_equal(renderer.read_accumulation(), renderer.read_sample())


@pytest.mark.parametrize("scene,tag", [(p.parent.name, p.name[: -len(".acc.json")])
                                       for p in sorted(GOLDEN.glob("*/*.acc.json"))])
def test_accumulation_matches_reference_fragment_shader(renderer, scene, tag):
    """Running mean against what the reference's screen shaders produced on llvmpipe."""
    rec = json.loads((GOLDEN / scene / f"{tag}.acc.json").read_text())
    w, h, b = rec["w"], rec["h"], rec["bounces"]
    load_golden_scene(renderer, scene)
    renderer.update_size(w, h)
    renderer.reset_accumulation()
    for f in range(*rec["frames"]):
        if f == rec.get("update_size_before_frame"):
            renderer.update_size(w, h)   # same size: the reference recreates its textures all the same
        render_frame(renderer, w, h, f, b)
        assert sha(renderer.read_sample()) == rec["sample_sha256_per_frame"][f]
        assert sha(renderer.read_accumulation()) == rec["acc_sha256_per_frame"][f], f"accumulation after frame {f}"
    assert renderer.rendered_frames == rec["frames"][1]


def test_sphere_obj_configuration_all_frames(renderer):
    """test.toml + sphere.obj at 1920x1080, 4 bounces, 16 samples per pixel: scene from the host
    pipeline, every sample image and running mean against the reference shaders' hashes."""
    from raytracer_amd import scene as S
    from raytracer_amd.camera import Camera

    rec = json.loads((GOLDEN / "c2_sphere_obj_1920x1080_b4.json").read_text())
    b = S.build_scene(json.loads((GOLDEN / "gally" / "scene.json").read_text()),
                      extra_meshes=[(np.load(GOLDEN / "meshes" / "sphere.npz")["triangles"], 26)])
    b.upload(renderer)
    inv_vp, eye = Camera().uniforms_f32()
    renderer.set_uniforms(inv_vp, eye)
    w, h, bn = rec["w"], rec["h"], rec["bounces"]
    renderer.update_size(w, h)
    renderer.reset_accumulation()
    for f in range(*rec["frames"]):
        render_frame(renderer, w, h, f, bn)
        assert sha(renderer.read_sample()) == rec["sample_sha256_per_frame"][f], f"sample {f}"
        assert sha(renderer.read_accumulation()) == rec["acc_sha256_per_frame"][f], f"running mean after {f}"


def test_bench_workload_matches_reference_shader(renderer):
    """sponza_synth at 1920x1080, 4 bounces -- what bench.py times -- against the reference
    shader's hashes: three sample images, first-hit ids and distances of frame 0, ray count."""
    from raytracer_amd import scene as S, synth
    from raytracer_amd.camera import Camera

    rec = json.loads((GOLDEN / "c4_sponza_synth_1920x1080_b4.json").read_text())
    scene, meshes = synth.sponza_synth(1.0)
    S.build_scene(scene, extra_meshes=meshes, sequential_mesh_indices=True).upload(renderer)
    inv_vp, eye = Camera().uniforms_f32()
    renderer.set_uniforms(inv_vp, eye)
    w, h, b = rec["w"], rec["h"], rec["bounces"]
    renderer.reset_stats()
    render_frame(renderer, w, h, 0, b, aux=True)
    assert sha(renderer.read_sample()) == rec["sample_sha256_per_frame"][0]
    aux = renderer.read_aux()
    assert sha(aux[..., 0]) == rec["primary_sha256"]["code"]
    assert sha(aux[..., 1]) == rec["primary_sha256"]["t_bits"]
    assert sha(aux[..., 2]) == rec["primary_sha256"]["exists"]
    # the counter-instrumented shader variant compiles slightly differently in rare pixels
    assert abs(renderer.stats()["rays"] - rec["rays_frame0"]) <= 1e-5 * rec["rays_frame0"]
    for f in (1, 2):
        render_frame(renderer, w, h, f, b)
        assert sha(renderer.read_sample()) == rec["sample_sha256_per_frame"][f], f"frame {f}"


def test_random_scenes_match_reference_shader(renderer):
    """48 random scenes pinned on llvmpipe: colour frames, first-hit ids and distances."""
    from util import fuzz_buffers, fuzz_cases

    first = np.load(GOLDEN / "fuzz" / "first_frames.npz")
    bad = []
    for path in fuzz_cases():
        rec = json.loads(path.read_text())
        b, u = fuzz_buffers(rec)
        b.upload(renderer)
        renderer.set_uniforms(u[0:16], u[16:19])
        n = path.stem[4:]
        for f in range(*rec["frames"]):
            img = render_frame(renderer, rec["w"], rec["h"], f, rec["bounces"], aux=(f == 0))
            if sha(img) != rec["sample_sha256_per_frame"][f]:
                bad.append((path.name, f, "colour"))
            if f == 0:
                aux = renderer.read_aux()
                code, tbits, exists = O.primary_from_golden(first["primary" + n])
                if not (np.array_equal(code, aux[..., 0]) and np.array_equal(tbits, aux[..., 1])
                        and np.array_equal(exists, aux[..., 2])):
                    bad.append((path.name, f, "first hit"))
    assert not bad, bad
    # the running mean over six frames of the cases whose samples reach far beyond 1
    for name, acc_rec in json.loads((GOLDEN / "fuzz" / "odd_accumulate.json").read_text()).items():
        rec = json.loads((GOLDEN / "fuzz" / f"{name}.json").read_text())
        b, u = fuzz_buffers(rec)
        b.upload(renderer)
        renderer.set_uniforms(u[0:16], u[16:19])
        renderer.update_size(rec["w"], rec["h"])
        renderer.reset_accumulation()
        for f in range(*acc_rec["frames"]):
            img = render_frame(renderer, rec["w"], rec["h"], f, rec["bounces"])
            assert sha(img) == acc_rec["sample_sha256_per_frame"][f], (name, f)
            assert sha(renderer.read_accumulation()) == acc_rec["acc_sha256_per_frame"][f], (name, f)


def test_fresh_random_scenes_match_the_oracle(renderer):
    """Sixty more random scenes from the generator the llvmpipe-pinned cases were drawn with (other
    seeds; sparse ones with all four primitive kinds and cameras anywhere, and dense ones with
    hundreds of small spheres and boxes), built by the host pipeline: image, first hit, end-of-path
    state and every work counter against the oracle, which is itself pinned on those cases."""
    import sys

    sys.path.insert(0, str(GOLDEN.parents[1] / "oracle"))
    import gen_fuzz as F   # scene generator only: nothing of the reference is needed to draw scenes
    from raytracer_amd import scene as S

    rng = np.random.default_rng(777)
    w, h = 96, 64
    bad = []
    for case in range(60):
        if case % 6 == 5:
            scene, cam = F.dense_scene(rng, 9000 + case, int(rng.integers(50, 400)), int(rng.integers(0, 200)), 3)
        else:
            scene, cam = F.random_scene(rng, 9000 + case)
        b = S.build_scene(scene)
        inv_vp, eye = F.camera_uniforms(cam)
        b.upload(renderer)
        renderer.set_uniforms(inv_vp, eye)
        u = np.concatenate([inv_vp, eye, b.scene_min, b.scene_extent]).astype(np.float32)
        sc = O.OracleScene(bvh=b.bvh.tobytes(), tris=b.triangles.tobytes(), spheres=b.spheres.tobytes(),
                           mats=b.materials.tobytes(), uniforms=u, boxes=b.boxes.tobytes(), planes=b.planes.tobytes())
        bounces = int(rng.integers(1, 9))
        frame = int(rng.integers(0, 1000))
        renderer.reset_stats()
        img = render_frame(renderer, w, h, frame, bounces, aux=True, stats=True)
        aux, st = renderer.read_aux(), renderer.stats()
        ref, raux, cnt = sc.render(w, h, frame, bounces, want_primary=True, want_counters=True)
        if not np.array_equal(img.view(np.uint32), ref.view(np.uint32)):
            bad.append((case, "image"))
        if not np.array_equal(aux, raux):
            bad.append((case, "state"))
        if any(st[k] != cnt[k] for k in ("rays", "nodes", "tri_tests", "sphere_tests", "box_tests", "max_stack")):
            bad.append((case, "counters"))
    assert not bad, bad


def test_the_reference_loaders_own_update_ssbo_arguments(renderer):
    """The arrays exactly as the reference's load_scene hands them to Renderer.update_ssbo
    (scene.py:62,97,121-122,191-193): structured dtypes, materials of shape (n,), spheres and
    triangles of shape (n, 1), and numpy arrays for scene_min_coords / scene_extent -- fed
    unchanged, in the loader's order, and rendered against the reference shader's images."""
    args = np.load(GOLDEN / "debug" / "update_ssbo_args.npz")
    assert args["spheres"].ndim == 2 and args["triangles"].ndim == 2 and args["materials"].dtype.names
    renderer.update_ssbo(renderer.materials, args["materials"], 9)
    renderer.update_ssbo(renderer.spheres, args["spheres"], 5)
    renderer.update_ssbo(renderer.triangles, args["triangles"], 12)
    renderer.update_ssbo(renderer.bvh, args["bvh"], 11)
    for binding in (6, 7):
        renderer.update_ssbo(0, np.zeros(0, np.uint8), binding)   # an earlier test's boxes and planes
    renderer.scene_min_coords = args["scene_min_coords"]
    renderer.scene_extent = args["scene_extent"]
    u = np.fromfile(GOLDEN / "debug" / "uniforms.bin", dtype=np.float32)
    renderer.set_uniforms(u[0:16], u[16:19])
    rec = json.loads((GOLDEN / "debug" / "c1_640x360_b2.json").read_text())
    for i, f in enumerate(range(*rec["frames"])):
        img = render_frame(renderer, rec["w"], rec["h"], f, rec["bounces"])
        assert sha(img) == rec["kinds"]["color"]["sha256_per_frame"][i], f


def test_edge_sizes_and_truncated_dispatch(renderer):
    """Sizes that are not multiples of 8x4 leave the edge pixels unwritten, as the reference's
    glDispatchCompute(int(W/8), int(H/4), 1) does."""
    load_golden_scene(renderer, "debug")
    sc = O.OracleScene.from_dir(GOLDEN / "debug")
    for (w, h) in ((37, 23), (8, 4), (7, 3), (129, 66)):
        img = render_frame(renderer, w, h, 0, 3)
        ref = sc.render(w, h, 0, 3)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (w, h)


def test_zero_bounces_and_empty_scene(renderer):
    from raytracer_amd.renderer import Renderer

    r = Renderer(device=0, singleton=False)
    try:
        u = np.fromfile(GOLDEN / "debug" / "uniforms.bin", dtype=np.float32)
        r.set_uniforms(u[0:16], u[16:19])
        r.scene_min_coords, r.scene_extent = u[19:22], u[22:25]
        img = render_frame(r, 64, 32, 0, 4)  # nothing uploaded: every read is a zero record
        assert np.all(img[..., :3] == 0) and np.all(img[..., 3] == 1)
        load_golden_scene(r, "debug")
        img = render_frame(r, 64, 32, 0, 0)
        assert np.all(img[..., :3] == 0) and np.all(img[..., 3] == 1)
    finally:
        r.close()


def _linear(img):
    c = img[..., :3].astype(np.float64)
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def test_sum_mode_tile_and_frame_shards_add_up(renderer):
    """rt_render_sum: shards by tile and by frame partition the work exactly; the sum image equals the
    sum of the oracle's linear radiance, and the finalised mean meets the radiance tolerance
    (max-abs <= 1e-3 per channel, PSNR >= 50 dB) against the oracle's mean."""
    import ctypes as C

    from raytracer_amd import _lib, dist as D

    w, h, b, frames = 320, 180, 4, 8
    load_golden_scene(renderer, "gally")
    renderer.update_size(w, h)
    renderer.bounces = b
    L, ctx = renderer._L, renderer._ctx

    def read_sum():
        renderer.sync()
        p, n = renderer.sum_buffer()
        host = np.empty((h, w, 4), np.float32)
        import ctypes

        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        assert hip.hipMemcpy(host.ctypes.data, C.c_void_p(p), n, 2) == 0
        return host

    renderer.clear_sum()
    renderer.render_sum(0, frames)
    full = read_sum()
    # tiles: 3 shards into one buffer
    renderer.clear_sum()
    for s in range(3):
        renderer.render_sum(0, frames, shard=s, nshards=3)
    tiled = read_sum()
    assert np.array_equal(full.view(np.uint32), tiled.view(np.uint32))
    # frames: 2 phases into one buffer (per-pixel addition order differs: fp32 reassociation only)
    renderer.clear_sum()
    for ph in range(2):
        renderer.render_sum(0, frames, frame_stride=2, frame_phase=ph)
    strided = read_sum()
    np.testing.assert_allclose(strided, full, rtol=2e-6, atol=1e-6)
    assert np.all(full[..., 3] == frames)
    # single shard covers exactly its mask
    renderer.clear_sum()
    renderer.render_sum(0, 1, shard=1, nshards=3)
    one = read_sum()
    assert np.array_equal(one[..., 3] == 1, D.owner_mask(w, h, 1, 3))
    # against the oracle
    sc = O.OracleScene.from_dir(GOLDEN / "gally")
    ref = np.zeros((h, w, 3), np.float64)
    for f in range(frames):
        ref += _linear(sc.render(w, h, f, b))
    # the device sums the kernel's own linear radiance; the oracle image went through sRGB and back
    np.testing.assert_allclose(full[..., :3], ref, rtol=1e-4, atol=1e-4)
    renderer.clear_sum()
    renderer.render_sum(0, frames)
    renderer.finalize_sum(frames)
    mean_gpu = renderer.read_accumulation()
    mean_ref = D.finalize_mean_srgb(np.concatenate([ref, np.zeros((h, w, 1))], -1), frames)
    err = np.abs(mean_gpu[..., :3].astype(np.float64) - mean_ref[..., :3])
    assert err.max() <= 1e-3
    mse = float(np.mean(err ** 2))
    peak = float(max(mean_ref[..., :3].max(), 1.0))
    assert mse == 0 or 10 * np.log10(peak * peak / mse) >= 50.0


def test_sum_mode_two_frames_in_flight_is_deterministic(renderer):
    """Sum mode keeps two frames in flight: launches alternate between the sum image and a second
    image that is folded in at the next synchronisation.  The result must be exactly
    (L0 + L2 + L4) + (L1 + L3 + L5), Lk = the linear radiance of launch k, whatever the timing."""
    import ctypes as C

    w, h, b = 320, 180, 4
    load_golden_scene(renderer, "matlab")
    renderer.update_size(w, h)
    renderer.bounces = b
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def read_sum():
        renderer.sync()
        p, n = renderer.sum_buffer()
        host = np.empty((h, w, 4), np.float32)
        assert hip.hipMemcpy(host.ctypes.data, C.c_void_p(p), n, 2) == 0
        return host

    per_frame = []
    for f in range(6):
        renderer.clear_sum()
        renderer.render_sum(f, 1)
        per_frame.append(read_sum())
    import os

    # two frames in flight: even and odd frames are summed apart (RT_OVERLAP=0 restores the single running sum)
    if renderer.pipeline_name in ("mega", "regroup") and os.environ.get("RT_OVERLAP", "1") != "0":
        even = (per_frame[0] + per_frame[2]) + per_frame[4]
        odd = (per_frame[1] + per_frame[3]) + per_frame[5]
        expect = even + odd
    else:  # the wavefront pipeline has one ray queue: frames run one at a time, summed in order
        expect = np.zeros_like(per_frame[0])
        for k in range(6):
            expect = expect + per_frame[k]
    for chunks in ((6,), (1, 1, 1, 1, 1, 1), (3, 3)):  # no synchronisation between the calls of a chunking
        renderer.clear_sum()
        f = 0
        for n in chunks:
            renderer.render_sum(f, n)
            f += n
        got = read_sum()
        assert np.array_equal(got.view(np.uint32), expect.view(np.uint32)), chunks


def test_interactive_hooks_reset_the_running_mean(renderer):
    """The app loop's hooks (controller.py:124-128,218-219, window.py:199, renderer.py:170-173): a
    camera change and a bounce-count change followed by reset_accumulation restart the running mean;
    a resize recreates the images but leaves `rendered_frames` alone (renderer.py:347-382), so the
    next frame is blended into black with weight 1/(rendered_frames+1) -- pinned against the
    reference's own shaders in test_accumulation_matches_reference_fragment_shader (resize_*); a
    host that assigns `rendered_frames` itself is obeyed; the frame counter that feeds `time`
    keeps counting."""
    load_golden_scene(renderer, "debug")
    w, h = 160, 88
    renderer.update_size(w, h)
    renderer.rendered_frames = 0   # the fixture's renderer has frames from earlier tests
    renderer.bounces = 2
    renderer.set_frame_index(0)
    renderer.camera.changed = False
    for _ in range(3):
        renderer.render()
    assert renderer.rendered_frames == 3
    mean3 = renderer.read_accumulation()
    assert not np.array_equal(mean3, renderer.read_sample())
    # camera moved: the next render() resets first (the old sample becomes the mean, count 0), so
    # the new mean is the accumulate pass of that one frame with weight 1
    sc = O.OracleScene.from_dir(GOLDEN / "debug")
    renderer.camera.changed = True
    renderer.render()
    assert renderer.rendered_frames == 1
    s3 = sc.render(w, h, 3, 2)  # `time` went on to 3
    assert np.array_equal(renderer.read_sample(), s3)
    assert np.array_equal(renderer.read_accumulation(), O.accumulate(s3, sc.render(w, h, 2, 2), 0))
    # bounces changed by the controller, which then resets
    renderer.render()
    renderer.bounces = 3
    renderer.reset_accumulation()
    assert renderer.rendered_frames == 0
    assert np.array_equal(renderer.read_accumulation(), sc.render(w, h, 4, 2))
    renderer.render()
    s5 = sc.render(w, h, 5, 3)
    assert np.array_equal(renderer.read_sample(), s5)
    assert np.array_equal(renderer.read_accumulation(), O.accumulate(s5, sc.render(w, h, 4, 2), 0))
    # resize: fresh (zero) images, the count of accumulated frames stays
    renderer.render()
    assert renderer.rendered_frames == 2
    renderer.update_size(w + 16, h + 8)
    assert renderer.rendered_frames == 2 and (renderer.width, renderer.height) == (w + 16, h + 8)
    renderer.render()
    s7 = sc.render(w + 16, h + 8, 7, 3)
    assert np.array_equal(renderer.read_sample(), s7)
    assert np.array_equal(renderer.read_accumulation(), O.accumulate(s7, np.zeros_like(s7), 2))
    assert renderer.rendered_frames == 3
    # the attribute is the host's to assign (renderer.py:106,255-257): a clean start after a resize
    renderer.update_size(w, h)
    renderer.rendered_frames = 0
    renderer.render()
    s8 = sc.render(w, h, 8, 3)
    assert np.array_equal(renderer.read_accumulation(), O.accumulate(s8, np.zeros_like(s8), 0))
    with pytest.raises(TypeError):
        renderer.bounces = -1


def test_finalize_then_more_frames_without_clear(renderer):
    """Progressive preview: render_sum, finalize_sum, render_sum, ... with no clear in between.  The
    finalize pass reads the sum image on the main stream while later frames add into it from the
    trace streams; they must be ordered.  The sum after every step equals the per-frame expectation
    bit for bit, and every preview equals the finalised image of that expectation."""
    import ctypes as C

    w, h, b = 320, 180, 4
    load_golden_scene(renderer, "gally")
    renderer.update_size(w, h)
    renderer.bounces = b
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def read_sum():
        renderer.sync()
        p, n = renderer.sum_buffer()
        host = np.empty((h, w, 4), np.float32)
        assert hip.hipMemcpy(host.ctypes.data, C.c_void_p(p), n, 2) == 0
        return host

    per_frame = []
    for f in range(5):
        renderer.clear_sum()
        renderer.render_sum(f, 1)
        per_frame.append(read_sum())
    # reference previews: finalize of a sum built in the same order, one clear + one finalize each
    def preview_of(total):
        renderer.clear_sum()
        p, n = renderer.sum_buffer()
        renderer.sync()
        assert hip.hipMemcpy(C.c_void_p(p), total.ctypes.data, n, 1) == 0
        renderer.finalize_sum(int(total[0, 0, 3]))
        return renderer.read_accumulation()

    for _ in range(3):  # repeated: an unordered read/write would show up only sometimes
        renderer.clear_sum()
        # frame k goes to the sum image itself when k is even, to the second image (folded in by
        # the next finalize) when k is odd; a finalize after every frame makes the order running
        total = np.zeros_like(per_frame[0])
        previews = []
        for f in range(5):
            renderer.render_sum(f, 1)
            total = total + per_frame[f]
            renderer.finalize_sum(f + 1)       # no sync, no clear
            if f in (1, 4):
                previews.append((f, renderer.read_accumulation(), total.copy()))
        got = read_sum()
        assert np.array_equal(got.view(np.uint32), total.view(np.uint32))
        for f, img, tot in previews:
            assert np.array_equal(img.view(np.uint32), preview_of(tot).view(np.uint32)), f"preview after frame {f}"


def test_mapped_images_equal_the_copied_ones(renderer):
    """rt_map_image: the page-locked view holds what rt_read_* copies out, stays unchanged while
    further frames render, and follows a resize."""
    load_golden_scene(renderer, "debug")
    render_frame(renderer, 320, 180, 0, 3)
    render_frame(renderer, 320, 180, 1, 3)
    acc = renderer.map_image()
    assert acc.shape == (180, 320, 4) and not acc.flags.writeable
    assert np.array_equal(acc.view(np.uint32), renderer.read_accumulation().view(np.uint32))
    kept = acc.copy()
    render_frame(renderer, 320, 180, 2, 3)  # rendering on does not touch the mapped image
    assert np.array_equal(acc, kept)
    smp = renderer.map_image(accumulation=False)  # the same buffer, now holding the last sample
    assert np.array_equal(smp.view(np.uint32), renderer.read_sample().view(np.uint32))
    render_frame(renderer, 160, 88, 0, 3)
    small = renderer.map_image()
    assert small.shape == (88, 160, 4)
    assert np.array_equal(small.view(np.uint32), renderer.read_accumulation().view(np.uint32))
    with pytest.raises(RuntimeError):
        from raytracer_amd import _lib
        import ctypes as C
        p = C.POINTER(C.c_float)()
        _lib.check(_lib.load().rt_map_image(renderer._ctx, 2, C.byref(p)))


def test_tile_ordering_renders_every_pixel_exactly_once(renderer):
    """Frames that run alone have their macro tiles launched in the order of their cost in the
    previous frame (RT_LPT, on by default).  At sizes with partial tiles and partial macro tiles,
    over several frames: every covered pixel gets exactly one sample per frame, and the sum image is
    bit for bit the one a context without the ordering produces."""
    import os

    from raytracer_amd.renderer import Renderer

    os.environ["RT_LPT"] = "0"
    os.environ["RT_PIPELINE"] = renderer.pipeline_name
    try:
        plain = Renderer(device=0, singleton=False)
    finally:
        os.environ.pop("RT_LPT", None)
        os.environ.pop("RT_PIPELINE", None)
    try:
        sums = []
        for r in (renderer, plain):
            load_golden_scene(r, "gally")
            r.bounces = 3
            per_size = []
            for (w, h) in ((129, 66), (1000, 600), (37, 23)):
                r.update_size(w, h)
                r.clear_sum()
                for f in range(5):
                    r.render_sum(f, 1)
                    r.sync()  # the next frame finds the device idle: it runs alone
                per_size.append(r.read_sum())
            sums.append(per_size)
        for (w, h), a, b in zip(((129, 66), (1000, 600), (37, 23)), *sums):
            ww, hh = (w // 8) * 8, (h // 4) * 4
            assert np.all(a[:hh, :ww, 3] == 5) and np.all(a[hh:, :, 3] == 0) and np.all(a[:, ww:, 3] == 0), (w, h)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (w, h)
    finally:
        plain.close()


def test_recorded_frame_followed_at_once_by_sum_frames(renderer):
    """A frame with the per-pixel record runs on the main stream, sum-mode frames on the trace
    streams, and both take their tile order from the same buffers: queued back to back without a
    wait in between, the recorded frame must still render every pixel once (its image is the
    oracle's) and so must the sum frames."""
    load_golden_scene(renderer, "gally")
    w, h, b = 640, 360, 4
    renderer.update_size(w, h)
    renderer.bounces = b
    sc = O.OracleScene.from_dir(GOLDEN / "gally")
    ref = {f: sc.render(w, h, f, b) for f in (0, 1)}
    for rep in range(6):
        f = rep & 1
        renderer.clear_sum()
        renderer.set_frame_index(f)
        renderer.set_flags(aux=True)
        renderer.camera.changed = False
        renderer.render()               # main stream, no wait
        renderer.render_sum(10 + rep, 3)  # trace streams, no wait
        img = renderer.read_sample()
        assert np.array_equal(img.view(np.uint32), ref[f].view(np.uint32)), rep
        assert np.all(renderer.read_sum()[..., 3] == 3), rep
    renderer.set_flags()


def test_failed_resize_keeps_the_context_usable(renderer):
    """rt_resize refuses an absurd size with a message and leaves the previous images in place."""
    load_golden_scene(renderer, "debug")
    sc = O.OracleScene.from_dir(GOLDEN / "debug")
    img = render_frame(renderer, 64, 32, 0, 2)
    L, ctx = renderer._L, renderer._ctx
    assert L.rt_resize(ctx, 1 << 20, 1 << 20) != 0
    assert b"too large" in L.rt_last_error()
    assert L.rt_resize(ctx, 0, 5) != 0
    assert np.array_equal(renderer.read_sample(), img)          # same size, same contents
    img2 = render_frame(renderer, 64, 32, 1, 2)
    assert np.array_equal(img2.view(np.uint32), sc.render(64, 32, 1, 2).view(np.uint32))


def test_error_reporting(renderer):
    with pytest.raises(ValueError):
        renderer.update_ssbo(0, np.zeros(64, np.uint8), 13)
    with pytest.raises(RuntimeError):
        renderer.update_ssbo(0, np.zeros(65, np.uint8), 11)  # not a multiple of the node stride
    with pytest.raises(TypeError):
        renderer.bounces = -1
    with pytest.raises(RuntimeError):
        renderer.render_sum(0, 1, shard=3, nshards=3)
    good = renderer._bounces
    for bad in (-1.0, float("nan")):  # underneath the Python property: the C ABI refuses them itself
        renderer._bounces = bad
        with pytest.raises(RuntimeError, match="bounces"):
            renderer.render()
        with pytest.raises(RuntimeError, match="bounces"):
            renderer.render_sum(0, 1)
    renderer._bounces = good


def test_traversal_variant_is_timed_once_a_scene_is_in_use_and_changes_no_pixel():
    """Left to itself the library picks a traversal variant by the size of the scene, and once the
    scene has seen 64 plain frames it times both variants and keeps the faster
    (rt_traversal_variant); the launches it does that with leave no trace in the images or the
    counters, and whichever variant runs, the images are the reference shader's."""
    from raytracer_amd.renderer import Renderer

    scene, tag = _cases()[0]
    rec = json.loads((GOLDEN / scene / f"{tag}.json").read_text())
    hashes = rec["kinds"]["color"]["sha256_per_frame"] if "kinds" in rec else rec["sha256_per_frame"]
    w, h, b = rec["w"], rec["h"], rec["bounces"]
    f0, f1 = rec["frames"]
    r = Renderer(device=0, singleton=False)
    try:
        load_golden_scene(r, scene)
        assert r.traversal_variant()[1] is False
        render_frame(r, w, h, f0, b, stats=True)  # an instrumented frame decides nothing
        assert r.traversal_variant()[1] is False
        rays_of_one_frame = r.stats()["rays"]
        r.reset_stats()
        n = 0
        while n < 66:  # 64 frames by the size rule, the timing before the 65th
            for f in range(f0, f1):
                assert sha(render_frame(r, w, h, f, b)) == hashes[f - f0], f"frame {f}"
                n += 1
                assert r.traversal_variant()[1] is (n > 64), n
        name, decided = r.traversal_variant()
        assert decided and name in ("unified", "leaf steps")
        assert r.stats()["rays"] <= rays_of_one_frame * n * 1.01  # the timing launches count elsewhere
        load_golden_scene(r, scene)  # the same scene uploaded again: undecided only until the next frame,
        assert r.traversal_variant()[1] is False  # which keeps the choice without timing anything
        assert sha(render_frame(r, w, h, f0, b)) == hashes[0]
        assert r.traversal_variant() == (name, True)
        r.update_size(w * 2, h * 2)  # a much larger image starts over
        assert r.traversal_variant()[1] is False
    finally:
        r.close()


def test_traversal_variant_can_be_fixed_per_context():
    """rt_set_traversal_variant: a host that must not see the one-off timing launches fixes the
    variant itself; no frame count then triggers them, both variants render the reference shader's
    images, and -1 hands the choice back to the library."""
    from raytracer_amd.renderer import Renderer

    scene, tag = _cases()[0]
    rec = json.loads((GOLDEN / scene / f"{tag}.json").read_text())
    hashes = rec["kinds"]["color"]["sha256_per_frame"] if "kinds" in rec else rec["sha256_per_frame"]
    w, h, b = rec["w"], rec["h"], rec["bounces"]
    f0 = rec["frames"][0]
    r = Renderer(device=0, singleton=False)
    try:
        load_golden_scene(r, scene)
        for name in ("unified", "leaf steps"):
            r.set_traversal_variant(name)
            assert r.traversal_variant() == (name, True)
            for _ in range(70):   # beyond the 64 frames after which the library would time the variants
                img = render_frame(r, w, h, f0, b)
            assert sha(img) == hashes[0]
            assert r.traversal_variant() == (name, True)
        r.set_traversal_variant(None)
        assert r.traversal_variant()[1] is False
        with pytest.raises(RuntimeError, match="variant"):
            from raytracer_amd import _lib
            _lib.check(_lib.load().rt_set_traversal_variant(r._ctx, 2))
    finally:
        r.close()


@pytest.mark.parametrize("knobs", [{"RT_NODE_LAYOUT": "0"}, {"RT_NODE_LAYOUT": "1"}, {"RT_NODE_LAYOUT": "2"},
                                   {"RT_PIPELINE": "wave", "RT_WAVE_BIN": "1"}])
def test_experiment_knobs_change_no_pixel(knobs):
    """RT_NODE_LAYOUT (which nodes form the LDS copy -- the breadth-first top instead of the default,
    the nodes with the largest boxes -- and depth-first subtrees / three-level treelets below it instead
    of breadth-first order) renumbers the device node records only, and RT_WAVE_BIN regroups the
    wavefront pipeline's rays by direction octant between bounces: images and first hits stay the
    reference shader's."""
    import os

    from raytracer_amd.renderer import Renderer

    os.environ.update(knobs)
    try:
        r = Renderer(device=0, singleton=False)
    finally:
        for k in knobs:
            os.environ.pop(k, None)
    try:
        for scene, tag in (("gally", "small_160x88_b4"), ("matlab", "small_160x88_b6")):
            rec = json.loads((GOLDEN / scene / f"{tag}.json").read_text())
            load_golden_scene(r, scene)
            f0 = rec["frames"][0]
            img = render_frame(r, rec["w"], rec["h"], f0, rec["bounces"], aux=True)
            assert sha(img) == rec["kinds"]["color"]["sha256_per_frame"][0]
            code, tbits, exists = O.primary_from_golden(np.load(GOLDEN / scene / f"{tag}.primary.npy")[0])
            aux = r.read_aux()
            assert np.array_equal(code, aux[..., 0]) and np.array_equal(tbits, aux[..., 1])
    finally:
        r.close()


def test_the_short_reciprocal_equals_the_division_for_every_float(renderer):
    """The triangle test's 1 / det is computed as the hardware estimate plus one refinement step where
    that is the correctly rounded quotient (1e-6 <= |det| <= 1e30) and by the general division
    elsewhere: the device compares the two over all 2^32 bit patterns."""
    assert renderer.selftest() == 0
