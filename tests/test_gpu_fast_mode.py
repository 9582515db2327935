"""Fast mode on the GPU (opt-in, not the reference's data: the library's own tree).

* bit for bit against the same traversal in the CPU oracle, over the very tree the library built:
  image, first hit, end-of-path state, node and primitive-test counters;
* against the default (parity) mode on the same rays: never a lost or farther first hit, the pixels
  where it finds a nearer one are counted;
* the 256-spp mean image lies within Monte-Carlo noise of the parity mode's."""

import numpy as np
import pytest

import oracle as O
from raytracer_amd import scene as S
from util import GOLDEN, load_golden_scene, render_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def renderer():
    from raytracer_amd.renderer import Renderer

    r = Renderer(device=0, singleton=False)
    yield r
    r.fast_mode = False
    r.close()


def _host_buffers(scene):
    d = GOLDEN / scene
    rd = lambda n: (d / n).read_bytes() if (d / n).exists() else b""
    return rd("tris.bin"), rd("spheres.bin"), rd("boxes.bin")


@pytest.mark.parametrize("scene,w,h,frame,bounces", [
    ("debug", 320, 180, 1, 3), ("debug_box", 320, 180, 0, 5), ("gally", 640, 360, 2, 4), ("gally", 256, 144, 9, 8),
    ("prims", 320, 180, 2, 6), ("matlab", 640, 360, 1, 6),
])
def test_fast_mode_matches_the_oracles_fast_traversal(renderer, scene, w, h, frame, bounces):
    load_golden_scene(renderer, scene)
    renderer.fast_mode = True
    try:
        renderer.reset_stats()
        img = render_frame(renderer, w, h, frame, bounces, aux=True, stats=True)
        aux = renderer.read_aux()
        stats = renderer.stats()
        nodes = renderer.fast_bvh()
    finally:
        renderer.fast_mode = False
    # the library's tree is the one the host builder produces
    assert np.array_equal(nodes, S.build_fast_bvh(*_host_buffers(scene)))
    sc = O.OracleScene.from_dir(GOLDEN / scene)
    try:
        O.set_fast_bvh(nodes)
        ref, raux, cnt = sc.render(w, h, frame, bounces, want_primary=True, want_counters=True)
    finally:
        O.set_fast_bvh(None)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    for k in range(16):
        assert np.array_equal(aux[..., k], raux[..., k]), f"aux word {k}"
    for k in ("rays", "nodes", "tri_tests", "sphere_tests", "box_tests"):
        assert stats[k] == cnt[k], k
    assert stats["stack_overflows"] == 0
    # switching back gives the reference's result again
    back = render_frame(renderer, w, h, frame, bounces)
    assert np.array_equal(back.view(np.uint32), sc.render(w, h, frame, bounces).view(np.uint32))


def test_fast_mode_on_the_bench_scene_vs_parity_mode(renderer):
    """sponza_synth at 1920x1080, 4 bounces: first hits of both modes on the same primary rays, a crop
    against the oracle's fast traversal, and the work per ray."""
    from raytracer_amd import synth
    from raytracer_amd.camera import Camera

    scene, meshes = synth.sponza_synth(1.0)
    bufs = S.build_scene(scene, extra_meshes=meshes, sequential_mesh_indices=True)
    bufs.upload(renderer)
    inv_vp, eye = Camera().uniforms_f32()
    renderer.set_uniforms(inv_vp, eye)
    w, h, b = 1920, 1080, 4
    renderer.reset_stats()
    render_frame(renderer, w, h, 0, b, aux=True, stats=True)
    paux, pst = renderer.read_aux(), renderer.stats()
    renderer.fast_mode = True
    try:
        renderer.reset_stats()
        fimg = render_frame(renderer, w, h, 0, b, aux=True, stats=True)
        faux, fst = renderer.read_aux(), renderer.stats()
        nodes = renderer.fast_bvh()
    finally:
        renderer.fast_mode = False
    tr, tf = paux[..., 1].view(np.float32), faux[..., 1].view(np.float32)
    hit_p, hit_f = paux[..., 2] == 1, faux[..., 2] == 1
    assert not np.any(hit_p & ~hit_f)
    both = hit_p & hit_f
    assert not np.any(tf[both] > tr[both] + 2e-6 * np.maximum(1.0, tr[both]))
    nearer = int(np.sum(both & (tf < tr - 2e-6 * np.maximum(1.0, tr))) + np.sum(hit_f & ~hit_p))
    same = float(np.mean(paux[..., 0] == faux[..., 0]))
    print(f"bench scene: same first primitive in {same:.6f} of the pixels; {nearer} pixels where the reference's tree "
          f"misses a nearer hit; node visits per ray {pst['nodes'] / pst['rays']:.2f} -> {fst['nodes'] / fst['rays']:.2f}, "
          f"triangle tests per ray {pst['tri_tests'] / pst['rays']:.2f} -> {fst['tri_tests'] / fst['rays']:.2f}")
    assert same > 0.99
    assert fst["nodes"] < pst["nodes"]
    assert fst["stack_overflows"] == 0
    # a crop, bit for bit against the oracle's fast traversal of the same tree
    u = np.concatenate([inv_vp, eye, bufs.scene_min, bufs.scene_extent]).astype(np.float32)
    sc = O.OracleScene(bvh=bufs.bvh.tobytes(), tris=bufs.triangles.tobytes(), spheres=bufs.spheres.tobytes(),
                       mats=bufs.materials.tobytes(), uniforms=u)
    x0, y0, cw, ch = 800, 450, 320, 180
    try:
        O.set_fast_bvh(nodes)
        ref = sc.render_window(w, h, x0, y0, cw, ch, 0, b)
    finally:
        O.set_fast_bvh(None)
    assert np.array_equal(fimg[y0:y0 + ch, x0:x0 + cw].view(np.uint32), ref.view(np.uint32))


def test_fast_mode_mean_image_is_within_monte_carlo_noise_of_parity_mode(renderer):
    """256 samples per pixel in both modes.  The yardstick is the parity mode against itself with
    other samples (frames 256..511): the fast mode's mean must be as close to the parity mean as
    that, up to a small factor."""
    from raytracer_amd import dist as D

    load_golden_scene(renderer, "gally")
    w, h, spp = 320, 180, 256
    renderer.update_size(w, h)
    renderer.bounces = 4

    def mean(first, fast):
        renderer.fast_mode = fast
        try:
            renderer.clear_sum()
            renderer.render_sum(first, spp)
            return renderer.read_sum()[..., :3].astype(np.float64) / spp
        finally:
            renderer.fast_mode = False

    parity_a, parity_b, fast_a = mean(0, False), mean(spp, False), mean(0, True)
    noise = np.sqrt(np.mean((parity_a - parity_b) ** 2))
    diff = np.sqrt(np.mean((fast_a - parity_a) ** 2))
    print(f"rms(parity A - parity B) = {noise:.5f}, rms(fast - parity) on the same samples = {diff:.5f}")
    assert noise > 0
    assert diff <= 1.5 * noise
    # and in the metric's own terms, on the sRGB-encoded means
    enc = lambda m: D.finalize_mean_srgb(np.concatenate([m, np.zeros((h, w, 1))], -1), 1)[..., :3].astype(np.float64)
    err = np.abs(enc(fast_a) - enc(parity_a))
    mse = float(np.mean(err ** 2))
    psnr = 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)
    print(f"fast vs parity, sRGB mean image: max abs {err.max():.4f}, PSNR {psnr:.1f} dB")
    assert psnr >= 40.0


def test_fast_mode_on_fresh_random_scenes(renderer):
    """Thirty random scenes (all four primitive kinds, cameras anywhere; every sixth dense with
    hundreds of spheres and boxes): the fast-mode kernel against the oracle's traversal of the tree
    the library built -- image, end-of-path state, work counters -- and never a lost first hit
    against the default mode."""
    import sys

    sys.path.insert(0, str(GOLDEN.parents[1] / "oracle"))
    import gen_fuzz as F

    rng = np.random.default_rng(4242)
    w, h = 96, 64
    bad = []
    for case in range(30):
        if case % 6 == 5:
            scene, cam = F.dense_scene(rng, 9500 + case, int(rng.integers(50, 400)), int(rng.integers(0, 200)), 3)
        else:
            scene, cam = F.random_scene(rng, 9500 + case)
        b = S.build_scene(scene)
        inv_vp, eye = F.camera_uniforms(cam)
        b.upload(renderer)
        renderer.set_uniforms(inv_vp, eye)
        u = np.concatenate([inv_vp, eye, b.scene_min, b.scene_extent]).astype(np.float32)
        sc = O.OracleScene(bvh=b.bvh.tobytes(), tris=b.triangles.tobytes(), spheres=b.spheres.tobytes(),
                           mats=b.materials.tobytes(), uniforms=u, boxes=b.boxes.tobytes(), planes=b.planes.tobytes())
        bounces, frame = int(rng.integers(1, 7)), int(rng.integers(0, 1000))
        render_frame(renderer, w, h, frame, bounces, aux=True)
        paux = renderer.read_aux()
        renderer.fast_mode = True
        try:
            renderer.reset_stats()
            img = render_frame(renderer, w, h, frame, bounces, aux=True, stats=True)
            aux, st, nodes = renderer.read_aux(), renderer.stats(), renderer.fast_bvh()
        finally:
            renderer.fast_mode = False
        try:
            O.set_fast_bvh(nodes)
            ref, raux, cnt = sc.render(w, h, frame, bounces, want_primary=True, want_counters=True)
        finally:
            O.set_fast_bvh(None)
        if not (np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and np.array_equal(aux, raux)):
            bad.append((case, "oracle"))
        if any(st[k] != cnt[k] for k in ("rays", "nodes", "tri_tests", "sphere_tests", "box_tests")):
            bad.append((case, "counters"))
        # first hits against the default mode: none lost, none farther (planes are outside both trees)
        tr, tf = paux[..., 1].view(np.float32), aux[..., 1].view(np.float32)
        hp, hf = paux[..., 2] == 1, aux[..., 2] == 1
        if np.any(hp & ~hf) or np.any(tf[hp & hf] > tr[hp & hf] + 2e-6 * np.maximum(1.0, tr[hp & hf])):
            bad.append((case, "lost or farther first hit"))
    assert not bad, bad
