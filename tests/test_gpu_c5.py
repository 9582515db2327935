"""The 4K tile-and-sample-sharded configuration (3840x2160, sponza-scale scene, 4 bounces) on one
GPU: the shards of a multi-GPU run, rendered one after another into one sum image, must add up to
the unsharded image; whole frames are checked against the reference shader's own 4K output and a
crop against the CPU oracle.  Also the RCCL entry points with a
one-rank communicator (all a one-GPU machine can form)."""

import ctypes as C

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

W, H, BOUNCES, SPP = 3840, 2160, 4, 4
CROP = (1760, 900, 320, 180)  # x0, y0, w, h: centre of the image, straddles 64x64 tile borders


@pytest.fixture(scope="module")
def setup():
    from raytracer_amd import scene as S, synth
    from raytracer_amd.camera import Camera
    from raytracer_amd.renderer import Renderer

    scene, meshes = synth.sponza_synth(1.0)
    bufs = S.build_scene(scene, extra_meshes=meshes, sequential_mesh_indices=True)
    r = Renderer(device=0, singleton=False)
    r.update_size(W, H)
    bufs.upload(r)
    inv_vp, eye = Camera().uniforms_f32()
    r.set_uniforms(inv_vp, eye)
    r.bounces = BOUNCES
    u = np.concatenate([inv_vp, eye, bufs.scene_min, bufs.scene_extent]).astype(np.float32)
    sc = O.OracleScene(bvh=bufs.bvh.tobytes(), tris=bufs.triangles.tobytes(), spheres=bufs.spheres.tobytes(),
                       mats=bufs.materials.tobytes(), uniforms=u)
    yield r, sc
    r.close()


def _linear(img):
    c = img[..., :3].astype(np.float64)
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def test_4k_tile_and_sample_shards_add_up(setup):
    import bench
    from raytracer_amd import dist as D

    r, sc = setup
    r.clear_sum()
    r.render_sum(0, SPP)
    full = r.read_sum()
    assert np.all(full[..., 3] == SPP)                      # 3840x2160 is covered entirely by 8x4 groups
    assert full[..., :3].max() > 0

    # tile shards only (every shard renders all samples, in the same order): bit for bit
    r.clear_sum()
    for s in range(2):
        r.render_sum(0, SPP, shard=s, nshards=2)
    tiled = r.read_sum()
    assert np.array_equal(tiled.view(np.uint32), full.view(np.uint32))

    # tiles x samples, as bench.py --config c5 deals them to 4 and to 8 ranks: the per-pixel order of
    # the additions differs, so fp32 reassociation only
    for world in (4, 8):
        tiles, phases = bench.c5_split(world)
        r.clear_sum()
        for rank in range(world):
            r.render_sum(0, SPP, shard=rank % tiles, nshards=tiles, frame_stride=phases, frame_phase=rank // tiles)
        both = r.read_sum()
        assert np.all(both[..., 3] == SPP), world         # every pixel-sample exactly once
        np.testing.assert_allclose(both[..., :3], full[..., :3], rtol=2e-6, atol=1e-6)

    # one rank's share: alpha counts its samples on its tiles and nothing elsewhere
    r.clear_sum()
    r.render_sum(0, SPP, shard=1, nshards=2, frame_stride=2, frame_phase=1)
    one = r.read_sum()
    mask = D.owner_mask(W, H, 1, 2)
    assert np.array_equal(one[..., 3], mask * (SPP // 2))

    # a crop of the sum against the oracle's linear radiance (its images are sRGB-encoded)
    x0, y0, cw, ch = CROP
    ref = np.zeros((ch, cw, 3), np.float64)
    for f in range(SPP):
        ref += _linear(sc.render_window(W, H, x0, y0, cw, ch, f, BOUNCES))
    np.testing.assert_allclose(full[y0:y0 + ch, x0:x0 + cw, :3], ref, rtol=1e-4, atol=1e-4)

    # finalised mean against the oracle's mean: the radiance tolerance (max-abs 1e-3, PSNR >= 50 dB)
    r.clear_sum()
    r.render_sum(0, SPP)
    r.finalize_sum(SPP)
    mean_gpu = r.read_accumulation()[y0:y0 + ch, x0:x0 + cw, :3].astype(np.float64)
    mean_ref = D.finalize_mean_srgb(np.concatenate([ref, np.zeros((ch, cw, 1))], -1), SPP)[..., :3]
    err = np.abs(mean_gpu - mean_ref)
    assert err.max() <= 1e-3
    mse = float(np.mean(err ** 2))
    assert mse == 0 or 10 * np.log10(max(mean_ref.max(), 1.0) ** 2 / mse) >= 50.0


def test_4k_frame_crop_is_bit_identical_to_the_oracle(setup):
    r, sc = setup
    x0, y0, cw, ch = CROP
    for frame in (0, 255):
        r.set_frame_index(frame)
        r.set_flags()
        r.camera.changed = False
        r.render()
        img = r.read_sample()
        ref = sc.render_window(W, H, x0, y0, cw, ch, frame, BOUNCES)
        assert np.array_equal(img[y0:y0 + ch, x0:x0 + cw].view(np.uint32), ref.view(np.uint32)), frame
    # a corner too (edge tiles)
    ref = sc.render_window(W, H, W - 64, H - 32, 64, 32, 255, BOUNCES)
    assert np.array_equal(img[H - 32:, W - 64:].view(np.uint32), ref.view(np.uint32))


def test_4k_frames_are_the_reference_shaders(setup):
    """Two whole 3840x2160 frames against the reference's compute.glsl on llvmpipe
    (oracle/gen_fixtures.py --only sponza4k; hashes only), in both traversal variants' kernels:
    the frames go through whichever the library picked, and the sum of a one-frame sum image
    (the kernel the multi-GPU ranks run) decodes from the same radiance."""
    import json

    from util import GOLDEN, sha

    rec = json.loads((GOLDEN / "c5_sponza_synth_3840x2160_b4.json").read_text())
    assert (rec["w"], rec["h"], rec["bounces"]) == (W, H, BOUNCES)
    r, _ = setup
    f0, f1 = rec["frames"]
    for i, f in enumerate(range(f0, f1)):
        r.set_frame_index(f)
        r.set_flags()
        r.camera.changed = False
        r.render()
        assert sha(r.read_sample()) == rec["sample_sha256_per_frame"][i], f"frame {f}"


def test_rccl_entry_points_with_a_one_rank_communicator(setup):
    """rt_comm_*: id, communicator, sum-reduce of the image (a no-op sum at one rank: the image must
    come back unchanged), scalar all-reduce, barrier; and the argument checks."""
    from raytracer_amd import _lib

    r, _ = setup
    L = _lib.load()
    assert L.rt_comm_reduce_sum(r._ctx, None, 0) != 0      # no communicator yet
    assert b"rt_comm_init" in L.rt_last_error()
    ident = C.create_string_buffer(_lib.RT_COMM_ID_BYTES)
    _lib.check(L.rt_comm_unique_id(ident))
    assert L.rt_comm_init(r._ctx, 1, 1, ident.raw) != 0    # rank out of range
    _lib.check(L.rt_comm_init(r._ctx, 0, 1, ident.raw))
    assert L.rt_comm_init(r._ctx, 0, 1, ident.raw) != 0    # only once per context
    r.clear_sum()
    r.render_sum(0, 2)
    before = r.read_sum()
    _lib.check(L.rt_comm_reduce_sum(r._ctx, None, 0))
    assert np.array_equal(r.read_sum().view(np.uint32), before.view(np.uint32))
    v = np.array([1.5, -2.0, 7.0], np.float64)
    _lib.check(L.rt_comm_all_reduce_f64(r._ctx, v.ctypes.data_as(C.POINTER(C.c_double)), 3, _lib.RT_COMM_MAX))
    assert v.tolist() == [1.5, -2.0, 7.0]
    _lib.check(L.rt_comm_all_reduce_f64(r._ctx, v.ctypes.data_as(C.POINTER(C.c_double)), 3, _lib.RT_COMM_SUM))
    assert v.tolist() == [1.5, -2.0, 7.0]
    _lib.check(L.rt_comm_barrier(r._ctx))
    assert L.rt_comm_all_reduce_f64(r._ctx, v.ctypes.data_as(C.POINTER(C.c_double)), 65, 0) != 0
