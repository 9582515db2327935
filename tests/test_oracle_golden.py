"""The CPU oracle (oracle/rt_oracle.c) against images the reference shader produced on llvmpipe.

Every comparison here is bit-for-bit.  The golden data under tests/golden/ was written by
oracle/gen_fixtures.py, which executes the reference's own shaders/compute.glsl.
"""

import hashlib
import json

import numpy as np
import pytest

import oracle as O


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _cases(golden):
    out = []
    for sdir in sorted(p for p in golden.iterdir() if p.is_dir() and p.name != "fuzz"):
        for j in sorted(sdir.glob("*.json")):
            if j.name in ("meta.json", "scene.json") or j.name.endswith((".state.json", ".acc.json")):
                continue
            out.append((sdir.name, j.stem))
    return out


from conftest import GOLDEN  # noqa: E402

CASES = _cases(GOLDEN)


@pytest.fixture(scope="module")
def scenes():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = O.OracleScene.from_dir(GOLDEN / name)
        return cache[name]

    return get


@pytest.mark.parametrize("scene,tag", CASES)
def test_colour_frames_bit_exact(scenes, scene, tag):
    rec = json.loads((GOLDEN / scene / f"{tag}.json").read_text())
    w, h, b = rec["w"], rec["h"], rec["bounces"]
    f0, f1 = rec["frames"]
    hashes = rec["kinds"]["color"]["sha256_per_frame"] if "kinds" in rec else rec["sha256_per_frame"]
    if w * h * (f1 - f0) > 640 * 360 * 8:
        f1 = f0 + 1 if w > 1000 else f0 + 16  # keep the CPU suite short; the GPU suite checks them all
    sc = scenes(scene)
    for i, f in enumerate(range(f0, f1)):
        img = sc.render(w, h, f, b)
        assert sha(img) == hashes[i], f"{scene}/{tag} frame {f}"


@pytest.mark.parametrize("scene,tag", [c for c in CASES if (GOLDEN / c[0] / f"{c[1]}.primary.npy").exists()])
def test_primary_hits_bit_exact(scenes, scene, tag):
    rec = json.loads((GOLDEN / scene / f"{tag}.json").read_text())
    g = np.load(GOLDEN / scene / f"{tag}.primary.npy")
    w, h, b = rec["w"], rec["h"], rec["bounces"]
    f0, f1 = rec["frames"]
    sc = scenes(scene)
    for i, f in enumerate(range(f0, f1)):
        _, aux = sc.render(w, h, f, b, want_primary=True)
        code, tbits, exists = O.primary_from_golden(g[i])
        assert np.array_equal(code, aux[..., 0])
        assert np.array_equal(tbits, aux[..., 1])
        assert np.array_equal(exists, aux[..., 2])


@pytest.mark.parametrize("scene,tag", [(p.parent.name, p.name[: -len(".state.npz")])
                                       for p in sorted(GOLDEN.glob("*/*.state.npz"))])
def test_per_bounce_state_bit_exact(scenes, scene, tag):
    """Ray direction, origin and throughput after bounces 0..2, as witnessed through the alpha channel."""
    meta = json.loads((GOLDEN / scene / f"{tag}.state.json").read_text())
    z = np.load(GOLDEN / scene / f"{tag}.state.npz")
    w, h, frame = meta["w"], meta["h"], meta["frame"]
    sc = scenes(scene)
    slot = {"dx": 4, "dy": 5, "dz": 6, "ox": 7, "oy": 8, "oz": 9, "thr_x": 10}
    for bnc in meta["capture_bounces"]:
        # rendering with bounces = bnc+1 leaves the state of bounce `bnc` in the aux record
        _, aux = sc.render(w, h, frame, bnc + 1, want_primary=True)
        # pixels that finished bounce `bnc`: the capture is non-zero in at least one direction component
        reached = (z[f"b{bnc}_dx"] != 0) | (z[f"b{bnc}_dy"] != 0) | (z[f"b{bnc}_dz"] != 0)
        assert reached.sum() > 1000
        for key, idx in slot.items():
            if key == "thr_x" and bnc >= 3:
                continue
            ref = z[f"b{bnc}_{key}"].view(np.uint32)
            assert np.array_equal(ref[reached], aux[..., idx][reached]), f"bounce {bnc} {key}"


def test_counters_close_to_instrumented_shader(scenes):
    """Work counters come from a differently compiled (instrumented) shader whose inexact
    optimisations differ in rare pixels, so they are compared with a small tolerance."""
    rec = json.loads((GOLDEN / "gally" / "mid_640x360_b4.json").read_text())
    c = rec["kinds"]["counters"]
    tot = {"rays": 0, "nodes": 0, "tri_tests": 0, "sphere_tests": 0}
    sc = scenes("gally")
    f0, f1 = rec["frames"]
    mx = 0
    for f in range(f0, f1):
        _, cnt = sc.render(rec["w"], rec["h"], f, rec["bounces"], want_counters=True)
        for k in tot:
            tot[k] += cnt[k]
        mx = max(mx, cnt["max_stack"])
    assert tot["rays"] == c["rays"]
    for k in ("nodes", "tri_tests", "sphere_tests"):
        assert abs(tot[k] - c[k]) <= 1e-4 * c[k]
    assert mx == c["max_stack"]


def test_known_answers():
    L = O.lib()
    # exp2/log2/pow edge behaviour of the polynomial forms
    assert L.rto_exp2(-127.0) == 0.0
    assert L.rto_exp2(128.0) == float("inf")
    assert L.rto_log2(0.0) == float("-inf")
    assert L.rto_pow(0.0, 5.0) == 0.0
    assert L.rto_pow(1.0, 2.4) == 1.0
    assert abs(L.rto_sin(1.0) - np.sin(1.0)) < 2e-7
    assert abs(L.rto_cos(5.0) - np.cos(5.0)) < 2e-7


@pytest.mark.parametrize("scene,tag", [(p.parent.name, p.name[: -len(".acc.json")])
                                       for p in sorted(GOLDEN.glob("*/*.acc.json"))])
def test_running_mean_bit_exact(scenes, scene, tag):
    """The accumulate pass (screen_fragment.glsl:94-108) as the reference's own fragment shader
    computed it on llvmpipe, frame after frame, driven as renderer.py:249-288 drives it."""
    rec = json.loads((GOLDEN / scene / f"{tag}.acc.json").read_text())
    kept = np.load(GOLDEN / scene / f"{tag}.acc.npz") if rec["kept"] else {}
    w, h, b = rec["w"], rec["h"], rec["bounces"]
    sc = scenes(scene)
    acc = np.zeros((h, w, 4), np.float32)
    f0, f1 = rec["frames"]
    if w > 1000:
        f1 = f0 + 3  # keep the CPU suite short; the GPU suite checks every frame
    for f in range(f0, f1):
        if f == rec.get("update_size_before_frame"):
            # Renderer.update_size (renderer.py:347-382): new, black textures; the frame count goes on
            acc = np.zeros((h, w, 4), np.float32)
        s = sc.render(w, h, f, b)
        assert sha(s) == rec["sample_sha256_per_frame"][f]
        acc = O.accumulate(s, acc, f)
        assert sha(acc) == rec["acc_sha256_per_frame"][f], f"accumulation after frame {f}"
        if f in rec["kept"]:
            assert np.array_equal(acc.view(np.uint32), kept[f"acc{f}"].view(np.uint32))


def test_sphere_obj_configuration_first_frames():
    """test.toml + sphere.obj at 1920x1080, 4 bounces (32 k triangles): scene built by the host
    pipeline, first two frames and their running mean against the reference shaders' hashes."""
    from raytracer_amd import scene as S
    from raytracer_amd.camera import Camera

    rec = json.loads((GOLDEN / "c2_sphere_obj_1920x1080_b4.json").read_text())
    b = S.build_scene(json.loads((GOLDEN / "gally" / "scene.json").read_text()),
                      extra_meshes=[(np.load(GOLDEN / "meshes" / "sphere.npz")["triangles"], 26)])
    inv_vp, eye = Camera().uniforms_f32()
    sc = O.OracleScene(bvh=b.bvh.tobytes(), tris=b.triangles.tobytes(), spheres=b.spheres.tobytes(),
                       mats=b.materials.tobytes(),
                       uniforms=np.concatenate([inv_vp, eye, b.scene_min, b.scene_extent]))
    w, h, bn = rec["w"], rec["h"], rec["bounces"]
    acc = np.zeros((h, w, 4), np.float32)
    for f in range(2):
        s = sc.render(w, h, f, bn)
        assert sha(s) == rec["sample_sha256_per_frame"][f]
        acc = O.accumulate(s, acc, f)
        assert sha(acc) == rec["acc_sha256_per_frame"][f]


def test_bench_workload_first_frame():
    """sponza_synth (265 k triangles) at 1920x1080, 4 bounces -- the bench workload -- against the
    reference shader's hashes: colour image, first-hit primitive ids and distances of frame 0."""
    from raytracer_amd import scene as S, synth
    from raytracer_amd.camera import Camera

    rec = json.loads((GOLDEN / "c4_sponza_synth_1920x1080_b4.json").read_text())
    scene, meshes = synth.sponza_synth(1.0)
    b = S.build_scene(scene, extra_meshes=meshes, sequential_mesh_indices=True)
    assert sha(b.bvh) == rec["buffers_sha256"]["bvh"] and sha(b.triangles) == rec["buffers_sha256"]["tris"]
    inv_vp, eye = Camera().uniforms_f32()
    sc = O.OracleScene(bvh=b.bvh.tobytes(), tris=b.triangles.tobytes(), spheres=b.spheres.tobytes(),
                       mats=b.materials.tobytes(),
                       uniforms=np.concatenate([inv_vp, eye, b.scene_min, b.scene_extent]))
    img, aux, cnt = sc.render(rec["w"], rec["h"], 0, rec["bounces"], want_primary=True, want_counters=True)
    assert sha(img) == rec["sample_sha256_per_frame"][0]
    assert sha(aux[..., 0]) == rec["primary_sha256"]["code"]
    assert sha(aux[..., 1]) == rec["primary_sha256"]["t_bits"]
    assert sha(aux[..., 2]) == rec["primary_sha256"]["exists"]
    # the counter-instrumented shader variant compiles slightly differently in rare pixels
    assert abs(cnt["rays"] - rec["rays_frame0"]) <= 1e-5 * rec["rays_frame0"]
    # the same scene at configuration C5's size, 3840x2160: one whole frame against the reference shader
    rec4k = json.loads((GOLDEN / "c5_sponza_synth_3840x2160_b4.json").read_text())
    assert rec4k["buffers_sha256"] == rec["buffers_sha256"]
    assert sha(sc.render(rec4k["w"], rec4k["h"], 1, rec4k["bounces"])) == rec4k["sample_sha256_per_frame"][1]


def test_random_scenes_bit_exact():
    """48 random scenes (spheres, quads, boxes, planes, materials over their whole ranges, cameras
    anywhere) rendered by the reference shader on llvmpipe: host pipeline bytes, three colour
    frames, first-hit ids and distances."""
    from util import fuzz_buffers, fuzz_cases

    first = np.load(GOLDEN / "fuzz" / "first_frames.npz")
    cases = fuzz_cases()
    assert len(cases) >= 32
    for path in cases:
        rec = json.loads(path.read_text())
        b, u = fuzz_buffers(rec)
        sc = O.OracleScene(bvh=b.bvh.tobytes(), tris=b.triangles.tobytes(), spheres=b.spheres.tobytes(),
                           mats=b.materials.tobytes(), boxes=b.boxes.tobytes(), planes=b.planes.tobytes(), uniforms=u)
        n = path.stem[4:]
        for f in range(*rec["frames"]):
            img, aux = sc.render(rec["w"], rec["h"], f, rec["bounces"], want_primary=True)
            assert sha(img) == rec["sample_sha256_per_frame"][f], f"{path.name} frame {f}"
            if f == 0:
                assert np.array_equal(img.view(np.uint32), first["color" + n].view(np.uint32))
                code, tbits, exists = O.primary_from_golden(first["primary" + n])
                assert np.array_equal(code, aux[..., 0]) and np.array_equal(tbits, aux[..., 1])
                assert np.array_equal(exists, aux[..., 2])


def test_child_words_outside_the_arrays_follow_the_reference_shader():
    """The gally tree with child words that name nodes and triangles beyond the arrays (the
    reference's empty-slot word on slots with a real box among them), rendered by the reference
    shader on llvmpipe (oracle/gen_out_of_range.py): out-of-range reads deliver zero words, and the
    restatement must agree bit for bit in the images and in the first hits."""
    from util import damaged_gally_bvh

    rec = json.loads((GOLDEN / "fuzz" / "damaged_gally.json").read_text())
    bvh, changed = damaged_gally_bvh()
    assert sha(bvh) == rec["bvh_sha256"] and changed == rec["changed_words"]
    d = GOLDEN / "gally"
    sc = O.OracleScene(bvh=bvh.tobytes(), tris=(d / "tris.bin").read_bytes(), spheres=(d / "spheres.bin").read_bytes(),
                       mats=(d / "mats.bin").read_bytes(), uniforms=np.fromfile(d / "uniforms.bin", dtype=np.float32))
    f0, f1 = rec["frames"]
    for i, f in enumerate(range(f0, f1)):
        img, aux = sc.render(rec["w"], rec["h"], f, rec["bounces"], want_primary=True)
        assert sha(img) == rec["kinds"]["color"]["sha256_per_frame"][i], f"frame {f}"
        if i == 0:
            p = rec["primary_sha256_first_frame"]
            assert sha(aux[..., 0]) == p["code"] and sha(aux[..., 1]) == p["t_bits"] and sha(aux[..., 2]) == p["exists"]


def test_material_indices_outside_the_array_follow_the_reference_shader():
    """Every seventh gally triangle and one sphere carry a material index beyond the material array;
    the reference shader on llvmpipe reads a zero material there."""
    from util import damaged_gally_materials

    rec = json.loads((GOLDEN / "fuzz" / "damaged_materials.json").read_text())
    tris, sph = damaged_gally_materials()
    assert sha(tris) == rec["tris_sha256"] and sha(sph) == rec["spheres_sha256"]
    d = GOLDEN / "gally"
    sc = O.OracleScene(bvh=(d / "bvh.bin").read_bytes(), tris=tris.tobytes(), spheres=sph.tobytes(),
                       mats=(d / "mats.bin").read_bytes(), uniforms=np.fromfile(d / "uniforms.bin", dtype=np.float32))
    f0, f1 = rec["frames"]
    for i, f in enumerate(range(f0, f1)):
        assert sha(sc.render(rec["w"], rec["h"], f, rec["bounces"])) == rec["sha256_per_frame"][i], f"frame {f}"


def test_uniforms_at_the_edges_follow_the_reference_shader(scenes):
    """Fractional bounce counts, one above the shader's own depth cap, and frame numbers where
    float32 no longer holds every integer (the shader's seed arithmetic is in float)."""
    for run in json.loads((GOLDEN / "fuzz" / "edge_uniforms.json").read_text()):
        sc = scenes(run["scene"])
        f0, f1 = run["frames"]
        for i, f in enumerate(range(f0, f1)):
            img = sc.render(run["w"], run["h"], f, run["bounces"])
            assert sha(img) == run["sha256_per_frame"][i], (run["scene"], run["bounces"], f)


def _axis_ray_uniforms(run):
    u = np.fromfile(GOLDEN / "gally" / "uniforms.bin", dtype=np.float32).copy()
    u[0:16] = np.eye(4, dtype=np.float32).reshape(-1)  # every pixel gets the direction (0, 0, 1)
    u[16:19] = np.array([int(x, 16) for x in run["eye_words"]], dtype=np.uint32).view(np.float32)
    return u


def test_axis_parallel_rays_from_box_planes_follow_the_reference_shader():
    """Rays along +z whose origin lies exactly on planes of child boxes: the plane distance is
    0 * inf = NaN there, and the outcome of min / max on a NaN decides whether the child is visited
    (oracle/gen_out_of_range.py, reference shader on llvmpipe)."""
    d = GOLDEN / "gally"
    runs = json.loads((GOLDEN / "fuzz" / "axis_rays.json").read_text())
    assert len(runs) >= 8 and sum(r["first_hits"] > 0 for r in runs) >= 6
    for run in runs:
        sc = O.OracleScene(bvh=(d / "bvh.bin").read_bytes(), tris=(d / "tris.bin").read_bytes(),
                           spheres=(d / "spheres.bin").read_bytes(), mats=(d / "mats.bin").read_bytes(),
                           uniforms=_axis_ray_uniforms(run))
        img, aux = sc.render(run["w"], run["h"], run["frame"], run["bounces"], want_primary=True)
        assert sha(img) == run["sample_sha256"], (run["node"], run["child"])
        assert sha(aux[..., 0]) == run["primary_sha256"]["code"] and sha(aux[..., 1]) == run["primary_sha256"]["t_bits"]


def test_running_mean_of_out_of_range_samples():
    """The accumulate pass on the fuzz cases with materials far outside their ranges, whose samples
    reach well beyond 1 (up to 17.7): six frames of the reference's own screen shaders on llvmpipe."""
    from util import fuzz_buffers

    recs = json.loads((GOLDEN / "fuzz" / "odd_accumulate.json").read_text())
    assert len(recs) >= 4
    for name, acc_rec in recs.items():
        rec = json.loads((GOLDEN / "fuzz" / f"{name}.json").read_text())
        b, u = fuzz_buffers(rec)
        sc = O.OracleScene(bvh=b.bvh.tobytes(), tris=b.triangles.tobytes(), spheres=b.spheres.tobytes(),
                           mats=b.materials.tobytes(), boxes=b.boxes.tobytes(), planes=b.planes.tobytes(), uniforms=u)
        acc = np.zeros((rec["h"], rec["w"], 4), np.float32)
        for f in range(*acc_rec["frames"]):
            smp = sc.render(rec["w"], rec["h"], f, rec["bounces"])
            assert sha(smp) == acc_rec["sample_sha256_per_frame"][f], (name, f)
            acc = O.accumulate(smp, acc, f)
            assert sha(acc) == acc_rec["acc_sha256_per_frame"][f], (name, f)
