#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config c4|c5]

Default workload `c4` (the metric's: "Mrays/s + ms/frame, sponza_full 1080p 4-bounce"): the
Sponza-scale scene at 1920x1080, 1 sample per pixel per frame, 4 bounces, reference default
camera.  `models/sponza_full.obj` is not distributed with the reference, so the deterministic
stand-in `sponza_synth` (about 265 k triangles, raytracer_amd/synth.py) is rendered unless
--obj names a model; the JSON line says which.

c4: one process per GPU; rank r renders the frames f with f % N == r (samples-per-pixel sharding,
scene replicated) into its own fp32 linear-radiance sum image (`rt_render_sum`); the timed region
holds the K steps (one frame per GPU each), the combine (one RCCL reduce, warmed up beforehand;
nothing to do at one GPU) and rank 0's finalise pass that turns the sum into the sRGB mean image.
The step is the same at every N, so values are comparable across N; per-GPU work is fixed: weak
scaling.  At one GPU the reference's own frame (`rt_render`: compute dispatch + accumulate pass,
what `Renderer.render()` does) is timed by the same loop too and reported as `reference_frame`.
After the K steps the same step runs for about two seconds more (`sustained`), with the shader
clock the kernel itself measured.

`--config c5` (strong scaling): one 3840x2160 image of 256 samples per pixel, split over the
ranks by image tiles AND by samples; a step is the whole image, timed from the first launch to
rank 0's finalised image, the reduce included.

With WORLD_SIZE unset and --gpus N > 1 this script starts the N rank processes itself (before it
touches a GPU) and relays rank 0's JSON line; under `python -m torch.distributed.run` it uses the
ranks it is given.  Either way the collectives are RCCL called through the C ABI; no torch.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
PMC_FILE = ROOT / "profiles" / "pmc_trace_kernel.json"  # counters of the dominant kernel (separate rocprofv3 passes)


def build_workload(args):
    from raytracer_amd import scene as S, synth

    if args.obj:
        sc, _ = synth.sponza_synth(0.001)
        meshes = [(S.read_obj(args.obj), 1)]
        name = f"obj:{Path(args.obj).name}"
    else:
        sc, meshes = synth.sponza_synth(args.detail)
        name = f"sponza_synth(detail={args.detail})"
    t = time.time()
    bufs = S.build_scene(sc, extra_meshes=meshes, sequential_mesh_indices=True)
    return name, bufs, time.time() - t


def cpu_baseline(bufs, cam_uniforms, width, height, bounces, threads, budget_s):
    """The CPU restatement (oracle/, OpenMP) timed on a bounded sample of the same workload: whole
    frames at the full size, as many as fit the time budget (at least one)."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as O

    u = np.concatenate([cam_uniforms[0], cam_uniforms[1], bufs.scene_min, bufs.scene_extent]).astype(np.float32)
    sc = O.OracleScene(bvh=bufs.bvh.tobytes(), tris=bufs.triangles.tobytes(), spheres=bufs.spheres.tobytes(),
                       mats=bufs.materials.tobytes(), uniforms=u)
    sc.render(64, 32, 0, bounces, threads=threads)  # warm up the thread pool
    t = time.time()
    frames = 0
    rays = 0
    while time.time() - t < budget_s or frames < 1:
        _, cnt = sc.render(width, height, frames, bounces, want_counters=True, threads=threads)
        rays += cnt["rays"]
        frames += 1
    dt = time.time() - t
    out = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "ms_per_frame": dt / frames * 1e3,
           "sample": f"{frames} full frame(s) of the same scene/camera/bounces at {width}x{height} "
                     f"({rays} rays in {dt:.1f} s, OpenMP C restatement of the reference shader)"}
    # the reference's own shader on Mesa llvmpipe cannot travel to the GPU box; its timing on this
    # very workload was taken where the goldens were generated and is committed with them
    g = ROOT / "tests" / "golden" / "c4_sponza_synth_1920x1080_b4.json"
    if g.exists():
        rec = json.loads(g.read_text())
        if (rec["w"], rec["h"], rec["bounces"]) == (width, height, bounces) and "llvmpipe_ms" in rec:
            ms = float(np.median(rec["llvmpipe_ms"]))
            out["reference_llvmpipe"] = {
                "ms_per_frame": ms, "value": rec["rays_frame0"] / ms / 1e3, "unit": "Mrays/s",
                "threads": rec["threads"], "where": "build container (8-core host), not this machine",
                "what": "the reference's compute.glsl on Mesa llvmpipe, same scene, camera, size and bounces",
                "source": str(g.relative_to(ROOT))}
    return out


def roofline_block(name, args, kernel_ms, rays_per_launch, per_ray):
    """Roofline of the dominant kernel.  `achieved` / `frac` follow the metric's definition
    (algorithmic bytes in the reference's layouts / kernel time / HBM peak): a bytes-per-ray proxy,
    most of which is served by LDS, L1 and L2, so it is not bounded by 1 and is not HBM utilisation.
    What the hardware counters show sits next to it: `hbm_counter_frac` (memory-side bytes / time /
    HBM peak), `td_busy` and `vmem_frac` (the CU's vector-memory return path, which is what binds),
    `l2_hit`; `binding` holds the rest."""
    n_node, n_tri, n_sph, hit_frac, rays_per_sample = per_ray
    # algorithmic bytes per ray in the reference's layouts: 64 B per node visit, 64 B per triangle
    # test, 32 B per sphere test, 12 B of material per hit, 16 B of image per pixel-sample
    bytes_per_ray = 64.0 * n_node + 64.0 * n_tri + 32.0 * n_sph + 12.0 * hit_frac + 16.0 / rays_per_sample
    achieved = rays_per_launch * bytes_per_ray / (kernel_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": None,
           "frac_is": "algorithmic bytes (reference layouts) / kernel time / HBM peak: the metric's bytes-per-ray proxy, "
                      "NOT the share of HBM bandwidth in use (that is hbm_counter_frac) and not bounded by 1; the "
                      "resource that binds is binding_resource",
           "binding_resource": None, "hbm_counter_frac": None, "td_busy": None, "vmem_frac": None, "l2_hit": None,
           "algorithmic_bytes_per_launch": rays_per_launch * bytes_per_ray,
           "kernel": "trace_kernel", "kernel_ms": kernel_ms,
           "kernel_ms_how": "HIP events on the launch's stream, launches issued one at a time after the timed "
                            "region; inside the timed region two launches overlap, so a kernel trace of this "
                            "command shows longer per-kernel durations (RT_OVERLAP=0 gives matching ones)",
           "bytes_per_ray": bytes_per_ray,
           "per_ray": {"node_visits": n_node, "triangle_tests": n_tri, "sphere_tests": n_sph,
                       "hit_fraction": hit_frac, "rays_per_pixel_sample": rays_per_sample}}
    key = f"{name} {args.width}x{args.height} {args.bounces}b"
    if PMC_FILE.exists():
        t = json.loads(PMC_FILE.read_text())
        if t.get("workload_key") == key:
            c = t["counters"]
            # FETCH_SIZE (KB) tallies a 128-byte line request as 64 B on gfx950 (calibrated for this access
            # pattern with tools/microbench/fetch_calib.hip), hence the factor 2
            traffic = (c["FETCH_SIZE"] * 2.0 + c["WRITE_SIZE"]) * 1024.0
            out["traffic"] = traffic
            pmc_ms = t["kernel_ms_in_pmc_run"]
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0           # the counter is summed over the 8 XCDs
            simds, cus = 1024.0, 256.0
            issue = t.get("valu_issue_cycles", 2.0)       # cycles one wave64 VALU instruction holds a SIMD's issue port
            vmem_insts = c.get("SQ_INSTS_VMEM_RD", 0.0) + c.get("SQ_INSTS_VMEM_WR", 0.0)
            out["binding_resource"] = "vector-memory return path of the CU (texture-data unit): td_busy"
            out["hbm_counter_frac"] = traffic / (pmc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["td_busy"] = c["TD_TD_BUSY_sum"] / (cus * cycles) if "TD_TD_BUSY_sum" in c else None
            # a vector-memory wave-instruction of 8 to 16 bytes per lane holds the CU's data-return path
            # for 16 cycles whatever its active lanes (tools/microbench/vmem_mask.hip)
            out["vmem_frac"] = vmem_insts * 16.0 / (cus * cycles) if vmem_insts else None
            out["l2_hit"] = c["TCC_HIT_sum"] / c["TCC_REQ_sum"] if c.get("TCC_REQ_sum") else None
            out["binding"] = {
                "source": str(PMC_FILE.relative_to(ROOT)),
                "hbm_counter_GBs": traffic / (pmc_ms * 1e-3) / 1e9,
                "hbm_counter_frac": out["hbm_counter_frac"],
                "valu_busy": c["SQ_INSTS_VALU"] * issue / (simds * cycles),
                "valu_lane_util": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]),
                # the vector-memory path: texture-address and texture-data units, one of each per CU
                "vmem_issue": c["TA_TA_BUSY_sum"] / (cus * cycles) if "TA_TA_BUSY_sum" in c else None,
                "vmem_data_return": out["td_busy"],
                "vmem_wave_instructions": vmem_insts or None,
                "td_waiting_for_cache": c["TD_TC_STALL_sum"] / (cus * cycles) if "TD_TC_STALL_sum" in c else None,
                "binds": "vector-memory path (texture-data unit)",
                "l1_lookups_per_clk_per_cu": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (cus * cycles),
                "waves_per_simd": c["SQ_WAVE_CYCLES"] * 4.0 / (simds * cycles),
                "shader_clock_GHz": cycles / (pmc_ms * 1e-3) / 1e9,
                "note": t.get("note", "")}
    return out


def pci_to_number(bus_id: str) -> float:
    """'0000:05:00.0' -> one exactly representable number (domain, bus, device, function packed)."""
    try:
        dom, bus, rest = bus_id.strip().split(":")
        dev, fn = rest.split(".")
        return float((int(dom, 16) << 16) | (int(bus, 16) << 8) | (int(dev, 16) << 3) | int(fn, 16))
    except ValueError:
        return -1.0


def number_to_pci(v: float) -> str:
    n = int(v)
    if n < 0:
        return "unknown"
    return f"{n >> 16:04x}:{(n >> 8) & 0xFF:02x}:{(n >> 3) & 0x1F:02x}.{n & 7:x}"


def run_c4(args, r, comm, rank, world, name, bufs, build_s, rehearse):
    """1080p frames.  The timed step is the same at every GPU count: each GPU renders its frames in
    linear-sum mode (`rt_render_sum`), then the images are combined (one RCCL reduce, N > 1) and
    rank 0 finalises the mean image -- all inside the timed region.  At one GPU the reference's own
    frame (`rt_render`: compute dispatch + accumulate pass) is timed too and reported as `reference_frame`."""
    variant, _ = r.traversal_variant()   # timed by the library during the settle step
    npix = args.width * args.height
    # per-ray work of this workload, from one untimed instrumented frame (counter kernel variant)
    r.set_flags(stats=True)
    r.set_frame_index(0)
    r.camera.changed = False
    r.render()
    r.sync()
    s0 = r.stats()
    rays0 = max(1, s0["rays"])
    per_ray = (s0["nodes"] / rays0, s0["tri_tests"] / rays0, s0["sphere_tests"] / rays0, s0["hits"] / rays0,
               rays0 / npix)
    max_stack = int(s0["max_stack"])
    r.set_flags()

    # The image reduce, twice, on a cleared image, before anything is timed: the first full-size
    # collective sets up RCCL's channels and buffers (tens of milliseconds), which must not land in
    # a timed region that is itself only tens of milliseconds long.
    reduce_warmup_ms = []
    for _ in range(2):
        r.clear_sum()
        r.sync()
        comm.barrier()
        t = time.perf_counter()
        comm.reduce_sum_image(0)
        reduce_warmup_ms.append((time.perf_counter() - t) * 1e3)

    def timed_loop(step, steps, combine):
        """`steps` steps between barriers; with `combine` the reduce to rank 0 and rank 0's finalise
        pass are part of the timed region.  Returns (seconds, rays, reduce seconds, shader clock GHz)."""
        for i in range(args.warmup):
            step(i)
        r.sync()
        comm.barrier()
        r.reset_stats()
        r.clear_sum()
        r.sync()
        comm.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        r.sync()
        t_reduce = 0.0
        if combine:
            t1 = time.perf_counter()
            comm.reduce_sum_image(0)       # RCCL over xGMI, once per run (nothing to do at one GPU)
            t_reduce = time.perf_counter() - t1
            if rank == 0:
                r.finalize_sum(steps * world)
                r.sync()
        comm.barrier()
        elapsed = time.perf_counter() - t0
        return elapsed, float(r.stats()["rays"]), t_reduce, r.shader_clock_ghz()

    def sum_step(i):
        r.render_sum(first_frame=i * world + rank, nframes=1)   # this rank's i-th frame under spp sharding

    def frame_step(i):
        r.set_frame_index(i)
        r.render()                                              # dispatch + accumulate pass

    elapsed, rays, reduce_s, clock_short = timed_loop(sum_step, args.steps, True)
    ref_frame = None
    if world == 1:
        r.reset_accumulation()
        f_elapsed, f_rays, _, _ = timed_loop(frame_step, args.steps, False)
        ref_frame = {"value": f_rays / f_elapsed / 1e6, "unit": "Mrays/s", "ms_per_step": f_elapsed / args.steps * 1e3,
                     "what": "the reference's own frame, rt_render: compute dispatch writing the sRGB sample + the "
                             "accumulate pass (renderer.py:241-288), same loop and barriers"}

    # device time of the dominant kernel, measured with HIP events on the library's own stream:
    # single launches, one at a time, outside the wall-clock region
    r.reset_stats()
    ev_ms = []
    for i in range(min(args.steps, 10)):
        r.render_sum(first_frame=i * world + rank, nframes=1)
        r.sync()
        ev_ms.append(r.stats()["last_trace_ms"])
    kernel_ms = float(np.mean(ev_ms))
    rays_per_launch = r.stats()["rays"] / len(ev_ms)

    # sustained leg: the same step for about args.sustain_seconds, so that the line also holds a
    # figure a power or thermal limit has had time to act on, with the clock the kernel ran at
    sustained = None
    if args.sustain_seconds > 0:
        n_sus = int(min(20000, max(args.steps, np.ceil(args.sustain_seconds / max(elapsed / args.steps, 1e-6)))))
        n_sus = int(comm.all_reduce([float(n_sus)], "max")[0])   # the same count on every rank
        s_elapsed, s_rays, s_reduce, clock_sus = timed_loop(sum_step, n_sus, True)
        s_tmax = comm.all_reduce([s_elapsed], "max")[0]
        s_rsum = comm.all_reduce([s_rays], "sum")[0]
        sustained = {"value": s_rsum / s_tmax / 1e6, "unit": "Mrays/s", "ms_per_step": s_tmax / n_sus * 1e3,
                     "frames": n_sus * world, "steps": n_sus, "seconds": s_tmax,
                     "shader_clock_GHz": {"timed_steps": clock_short, "sustained": clock_sus},
                     "what": "the timed step repeated for about %g s (reduce and finalise included once); the clock is "
                             "what the trace kernel's own wavefronts measured (cycle counter / 100 MHz counter)"
                             % args.sustain_seconds}
    overflows = float(r.stats()["stack_overflows"])

    # the opt-in fast mode (own tree, free traversal order: NOT the reference's algorithm, never the
    # headline) on the same workload, reported separately
    fast = None
    if world == 1:
        r.fast_mode = True
        try:
            r.set_flags(stats=True)
            r.reset_stats()
            r.set_frame_index(0)
            r.render()
            r.sync()
            fs = r.stats()
            r.set_flags()
            q_elapsed, q_rays, _, _ = timed_loop(sum_step, args.steps, True)
            r.reset_stats()
            f_ev = []
            for i in range(min(args.steps, 10)):
                r.render_sum(first_frame=i, nframes=1)
                r.sync()
                f_ev.append(r.stats()["last_trace_ms"])
            fast = {"value": q_rays / q_elapsed / 1e6, "unit": "Mrays/s", "ms_per_step": q_elapsed / args.steps * 1e3,
                    "kernel_ms": float(np.mean(f_ev)),
                    "node_visits_per_ray": fs["nodes"] / max(1, fs["rays"]),
                    "triangle_tests_per_ray": fs["tri_tests"] / max(1, fs["rays"]),
                    "what": "opt-in, default off, not the reference's data: the library's own tree under its own "
                            "traversal order; same timed step as the headline. Results differ from the reference where "
                            "its tree decides which hit is found (tests/test_gpu_fast_mode.py counts them)"}
        finally:
            r.fast_mode = False

    tmax = comm.all_reduce([elapsed, reduce_s], "max")
    tsum = comm.all_reduce([rays], "sum")
    dev, pci = r.device_info()
    crank, cranks = comm.info()
    per = comm.gather([float(dev), pci_to_number(pci), elapsed * 1e3, reduce_s * 1e3, clock_short, overflows,
                       float(crank)])
    if rank != 0:
        return None
    meas = {"elapsed": tmax[0], "rays": tsum[0], "reduce_ms_max": tmax[1] * 1e3, "reduce_warmup_ms": reduce_warmup_ms,
            "per_rank": per, "rccl_ranks": cranks, "ref_frame": ref_frame, "sustained": sustained, "fast": fast,
            "kernel_ms": kernel_ms, "rays_per_launch": rays_per_launch, "per_ray": per_ray, "variant": variant,
            "max_stack": max_stack, "build_s": build_s}
    out = assemble_c4(args, world, name, int(bufs.triangles.shape[0]), int(bufs.bvh.shape[0]), rehearse, meas)
    if not args.no_cpu_baseline and world == 1:
        from raytracer_amd.camera import Camera

        threads = min(16, os.cpu_count() or 1)
        out["cpu_baseline"] = cpu_baseline(bufs, Camera().uniforms_f32(), args.width, args.height, args.bounces,
                                           threads, args.cpu_seconds)
    return out


def assemble_c4(args, world, name, n_tris, n_nodes, rehearse, m):
    """The JSON line of a c4 run from rank 0's measurements (no GPU involved: testable on its own)."""
    per = np.asarray(m["per_rank"], dtype=np.float64).reshape(world, -1)
    steps = args.steps
    out = {
        "metric": "Mrays/s", "value": m["rays"] / m["elapsed"] / 1e6, "unit": "Mrays/s",
        "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": m["elapsed"] / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32",
        "data": "synthetic" if not rehearse else "synthetic; REHEARSAL on one shared GPU with host-file collectives, not a measurement",
        "config": {"workload": f"{name} {args.width}x{args.height} 1spp/frame {args.bounces} bounces, "
                               f"default reference camera; stands in for sponza_full.obj (not distributed)",
                   "step": "one frame per GPU: compute dispatch accumulating linear radiance (rt_render_sum); the timed "
                           "region ends with the image combine (one RCCL reduce when n_gpus > 1) and rank 0's "
                           "finalise pass (sRGB mean image) -- the same step at every n_gpus",
                   "triangles": n_tris, "bvh_nodes": n_nodes,
                   "rays_per_frame": m["rays_per_launch"], "frames_per_step_per_gpu": 1,
                   "parallelism": f"spp-sharded x{world}, scene replicated, 1 RCCL reduce (C ABI)" if world > 1 else "single GPU",
                   "bvh_build_s": round(m["build_s"], 3)},
        "ms_per_frame": m["elapsed"] / steps * 1e3 / world,
        "reduce_ms": m["reduce_ms_max"],
        "reduce_warmup_ms": m["reduce_warmup_ms"],
        "rccl_ranks": m["rccl_ranks"],
        "ranks": [{"rank": k, "comm_rank": int(per[k, 6]), "device": int(per[k, 0]), "pci_bus_id": number_to_pci(per[k, 1]),
                   "timed_region_ms": per[k, 2], "reduce_ms": per[k, 3], "shader_clock_GHz": per[k, 4],
                   "stack_overflows": int(per[k, 5])} for k in range(world)],
        "stack_overflows": int(per[:, 5].sum()),
        "max_stack": m["max_stack"],
        "sustained": m["sustained"],
        "reference_frame": m["ref_frame"],
        "roofline": dict(roofline_block(name, args, m["kernel_ms"], m["rays_per_launch"], m["per_ray"]),
                         kernel="trace_leafstep_kernel" if m["variant"] == "leaf steps" else "trace_kernel",
                         traversal_variant=m["variant"]),
        "cpu_baseline": None,
    }
    if world == 1:
        out["fast_mode"] = m["fast"]
    return out


def c5_split(world: int):
    """(tile shards, sample shards) of a strong-scaling run: two tile shards whenever the rank count
    is even, the rest by samples; rank -> (rank % tiles, rank // tiles)."""
    tiles = 2 if world % 2 == 0 else 1
    return tiles, world // tiles


def run_c5(args, r, comm, rank, world, name, bufs, build_s, rehearse):
    """One 4K image of `--spp` samples per pixel split over the ranks by tiles and samples."""
    spp = args.spp
    tiles, phases = c5_split(world)
    shard, phase = rank % tiles, rank // tiles

    def image():
        r.clear_sum()
        t0 = time.perf_counter()
        r.render_sum(first_frame=0, nframes=spp, shard=shard, nshards=tiles, frame_stride=phases, frame_phase=phase)
        r.sync()
        t_render = time.perf_counter() - t0
        comm.reduce_sum_image(0)
        t_reduce = time.perf_counter() - t0 - t_render
        if rank == 0:
            r.finalize_sum(spp)
            r.sync()
        return t_render, t_reduce, time.perf_counter() - t0

    for _ in range(args.warmup):
        image()
    comm.barrier()
    r.reset_stats()
    r.sync()
    comm.barrier()
    t0 = time.perf_counter()
    renders, reduces = [], []
    for _ in range(args.steps):
        a, b, _ = image()
        renders.append(a)
        reduces.append(b)
    comm.barrier()
    elapsed = time.perf_counter() - t0
    rays = float(r.stats()["rays"])
    my_render = float(np.mean(renders))
    tmax = comm.all_reduce([elapsed, my_render], "max")
    tsum = comm.all_reduce([rays, my_render], "sum")
    dev, pci = r.device_info()
    crank, cranks = comm.info()
    per = comm.gather([float(dev), pci_to_number(pci), my_render * 1e3, float(np.mean(reduces)) * 1e3,
                       r.shader_clock_ghz(), float(r.stats()["stack_overflows"]), float(crank)])
    if rank != 0:
        return None
    return {
        "metric": "Mrays/s", "value": tsum[0] / tmax[0] / 1e6, "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": tmax[0] / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32",
        "data": "synthetic" if not rehearse else "synthetic; REHEARSAL on one shared GPU with host-file collectives, not a measurement",
        "config": {"workload": f"{name} {args.width}x{args.height} {spp} spp {args.bounces} bounces, default reference "
                               f"camera; stands in for sponza_full.obj (not distributed)",
                   "step": "one whole image: first launch -> reduce -> rank 0's finalised sRGB mean image",
                   "triangles": int(bufs.triangles.shape[0]), "bvh_nodes": int(bufs.bvh.shape[0]),
                   "parallelism": f"{tiles} tile shard(s) (64x64 Morton round-robin) x {phases} sample shard(s), scene "
                                  f"replicated, 1 RCCL reduce (C ABI) per image",
                   "bvh_build_s": round(build_s, 3)},
        "time_to_image_ms": tmax[0] / args.steps * 1e3,
        "per_rank_render_ms": [float(x) for x in per[:, 2]],
        "reduce_ms": float(per[:, 3].max()),
        "reduce_ms_rank0": float(np.mean(reduces)) * 1e3,
        "rccl_ranks": cranks,
        "ranks": [{"rank": k, "comm_rank": int(per[k, 6]), "device": int(per[k, 0]), "pci_bus_id": number_to_pci(per[k, 1]),
                   "render_ms": per[k, 2], "reduce_ms": per[k, 3], "shader_clock_GHz": per[k, 4],
                   "stack_overflows": int(per[k, 5])} for k in range(world)],
        "stack_overflows": int(per[:, 5].sum()),
        "imbalance_max_over_mean": float(tmax[1] / (tsum[1] / world)),
        "roofline": None, "cpu_baseline": None,
    }


def worker(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) were started (WORLD_SIZE={world})")
    # RT_BENCH_REHEARSE=1: every rank uses GPU 0 and the collectives go through host files.
    # For rehearsing the N>1 control flow on a one-GPU box; never a measurement.
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    # the host driver only supports dmabuf IPC; RCCL's peer buffers need this set before HIP starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:
        os.environ.setdefault("NCCL_DEBUG", "WARN")   # RCCL's own reason, should the communicator fail to form

    from raytracer_amd import dist as D
    from raytracer_amd.camera import Camera
    from raytracer_amd.renderer import Renderer

    name, bufs, build_s = build_workload(args)
    r = Renderer(device=0 if rehearse else local_rank, camera=Camera(), singleton=False)
    try:
        r.update_size(args.width, args.height)
        bufs.upload(r)
        r.bounces = args.bounces
        # The library times its two traversal variants once a scene has rendered 64 frames
        # (rt_traversal_variant in the header) and keeps the faster: let that happen before any timing.
        r.render_sum(first_frame=0, nframes=64)
        r.sync()
        r.clear_sum()
        r.reset_stats()
        comm = D.Comm(r, rank, world, rehearse=rehearse)
        joined = comm.all_reduce([1.0], "sum")[0]
        if int(round(joined)) != args.gpus:
            raise SystemExit(f"bench.py: {joined:.0f} rank(s) joined, --gpus {args.gpus} expected")
        run = run_c5 if args.config == "c5" else run_c4
        out = run(args, r, comm, rank, world, name, bufs, build_s, rehearse)
        if out is not None:
            print(json.dumps(out), flush=True)
        comm.finish()
    finally:
        r.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["c4", "c5"], default="c4",
                    help="c4: 1080p frames, weak scaling (default); c5: one 4K 256-spp image, strong scaling")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=256, help="c5: samples per pixel of the image")
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--obj", default=None, help="render this OBJ (e.g. sponza_full.obj) instead of sponza_synth")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="time budget of the CPU baseline sample")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="c4: length of the sustained leg after the timed steps (0 = none)")
    args = ap.parse_args()
    c5 = args.config == "c5"
    if args.steps is None:
        args.steps = 2 if c5 else 20
    if args.warmup is None:
        args.warmup = 1 if c5 else 3
    if args.width is None:
        args.width = 3840 if c5 else 1920
    if args.height is None:
        args.height = 2160 if c5 else 1080
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the ranks here, before anything in this process touches a GPU
        from raytracer_amd import dist as D

        sys.exit(D.spawn_ranks([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], args.gpus))
    worker(args)


if __name__ == "__main__":
    main()
