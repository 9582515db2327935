#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json's metric, "Mrays/s + ms/frame, sponza_full 1080p 4-bounce"): the
Sponza-scale scene at 1920x1080, 1 sample per pixel per frame, 4 bounces, reference default
camera.  `models/sponza_full.obj` is not distributed with the reference, so the deterministic
stand-in `sponza_synth` (about 265 k triangles, raytracer_amd/synth.py) is rendered unless
--obj names a model; the JSON line says which.

A step = one full frame per GPU.  With N GPUs (one process per GPU, launched by
torch.distributed.run) rank r renders the frames f with f % N == r — samples-per-pixel sharding,
scene replicated — into its own fp32 linear-radiance sum image, and the images are combined
with one RCCL reduce inside the timed region; per-GPU work is fixed, so scaling is weak and
`value` is the rays all ranks traced per second.
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


def cpu_baseline(bufs, cam_uniforms, width, height, bounces, threads):
    """The CPU restatement (oracle/, OpenMP) timed on a bounded sample of the same workload:
    the same scene, camera and bounce count at reduced resolution."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as O

    u = np.concatenate([cam_uniforms[0], cam_uniforms[1], bufs.scene_min, bufs.scene_extent]).astype(np.float32)
    sc = O.OracleScene(bvh=bufs.bvh.tobytes(), tris=bufs.triangles.tobytes(), spheres=bufs.spheres.tobytes(),
                       mats=bufs.materials.tobytes(), uniforms=u)
    w, h = width // 2, height // 2
    sc.render(64, 32, 0, bounces, threads=threads)  # warm up the thread pool
    t = time.time()
    frames = 0
    rays = 0
    while time.time() - t < 8.0 or frames < 1:
        _, cnt = sc.render(w, h, frames, bounces, want_counters=True, threads=threads)
        rays += cnt["rays"]
        frames += 1
    dt = time.time() - t
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{frames} frame(s) of the same scene/camera/bounces at {w}x{h} "
                      f"({rays} rays in {dt:.1f} s, OpenMP C restatement of the reference shader)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--obj", default=None, help="render this OBJ (e.g. sponza_full.obj) instead of sponza_synth")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # RT_BENCH_REHEARSE=1: every rank uses GPU 0 and the collectives run over gloo on host copies.
    # For rehearsing the N>1 control flow on a one-GPU box; never a measurement.
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    dist = torch = None
    if world > 1:
        import torch
        import torch.distributed as dist

        if rehearse:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def collective(fn, t, **kw):
        """Runs a torch.distributed collective on a device tensor (through host memory when rehearsing)."""
        if rehearse:
            h = t.cpu()
            fn(h, **kw)
            t.copy_(h)
        else:
            fn(t, **kw)

    from raytracer_amd.camera import Camera
    from raytracer_amd.renderer import Renderer

    name, bufs, build_s = build_workload(args)
    r = Renderer(device=local_rank, camera=Camera(), singleton=False)
    r.update_size(args.width, args.height)
    bufs.upload(r)
    r.bounces = args.bounces
    npix = args.width * args.height

    # per-ray work of this workload, from one untimed instrumented frame (counter kernel variant)
    r.set_flags(stats=True)
    r.set_frame_index(0)
    r.camera.changed = False
    r.render()
    r.sync()
    s0 = r.stats()
    rays0 = max(1, s0["rays"])
    n_node, n_tri, n_sph = s0["nodes"] / rays0, s0["tri_tests"] / rays0, s0["sphere_tests"] / rays0
    hit_frac = s0["hits"] / rays0
    rays_per_sample = rays0 / npix
    # algorithmic bytes per ray in the reference's layouts: 64 B per node visit, 64 B per triangle
    # test, 32 B per sphere test, 12 B of material per hit, 16 B of image per pixel-sample
    bytes_per_ray = 64.0 * n_node + 64.0 * n_tri + 32.0 * n_sph + 12.0 * hit_frac + 16.0 / rays_per_sample
    r.set_flags()

    if world > 1:
        sum_t = torch.zeros((args.height, args.width, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
        sum_ptr = sum_t.data_ptr()
    else:
        sum_t, sum_ptr = None, None
        r.clear_sum()

    def step(i):
        # frame index of this rank's i-th frame under spp sharding
        r.render_sum(first_frame=i * world + rank, nframes=1, sum_ptr=sum_ptr)

    def fence():
        r.sync()
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    r.reset_stats()
    if world > 1:
        sum_t.zero_()
        torch.cuda.synchronize()
    else:
        r.clear_sum()
    fence()
    trace_ms = 0.0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        # the per-launch device time of the trace kernel is read after the timed region
    r.sync()
    if world > 1:
        collective(dist.reduce, sum_t, dst=0, op=dist.ReduceOp.SUM)  # RCCL over xGMI, once per run
    fence()
    elapsed = time.perf_counter() - t0
    if rank == 0:
        r.finalize_sum(args.steps * world, sum_ptr)
        r.sync()
    st = r.stats()
    # device time of the dominant kernel, measured with HIP events on the library's own stream:
    # time K single launches back to back outside the wall-clock region
    r.reset_stats()
    ev_ms = []
    for i in range(min(args.steps, 10)):
        r.render_sum(first_frame=i * world + rank, nframes=1, sum_ptr=sum_ptr)
        r.sync()
        ev_ms.append(r.stats()["last_trace_ms"])
    kernel_ms = float(np.mean(ev_ms))
    rays_per_launch = r.stats()["rays"] / len(ev_ms)

    my = np.array([elapsed, float(st["rays"])], dtype=np.float64)
    if world > 1:
        t = torch.tensor(my, device=f"cuda:{local_rank}")
        tmax = t.clone()
        collective(dist.all_reduce, tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        collective(dist.all_reduce, tsum, op=dist.ReduceOp.SUM)
        elapsed_max, total_rays = float(tmax[0]), float(tsum[1])
    else:
        elapsed_max, total_rays = elapsed, float(st["rays"])

    if rank == 0:
        achieved = rays_per_launch * bytes_per_ray / (kernel_ms * 1e-3) / 1e9
        # memory-side bytes per launch come from separate rocprofv3 --pmc passes (they cannot be
        # collected inside this process); the committed summary is used when it is for this workload
        traffic = None
        tj = ROOT / "profiles" / "r1" / "traffic.json"
        key = f"{name} {args.width}x{args.height} {args.bounces}b"
        if tj.exists():
            t = json.loads(tj.read_text())
            if t.get("workload_key") == key:
                # FETCH_SIZE tallies a 128-byte line request as 64 B on gfx950 (calibrated for this
                # access pattern with tools/microbench/fetch_calib.hip), hence the correction factor
                traffic = (t["fetch_size_kb"] * t.get("fetch_correction", 2.0) + t["write_size_kb"]) * 1024.0
        out = {
            "metric": "Mrays/s", "value": total_rays / elapsed_max / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not rehearse else "synthetic; REHEARSAL on one shared GPU over gloo, not a measurement",
            "config": {"workload": f"{name} {args.width}x{args.height} 1spp/frame {args.bounces} bounces, "
                                   f"default reference camera; stands in for sponza_full.obj (not distributed)",
                       "triangles": int(bufs.triangles.shape[0]), "bvh_nodes": int(bufs.bvh.shape[0]),
                       "rays_per_frame": rays_per_launch, "frames_per_step_per_gpu": 1,
                       "parallelism": f"spp-sharded x{world}, scene replicated, 1 RCCL reduce" if world > 1 else "single GPU",
                       "bvh_build_s": round(build_s, 3)},
            "ms_per_frame": kernel_ms,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": rays_per_launch * bytes_per_ray,
                         "kernel": "trace_kernel", "kernel_ms": kernel_ms,
                         "kernel_ms_how": "HIP events on the launch's stream, launches issued one at a time after "
                                          "the timed region; inside the timed region two launches overlap, so a "
                                          "kernel trace of this command shows longer per-kernel durations "
                                          "(RT_OVERLAP=0 gives matching ones)",
                         "bytes_per_ray": bytes_per_ray,
                         "per_ray": {"node_visits": n_node, "triangle_tests": n_tri, "sphere_tests": n_sph,
                                     "hit_fraction": hit_frac, "rays_per_pixel_sample": rays_per_sample}},
        }
        if not args.no_cpu_baseline and world == 1:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(bufs, Camera().uniforms_f32(), args.width, args.height, args.bounces,
                                               threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
