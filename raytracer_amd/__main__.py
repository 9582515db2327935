"""Headless front end: renders a scene file of the reference's schema to a PNG.

  python -m raytracer_amd SCENE.toml --out image.png [--width W --height H --spp N --bounces B]
         [--mesh ./models/a.obj=./models/b.obj ...] [--first-frame F] [--device D] [--fast] [--gpus G]

It stands where the reference's window loop stands (src/main.py:26-57: load_scene, then
Renderer.render() once per window frame, each adding one sample per pixel to the running mean), with
the window replaced by a file: `--spp` frames are rendered with `time` = first-frame, first-frame + 1,
... and the accumulation image is written as the upscale pass would present it (shaders/
upscale_fragment.glsl:8-17 with sharpness 0 is a copy).  Mesh paths in a scene file are relative to
the working directory, as in the reference (scenes/test.toml: "./models/...").

One GPU: the reference's own frames (Renderer.render(): compute dispatch + running mean).  `--gpus G`:
one process per GPU, the samples dealt round-robin (rank r renders the frames f with f % G == r),
linear radiance summed per GPU and combined with one RCCL reduce; rank 0 writes the mean image.  That
image differs from the running mean only by the rounding of the sRGB round trips the running mean
makes after every frame.  A JSON line on stdout reports what was rendered and how long it took.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path


def _worker(args) -> int:
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"raytracer_amd: --gpus {args.gpus} but {world} rank(s) were started")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # RT_BENCH_REHEARSE=1: every rank on GPU 0, collectives through host files (RCCL refuses several
    # ranks on one device): for rehearsing the multi-rank path on a one-GPU machine
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"

    from . import dist as D, scene as S
    from .camera import Camera
    from .png import save_png
    from .renderer import Renderer

    override = {}
    for item in args.mesh or []:
        old, sep, new = item.partition("=")
        if not sep:
            raise SystemExit(f"raytracer_amd: --mesh wants OLD=NEW, got {item!r}")
        override[old] = new
    r = Renderer(device=args.device if world == 1 else (0 if rehearse else local_rank), camera=Camera(), singleton=False)
    try:
        t0 = time.perf_counter()
        r.update_size(args.width, args.height)
        S.load_scene(args.scene, renderer=r, mesh_override=override or None)
        r.bounces = args.bounces
        r.fast_mode = bool(args.fast)
        t_load = time.perf_counter() - t0
        r.reset_stats()
        t1 = time.perf_counter()
        if world == 1:
            r.set_frame_index(args.first_frame)
            r.camera.changed = False
            r.rendered_frames = 0
            r.render(frames=args.spp)
            img = r.read_accumulation()
            rays = float(r.stats()["rays"])
            overflows = float(r.stats()["stack_overflows"])
        else:
            comm = D.Comm(r, rank, world, rehearse=rehearse)
            r.clear_sum()
            r.render_sum(args.first_frame, args.spp, frame_stride=world, frame_phase=rank)
            comm.reduce_sum_image(0)
            rays, overflows = comm.all_reduce([float(r.stats()["rays"]), float(r.stats()["stack_overflows"])], "sum")
            img = None
            if rank == 0:
                r.finalize_sum(args.spp)
                img = r.read_accumulation()
            comm.finish()
        t_render = time.perf_counter() - t1
        if rank == 0:
            save_png(args.out, img)
            print(json.dumps({"scene": str(args.scene), "out": str(args.out), "size": [args.width, args.height],
                              "spp": args.spp, "bounces": args.bounces, "first_frame": args.first_frame,
                              "gpus": world, "fast_mode": bool(args.fast), "rays": rays,
                              "load_and_build_s": round(t_load, 4), "render_s": round(t_render, 4),
                              "Mrays_per_s": rays / t_render / 1e6, "stack_overflows": int(overflows)}), flush=True)
    finally:
        r.close()
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m raytracer_amd", description=__doc__.split("\n\n")[0])
    ap.add_argument("scene", help="scene file in the reference's TOML schema")
    ap.add_argument("--out", required=True, help="PNG to write")
    ap.add_argument("--width", type=int, default=1280)    # the reference's window: src/window/window.py:35
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--spp", type=int, default=64, help="frames to accumulate (one sample per pixel each)")
    ap.add_argument("--bounces", type=int, default=10)    # src/renderer/renderer.py:112
    ap.add_argument("--first-frame", type=int, default=0, help="value of the `time` uniform of the first frame")
    ap.add_argument("--mesh", action="append", metavar="OLD=NEW", help="replace a [[meshes]] path of the scene file")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--fast", action="store_true", help="opt-in fast mode: the library's own tree, not the reference's")
    ap.add_argument("--gpus", type=int, default=1)
    args = ap.parse_args(argv)
    if args.spp < 1 or args.gpus < 1:
        raise SystemExit("raytracer_amd: --spp and --gpus must be at least 1")
    if not Path(args.scene).exists():
        raise SystemExit(f"raytracer_amd: no such scene file: {args.scene}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from . import dist as D   # starts the ranks before anything in this process touches a GPU

        return D.spawn_ranks([sys.executable, "-m", "raytracer_amd"] + list(sys.argv[1:] if argv is None else argv), args.gpus)
    return _worker(args)


if __name__ == "__main__":
    sys.exit(main())
