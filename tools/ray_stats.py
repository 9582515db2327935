#!/usr/bin/env python3
"""Per-ray work counters of the bench workload (one instrumented frame): tools/ray_stats.py [--fast]
RT_MI355X_LIB selects the library variant."""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--obj", default=None)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--fast", action="store_true")
    args = ap.parse_args()
    from raytracer_amd.camera import Camera
    from raytracer_amd.renderer import Renderer

    _, bufs, _ = bench.build_workload(args)
    r = Renderer(device=0, camera=Camera(), singleton=False)
    try:
        r.update_size(args.width, args.height)
        bufs.upload(r)
        r.bounces = args.bounces
        r.fast_mode = args.fast
        r.set_flags(stats=True)
        r.set_frame_index(0)
        r.camera.changed = False
        r.render()
        r.sync()
        s = r.stats()
        rays = max(1, s["rays"])
        print(json.dumps({k: (v / rays if isinstance(v, (int, float)) and k != "rays" else v) for k, v in s.items()}))
    finally:
        r.close()


if __name__ == "__main__":
    main()
