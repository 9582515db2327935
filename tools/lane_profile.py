#!/usr/bin/env python3
"""Lane-utilisation breakdown of the megakernel on the bench workload.

Needs a library built with -DRT_LANE_PROFILE=1 (RT_MI355X_LIB=build/ab/prof.so): in that build
the statistics fields box_tests / hits / stack_overflows count wavefront-level executions of a
node step / a leaf test / a bounce iteration, so that lanes-per-execution can be derived.
"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--obj", default=None)
    args = ap.parse_args()
    from raytracer_amd.renderer import Renderer

    name, bufs, _ = bench.build_workload(args)
    r = Renderer(device=0, singleton=False)
    bufs.upload(r)
    r.update_size(args.width, args.height)
    r.bounces = args.bounces
    r.set_flags(stats=True)
    r.reset_stats()
    r.set_frame_index(0)
    r.render()
    s = r.stats()
    out = {"workload": name, "rays": s["rays"], "node_lane_visits": s["nodes"],
           "leaf_lane_tests": s["tri_tests"] + s["sphere_tests"],
           "wave_node_steps": s["box_tests"], "wave_leaf_steps": s["hits"], "wave_bounces": s["stack_overflows"]}
    out["lanes_per_node_step"] = out["node_lane_visits"] / max(1, out["wave_node_steps"])
    out["lanes_per_leaf_step"] = out["leaf_lane_tests"] / max(1, out["wave_leaf_steps"])
    out["lanes_per_bounce"] = out["rays"] / max(1, out["wave_bounces"])
    out["leaf_steps_per_node_step"] = out["wave_leaf_steps"] / max(1, out["wave_node_steps"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
