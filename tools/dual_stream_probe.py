import sys, time, json, argparse
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
from raytracer_amd.renderer import Renderer
from raytracer_amd.camera import Camera
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=2); ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
class A: obj=None; detail=1.0
name, bufs, _ = bench.build_workload(A)
rs = []
for k in range(a.n):
    r = Renderer(device=0, camera=Camera(), singleton=False)
    r.update_size(1920, 1080); bufs.upload(r); r.bounces = 4; r.clear_sum(); r.set_flags(stats=True)
    rs.append(r)
for i in range(6):
    rs[i % a.n].render_sum(first_frame=i, nframes=1)
for r in rs: r.sync(); r.reset_stats()
t0 = time.perf_counter()
for i in range(a.steps):
    rs[i % a.n].render_sum(first_frame=i, nframes=1)
for r in rs: r.sync()
dt = time.perf_counter() - t0
rays = sum(r.stats()["rays"] for r in rs)
print(json.dumps({"contexts": a.n, "ms_per_frame": dt / a.steps * 1e3, "Mrays/s": rays / dt / 1e6}))
