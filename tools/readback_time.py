import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from raytracer_amd.renderer import Renderer
from util import load_golden_scene, render_frame
r = Renderer(device=0, singleton=False)
load_golden_scene(r, "gally")
for (w, h) in ((1920, 1080), (3840, 2160)):
    render_frame(r, w, h, 0, 4)
    for name, fn in (("read_accumulation (copy into numpy)", r.read_accumulation), ("map_image (page-locked view)", r.map_image)):
        fn()
        t = []
        for _ in range(10):
            t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
        print(f"{w}x{h} {name}: median {np.median(t)*1e3:.2f} ms, {w*h*16/np.median(t)/1e9:.1f} GB/s")
r.close()
