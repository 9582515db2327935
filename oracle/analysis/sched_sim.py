#!/usr/bin/env python3
"""Offline study of wavefront scheduling policies for the traversal kernel.

Traces every ray of one frame of the bench workload with the CPU restatement (node visits and
leaf intersections per visit), then replays 64-lane wavefronts under different policies and
reports wavefront-instructions per ray with a simple cost model (one node step, one leaf step,
one shading step).  TEST INFRASTRUCTURE (part of oracle/): analysis tooling only, never on the product path.
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
import bench  # noqa: E402
import oracle as O  # noqa: E402

REC = 256
C_NODE, C_LEAF, C_SHADE, C_GEN = 185, 85, 700, 60


def trace_frame(args):
    from raytracer_amd.camera import Camera

    name, bufs, _ = bench.build_workload(args)
    inv_vp, eye = Camera().uniforms_f32()
    u = np.concatenate([inv_vp, eye, bufs.scene_min, bufs.scene_extent]).astype(np.float32)
    sc = O.OracleScene(bvh=bufs.bvh.tobytes(), tris=bufs.triangles.tobytes(), spheres=bufs.spheres.tobytes(),
                       mats=bufs.materials.tobytes(), uniforms=u)
    w, h = args.width, args.height
    img = np.zeros((h, w, 4), np.float32)
    tr = np.zeros((h, w, 8, REC), np.uint8)
    L = O.lib()
    L.rto_render_trace.restype = C.c_int
    rc = L.rto_render_trace(C.byref(sc.s), w, h, C.c_float(0.0), C.c_float(args.bounces), C.c_void_p(img.ctypes.data),
                            C.c_void_p(tr.ctypes.data), 0)
    assert rc == 0
    return tr


def wave_rays(tr, tx, ty):
    """[lane][bounce] -> (nsteps, leafcounts) for the 8x8 pixel tile (tx, ty)."""
    t = tr[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8].reshape(64, 8, REC)
    n = t[:, :, 0].astype(np.int32) | (t[:, :, 1].astype(np.int32) << 8)
    return n, t[:, :, 2:] & 0x7F   # (bit 7 of a visit's byte: the popped node lay beyond the best hit)


def lockstep(n, lc):
    """Current megakernel: all lanes do bounce b together."""
    cost = nodes = leaves = rays = 0
    for b in range(8):
        nb = n[:, b]
        alive = nb > 0
        if not alive.any():
            break
        rays += int(alive.sum())
        m = int(nb.max())
        steps = min(m, REC - 2)
        # a lane takes part in step j while j < its own count
        idx = np.arange(steps)[None, :]
        act = idx < nb[:, None]
        l = np.where(act, lc[:, b, :steps], 0).max(axis=0)
        nodes += m
        leaves += int(l.sum())
        cost += m * C_NODE + int(l.sum()) * C_LEAF + C_SHADE
    return cost, nodes, leaves, rays


def lockstep_stall(n, lc, threshold):
    """Lockstep per bounce, but a lane with pending leaves waits (the others go on with node steps) until
    `threshold` lanes wait or nobody can do a node step; then one leaf step serves one leaf per waiting lane."""
    cost = nodes = leaves = rays = 0
    for b in range(8):
        nb = n[:, b].copy()
        if not (nb > 0).any():
            break
        rays += int((nb > 0).sum())
        step = np.zeros(64, np.int32)
        pend = np.zeros(64, np.int32)
        while True:
            ready = (step < nb) & (pend == 0)
            waiting = pend > 0
            if not ready.any() and not waiting.any():
                break
            if waiting.sum() >= threshold or not ready.any():
                pend[waiting] -= 1
                leaves += 1
                cost += C_LEAF
                continue
            lanes = np.nonzero(ready)[0]
            pend[lanes] = lc[lanes, b, np.minimum(step[lanes], REC - 3)]
            step[lanes] += 1
            nodes += 1
            cost += C_NODE
        cost += C_SHADE
    return cost, nodes, leaves, rays


def persistent(waves, shade_threshold, pool_tiles):
    """State machine: a wavefront owns pool_tiles tiles (64 pixels each); a lane whose path ends takes the next
    unassigned pixel.  Each iteration is a shading step (when >= shade_threshold lanes wait for one, or nobody
    traverses) or a node step followed by the leaf steps it needs."""
    cost = nodes = leaves = rays = shades = 0
    for g in range(0, len(waves) - pool_tiles + 1, pool_tiles):
        pool_n = np.concatenate([waves[g + k][0] for k in range(pool_tiles)])  # [pixels][8]
        pool_lc = np.concatenate([waves[g + k][1] for k in range(pool_tiles)])
        npx = pool_n.shape[0]
        nxt = 64
        pix = np.arange(64)
        bnc = np.zeros(64, np.int32)
        step = np.zeros(64, np.int32)
        # state: 0 traverse, 1 wait for shade, 2 done
        state = np.zeros(64, np.int32)
        rays += 64
        while True:
            trav = state == 0
            wait = state == 1
            if not trav.any() and not wait.any():
                break
            if wait.sum() >= shade_threshold or not trav.any():
                cost += C_SHADE
                shades += 1
                for l in np.nonzero(wait)[0]:
                    b = bnc[l] + 1
                    if b < 8 and pool_n[pix[l], b] > 0:
                        bnc[l], step[l], state[l] = b, 0, 0
                        rays += 1
                    elif nxt < npx:
                        pix[l], bnc[l], step[l], state[l] = nxt, 0, 0, 0
                        nxt += 1
                        rays += 1
                    else:
                        state[l] = 2
                continue
            lanes = np.nonzero(trav)[0]
            s = step[lanes]
            lcs = pool_lc[pix[lanes], bnc[lanes], np.minimum(s, REC - 3)]
            nl = int(lcs.max())
            cost += C_NODE + nl * C_LEAF
            nodes += 1
            leaves += nl
            step[lanes] += 1
            fin = step[lanes] >= pool_n[pix[lanes], bnc[lanes]]
            state[lanes[fin]] = 1
    return cost, nodes, leaves, rays, shades


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--obj", default=None)
    ap.add_argument("--sample", type=int, default=1500, help="wavefronts to replay")
    args = ap.parse_args()
    tr = trace_frame(args)
    tiles_x, tiles_y = args.width // 8, args.height // 8
    rng = np.random.default_rng(1)
    # groups of 8 horizontally adjacent tiles, so that pools are spatially compact
    starts = rng.choice(tiles_y * (tiles_x // 8), size=args.sample // 8, replace=False)
    waves = []
    for s in starts:
        ty, gx = divmod(int(s), tiles_x // 8)
        for k in range(8):
            waves.append(wave_rays(tr, gx * 8 + k, ty))
    c = nd = lf = ry = 0
    for n, lc in waves:
        a, b, d, e = lockstep(n, lc)
        c += a; nd += b; lf += d; ry += e
    print(f"lockstep:            {c / ry:7.1f} wave-instr/ray   node steps/ray {nd / ry:.3f}  leaf steps/ray {lf / ry:.3f}")
    for thr in (1, 2, 4, 8, 16, 32):
        c = nd = lf = ry = 0
        for n, lc in waves:
            a, b, d, e = lockstep_stall(n, lc, thr)
            c += a; nd += b; lf += d; ry += e
        print(f"lockstep stall>={thr:2d}:   {c / ry:7.1f} wave-instr/ray   node steps/ray {nd / ry:.3f}  leaf steps/ray {lf / ry:.3f}")
    for pool in ():
        for thr in (8, 16, 32, 48):
            a, b, d, e, sh = persistent(waves, thr, pool)
            print(f"persistent pool={pool} shade>={thr:2d}: {a / e:7.1f} wave-instr/ray   node steps/ray {b / e:.3f}  "
                  f"leaf steps/ray {d / e:.3f}  shade steps/ray {sh / e:.4f}")


if __name__ == "__main__":
    main()
