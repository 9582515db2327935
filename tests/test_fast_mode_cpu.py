"""Fast mode (opt-in, not the reference's data) without a GPU: the host builder of its tree and
the CPU restatement of the traversal over it.

* structure: every primitive appears exactly once, every child box decodes (with the arithmetic the
  traversal uses) to a box that contains its primitives' boxes, i.e. the tree is conservative;
* against the reference's tree on the same rays: the traversal over the fast tree never loses a
  hit (it may find a nearer one where the reference's nearest-rounded boxes hide it, and those
  pixels are counted)."""

import numpy as np
import pytest

import oracle as O
from raytracer_amd import scene as S
from util import GOLDEN

SCENES = ("debug", "debug_box", "gally", "prims", "matlab")


def _buffers(scene):
    d = GOLDEN / scene
    rd = lambda n: (d / n).read_bytes() if (d / n).exists() else b""
    return rd("tris.bin"), rd("spheres.bin"), rd("boxes.bin")


def _prim_boxes(tris, spheres, boxes):
    """meta word -> (lo, hi) of every uploaded primitive"""
    out = {}
    t = np.frombuffer(tris, np.float32).reshape(-1, 16)
    for i, r in enumerate(t):
        v = np.stack([r[0:3], r[4:7], r[8:11]])
        out[i] = (v.min(0), v.max(0))
    sp = np.frombuffer(spheres, np.float32).reshape(-1, 8)
    for i, r in enumerate(sp):
        rad = np.sqrt(r[4])
        out[(1 << 30) | i] = (r[0:3] - rad, r[0:3] + rad)
    bx = np.frombuffer(boxes, np.float32).reshape(-1, 8)
    for i, r in enumerate(bx):
        out[(2 << 30) | i] = (np.minimum(r[0:3], r[4:7]), np.maximum(r[0:3], r[4:7]))
    return out


@pytest.mark.parametrize("scene", SCENES)
def test_tree_is_complete_and_conservative(scene):
    tris, spheres, boxes = _buffers(scene)
    nodes = S.build_fast_bvh(tris, spheres, boxes)
    pbox = _prim_boxes(tris, spheres, boxes)
    assert np.all(nodes[-1, 12:] == 0xFFFFFFFF)          # the sentinel
    n = nodes.shape[0] - 1
    seen = {}
    visited = np.zeros(n, np.int32)
    # walk the tree: (node, decoded boxes of the path from the root to it)
    stack = [(0, [])]
    while stack:
        ni, path = stack.pop()
        visited[ni] += 1
        nd = nodes[ni]
        origin = nd[0:3].view(np.float32).astype(np.float64)
        scale = np.array([np.array([((nd[3] >> (8 * a)) & 0xFF) << 23], np.uint32).view(np.float32)[0] for a in range(3)],
                         np.float64)
        for c in range(4):
            m = int(nd[12 + c])
            if m == 0xFFFFFFFF:
                continue
            lo = np.array([origin[a] + ((int(nd[4 + 2 * a]) >> (8 * c)) & 0xFF) * scale[a] for a in range(3)])
            hi = np.array([origin[a] + ((int(nd[5 + 2 * a]) >> (8 * c)) & 0xFF) * scale[a] for a in range(3)])
            assert np.all(lo <= hi)
            if (m >> 30) == 3:
                assert (m & 0x3FFFFFFF) < n
                stack.append((m & 0x3FFFFFFF, path + [(lo, hi)]))
            else:
                seen[m] = seen.get(m, 0) + 1
                plo, phi = pbox[m]
                # a ray that hits the primitive enters every box on the way down to it
                for blo, bhi in path + [(lo, hi)]:
                    assert np.all(blo <= plo) and np.all(bhi >= phi), (ni, c, m)
    assert sorted(seen) == sorted(pbox) and all(v == 1 for v in seen.values())   # every primitive exactly once
    assert np.all(visited == 1)


def test_empty_and_single_primitive_scenes():
    nodes = S.build_fast_bvh()
    assert nodes.shape == (2, 16) and np.all(nodes[:, 12:] == 0xFFFFFFFF)    # an empty root and the sentinel
    tris, _, _ = _buffers("debug")
    nodes = S.build_fast_bvh(tris[:64])
    assert nodes.shape == (2, 16) and nodes[0, 12] == 0 and np.all(nodes[0, 13:] == 0xFFFFFFFF)  # one leaf: triangle 0


@pytest.mark.parametrize("scene,w,h,bounces", [("debug", 160, 88, 3), ("gally", 320, 180, 4), ("prims", 160, 88, 4),
                                               ("matlab", 320, 180, 4)])
def test_fast_traversal_never_loses_a_hit(scene, w, h, bounces):
    sc = O.OracleScene.from_dir(GOLDEN / scene)
    nodes = S.build_fast_bvh(*_buffers(scene))
    try:
        O.set_fast_bvh(None)
        ref, raux, rc = sc.render(w, h, 0, bounces, want_primary=True, want_counters=True)
        O.set_fast_bvh(nodes)
        img, faux, fc = sc.render(w, h, 0, bounces, want_primary=True, want_counters=True)
    finally:
        O.set_fast_bvh(None)
    tr, tf = raux[..., 1].view(np.float32), faux[..., 1].view(np.float32)
    hit_r, hit_f = raux[..., 2] == 1, faux[..., 2] == 1
    assert not np.any(hit_r & ~hit_f)                    # a first hit the reference finds is never missed
    both = hit_r & hit_f
    assert not np.any(tf[both] > tr[both] + 2e-6 * np.maximum(1.0, tr[both]))   # and never replaced by a farther one
    nearer = int(np.sum(both & (tf < tr - 2e-6 * np.maximum(1.0, tr))) + np.sum(hit_f & ~hit_r))
    same = float(np.mean(raux[..., 0] == faux[..., 0]))
    print(f"{scene}: same first primitive in {same:.5f} of the pixels, {nearer} pixels where the reference misses a nearer hit; "
          f"node visits per ray {rc['nodes'] / rc['rays']:.2f} -> {fc['nodes'] / fc['rays']:.2f}")
    assert same > 0.95
    assert fc["nodes"] < rc["nodes"] or scene == "debug"
