"""sponza_synth: a deterministic stand-in for the reference's missing Sponza model.

The reference's headline scene (``models/sponza_full.obj``, about 260 k triangles) is not
distributed with it.  This module generates an atrium of the same scale from closed-form
geometry — tessellated floor and walls with relief, two storeys of colonnades with arches,
hanging cloths, vases — about 262 k triangles at ``detail=1.0``, lit by emissive quads, using the
reference's own scene schema (materials / spheres / quads as in its TOML files, meshes as
triangle arrays).  Every report that uses it says so; a user-supplied OBJ replaces it.

No random numbers: all relief is sums of sines, so the triangle soup is bit-stable.
"""

from __future__ import annotations

import numpy as np


def _grid(p00, du, dv, nu, nv, disp=None):
    """(nu*nv*2, 3, 3) triangles of a parametric patch p00 + u*du + v*dv, optional displacement
    function disp(u, v) -> (n,3) offsets, u and v in [0,1]."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    p = (np.asarray(p00, float)[None, None, :] + uu[..., None] * np.asarray(du, float)[None, None, :]
         + vv[..., None] * np.asarray(dv, float)[None, None, :])
    if disp is not None:
        p = p + disp(uu, vv)
    a, b, c, d = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
    t1 = np.stack([a, b, c], axis=-2).reshape(-1, 3, 3)
    t2 = np.stack([a, c, d], axis=-2).reshape(-1, 3, 3)
    return np.concatenate([t1, t2])


def _surface(fn, nu, nv):
    """Triangles of a parametric surface fn(u, v) -> (..., 3), u and v in [0,1]."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    p = fn(uu, vv)
    a, b, c, d = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
    t1 = np.stack([a, b, c], axis=-2).reshape(-1, 3, 3)
    t2 = np.stack([a, c, d], axis=-2).reshape(-1, 3, 3)
    return np.concatenate([t1, t2])


def _column(cx, cz, y0, y1, radius, seg, rings):
    def fn(u, v):
        ang = 2.0 * np.pi * u
        flute = 1.0 + 0.04 * np.cos(12.0 * ang)          # fluting
        bulge = 1.0 + 0.08 * np.sin(np.pi * v)           # entasis
        r = radius * flute * bulge
        return np.stack([cx + r * np.cos(ang), y0 + (y1 - y0) * v, cz + r * np.sin(ang)], axis=-1)

    return _surface(fn, seg, rings)


def _arch(x0, x1, z, y, tube, seg, around):
    """Half torus spanning x0..x1 at height y in the plane z."""
    cx, R = 0.5 * (x0 + x1), 0.5 * abs(x1 - x0)

    def fn(u, v):
        a = np.pi * u
        b = 2.0 * np.pi * v
        r = R + tube * np.cos(b)
        return np.stack([cx - r * np.cos(a), y + r * np.sin(a), z + tube * np.sin(b)], axis=-1)

    return _surface(fn, seg, around)


def _vase(cx, cy, cz, seg, rings):
    def fn(u, v):
        ang = 2.0 * np.pi * u
        prof = 0.18 + 0.22 * np.sin(np.pi * v) ** 2 + 0.05 * np.sin(3.0 * np.pi * v)
        return np.stack([cx + prof * np.cos(ang), cy + 1.1 * v, cz + prof * np.sin(ang)], axis=-1)

    return _surface(fn, seg, rings)


def _cloth(x0, x1, ytop, ybot, z, n, phase):
    def fn(u, v):
        x = x0 + (x1 - x0) * u
        y = ytop + (ybot - ytop) * v
        sag = 0.35 * np.sin(np.pi * u) * v
        fold = 0.12 * np.sin(9.0 * np.pi * u + phase) * (0.3 + v)
        return np.stack([x, y - sag, z + fold + 0.05 * np.sin(5.0 * np.pi * v + phase)], axis=-1)

    return _surface(fn, n, n)


def _box(c, h):
    c = np.asarray(c, float)
    h = np.asarray(h, float)
    s = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], float)
    v = c + s * h
    f = [(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (3, 7, 6), (3, 6, 2), (0, 4, 7), (0, 7, 3),
         (1, 2, 6), (1, 6, 5)]
    return np.array([[v[a], v[b], v[c2]] for a, b, c2 in f])


def sponza_synth(detail: float = 1.0):
    """Returns (scene_dict, meshes): scene_dict follows the reference's TOML schema (materials,
    spheres, quads); meshes is a list of ((n,3,3) float32 triangles, material index)."""
    d = float(detail)

    def N(n, lo=2):
        return max(lo, int(round(n * np.sqrt(d))))

    materials = [
        {"color": [1.0, 1.0, 1.0], "emission": [1.0, 0.96, 0.9, 12.0]},                # 0 sky light
        {"color": [0.78, 0.74, 0.66], "smoothness": 0.1},                               # 1 stone
        {"color": [0.85, 0.83, 0.8], "smoothness": 0.6},                                # 2 marble floor
        {"color": [0.7, 0.12, 0.1]},                                                     # 3 red cloth
        {"color": [0.12, 0.5, 0.18]},                                                    # 4 green cloth
        {"color": [0.15, 0.2, 0.65]},                                                    # 5 blue cloth
        {"color": [0.8, 0.55, 0.25], "smoothness": 0.7, "metallic": 0.9},               # 6 bronze
        {"color": [1.0, 1.0, 1.0], "smoothness": 1.0, "transmission": 1.0, "ior": 1.5},  # 7 glass
        {"color": [0.9, 0.9, 0.9], "smoothness": 1.0},                                   # 8 polished white
        {"color": [1.0, 0.6, 0.3], "emission": [1.0, 0.6, 0.3, 5.0]},                  # 9 warm lamp
        {"color": [0.62, 0.6, 0.58], "smoothness": 0.3, "metallic": 0.2},               # 10 column stone
    ]
    x0, x1, zw, ytop = -16.0, 12.0, 6.0, 12.0
    L = x1 - x0
    meshes = []

    def relief(amp, fu, fv, axis):
        def f(uu, vv):
            o = np.zeros(uu.shape + (3,))
            o[..., axis] = amp * (np.sin(fu * np.pi * uu) * np.sin(fv * np.pi * vv)
                                  + 0.5 * np.sin(2.3 * fu * np.pi * uu + 1.0) * np.sin(1.7 * fv * np.pi * vv + 2.0))
            return o
        return f

    # floor, ceiling, walls (with relief so that boxes are not axis-aligned slabs)
    meshes.append((_grid([x0, 0.0, -zw], [L, 0, 0], [0, 0, 2 * zw], N(64), N(32), relief(0.015, 40, 20, 1)), 2))
    meshes.append((_grid([x0, ytop, -zw], [0, 0, 2 * zw], [L, 0, 0], N(32), N(64), relief(0.05, 9, 14, 1)), 1))
    meshes.append((_grid([x0, 0.0, -zw], [0, ytop, 0], [L, 0, 0], N(48), N(128), relief(0.04, 24, 56, 2)), 1))
    meshes.append((_grid([x0, 0.0, zw], [L, 0, 0], [0, ytop, 0], N(128), N(48), relief(0.04, 56, 24, 2)), 1))
    meshes.append((_grid([x0, 0.0, -zw], [0, 0, 2 * zw], [0, ytop, 0], N(48), N(48), relief(0.04, 20, 20, 0)), 1))
    meshes.append((_grid([x1, 0.0, -zw], [0, ytop, 0], [0, 0, 2 * zw], N(48), N(48), relief(0.04, 20, 20, 0)), 1))
    # gallery slabs of the upper storey
    for zs in (-1.0, 1.0):
        meshes.append((_grid([x0, 5.6, zs * 3.4], [L, 0, 0], [0, 0, zs * 2.6], N(64), N(8), relief(0.01, 30, 4, 1)), 1))
    # colonnades: 2 rows x 10 columns x 2 storeys, with capitals, and arches between neighbours
    ncol = 10
    xs = np.linspace(x0 + 2.0, x1 - 2.5, ncol)
    for zs in (-3.4, 3.4):
        for storey, (ya, yb, rad) in enumerate(((0.0, 4.6, 0.34), (5.65, 9.6, 0.27))):
            for cx in xs:
                meshes.append((_column(cx, zs, ya, yb, rad, N(48, 6), N(28, 3)), 10))
                meshes.append((_box([cx, yb + 0.12, zs], [rad * 1.5, 0.12, rad * 1.5]), 1))
            for a, b in zip(xs[:-1], xs[1:]):
                meshes.append((_arch(a, b, zs, yb + 0.24, 0.17, N(32, 4), N(16, 3)), 1))
    # balusters along the galleries
    nb = max(4, int(100 * d))
    for zs in (-3.4, 3.4):
        for bx in np.linspace(x0 + 1.0, x1 - 1.0, nb):
            meshes.append((_box([bx, 6.05, zs * 0.995], [0.04, 0.4, 0.04]), 8))
    # hanging cloths
    cloth_mats = (3, 4, 5, 3, 5, 4, 3)
    for i, m in enumerate(cloth_mats):
        xa = x0 + 3.0 + i * 3.4
        meshes.append((_cloth(xa, xa + 2.6, 9.3, 5.9, (-1.0) ** i * 2.9, N(64, 4), 0.7 * i), m))
    # vases on the floor
    for i in range(8):
        meshes.append((_vase(x0 + 4.0 + 3.0 * i, 0.0, (-1.0) ** i * 1.6, N(32, 6), N(32, 4)), 6))

    spheres = [
        {"center": [2.0, 0.8, 0.0], "radius": 0.8, "material": 7},
        {"center": [-3.0, 0.6, 1.2], "radius": 0.6, "material": 8},
        {"center": [-8.0, 0.7, -1.0], "radius": 0.7, "material": 6},
        {"center": [5.0, 0.5, -2.0], "radius": 0.5, "material": 7},
    ]
    quads = [
        {"center": [-2.0, ytop - 0.06, 0.0], "width": 5, "height": 16, "normal": [0.0, -1.0, 0.0], "material": 0},
        {"center": [8.0, ytop - 0.06, 0.0], "width": 4, "height": 4, "normal": [0.0, -1.0, 0.0], "material": 0},
        {"center": [x0 + 0.06, 3.0, 0.0], "width": 3, "height": 3, "normal": [1.0, 0.0, 0.0], "material": 9},
        {"center": [-6.0, 4.5, -zw + 0.08], "width": 2, "height": 1.5, "normal": [0.0, 0.0, 1.0], "material": 9},
        {"center": [1.0, 4.5, zw - 0.08], "width": 2, "height": 1.5, "normal": [0.0, 0.0, -1.0], "material": 9},
    ]
    scene = {"materials": materials, "spheres": spheres, "quads": quads}
    meshes = [(np.ascontiguousarray(m, dtype=np.float32), mat) for m, mat in meshes]
    return scene, meshes


def triangle_count(meshes) -> int:
    return int(sum(m.shape[0] for m, _ in meshes))
