"""The divide-free decode of a 16-bit box coordinate used by the HIP kernel equals the IEEE
quotient float(q)/65535.0 for every possible q (the reference shader divides, compute.glsl:244-251)."""

import numpy as np


def test_markstein_quotient_is_exact_for_all_codes():
    f32 = np.float32
    q = np.arange(65536, dtype=np.uint32).astype(f32)
    c = f32(1.0) / f32(65535.0)
    q0 = (q * c).astype(f32)
    # fma(-q0, 65535, q) and fma(r, c, q0): emulate the single rounding in float64 (products of two
    # float32 values are exact in float64, and the sums here are exact or far from a double-rounding tie)
    r = (-(q0.astype(np.float64)) * 65535.0 + q.astype(np.float64)).astype(f32)
    f = (r.astype(np.float64) * np.float64(c) + q0.astype(np.float64)).astype(f32)
    ref = (q / f32(65535.0)).astype(f32)
    assert np.array_equal(f.view(np.uint32), ref.view(np.uint32))
    # the naive product is NOT enough, which is why the residual step exists
    assert (q0 != ref).sum() > 0
