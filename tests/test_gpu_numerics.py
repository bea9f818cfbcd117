"""Device-side arithmetic shortcuts, verified exhaustively on the GPU against the IEEE operations the
reference uses on the CPU."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _weights_for(sigma):
    """Every divisor the Gaussian kernels can meet for this sigma: the float sum of taps lo..hi in
    ascending order (reference src/utils.cpp:41-46: `count += kernel[center+k]` over in-image taps)."""
    taps = oracle.gaussian_kernel(sigma)
    out = set()
    n = len(taps)
    for lo in range(n):
        s = np.float32(0)
        for hi in range(lo, n):
            s = np.float32(s + taps[hi])
            if lo <= n // 2 <= hi:          # the centre tap is always in the image
                out.add(float(s))
    return sorted(out)


def test_reciprocal_division_matches_ieee_for_every_dividend(hip):
    """div_by(a, b, RN(1/b)) == a / b for EVERY float a in [0, 256] (1.13e9 values per divisor), for the
    full-window weight of a sigma sweep and for every border weight of the benchmark sigmas."""
    divisors = set()
    for sigma in np.arange(0.2, 2.67, 0.05):
        w = _weights_for(float(np.float32(sigma)))
        divisors.add(w[-1])                                  # full window
    for sigma in (0.5, 1.0, 1.4, 2.0):
        divisors.update(_weights_for(sigma))
    divisors.update([1.0, 0.5, 0.99999994, 1.0000001, 0.33333334, 0.7865707])
    with hip.Context(0) as c:
        bad = {d: c.selftest_div(d) for d in sorted(divisors)}
    wrong = {d: n for d, n in bad.items() if n}
    assert not wrong, f"reciprocal division differs from IEEE for divisors {wrong}"
    assert len(bad) > 60
