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
    """The marching Gaussian computes a/b as q0=a*y, two fma residual corrections, y=RN(1/b) (the tail of
    the IEEE division expansion without operand scaling).  On the device, for EVERY float a in [0,256]
    (1.13e9 values per divisor) it must equal the IEEE quotient, except where the residual underflows:
    dividends below 2^-100.  The host only selects that kernel when every non-zero dividend is >= 2^-97
    (min tap >= 2^-48), see canny_capi.hip::dev_gaussian."""
    divisors = set()
    for sigma in np.arange(0.15, 2.67, 0.05):
        w = _weights_for(float(np.float32(sigma)))
        divisors.add(w[-1])                                  # full window
    for sigma in (0.5, 1.0, 1.4, 2.0):
        divisors.update(_weights_for(sigma))                  # every border weight of the BASELINE sigmas
    divisors.update([1.0, 0.5, 0.99999994, 1.0000001, 0.33333334, 0.7865707, 0.2, 0.05])
    limit = 2.0 ** -100
    with hip.Context(0) as c:
        res = {d: c.selftest_div(d) for d in sorted(divisors)}
    wrong = {d: (n, worst) for d, (n, worst) in res.items() if n and worst >= limit}
    assert not wrong, f"reciprocal division differs from IEEE above 2^-100 for divisors {wrong}"
    assert len(res) > 60


def test_single_fma_division_table_is_exact_for_every_dividend(hip):
    """Interior Gaussian waves divide by the full-window weight S with ONE instruction, fma(a, c, a), when
    (S, c) is in the kernels' built-in table.  Every entry must equal the IEEE quotient for every float a in
    [0, 256] -- all 1.13e9 of them, no exceptions (not even the tiny dividends the 5-op form misses)."""
    table = hip.fma_div_table()
    assert len(table) >= 7 and (1.0, 0.0) in table
    with hip.Context(0) as c:
        for s, cc in table:
            bad, worst = c.selftest_div_fma(s, cc)
            assert bad == 0, f"fma(a, {cc!r}, a) != a / {s!r} for {bad} dividends (largest {worst!r})"
        # and the check itself can fail: a wrong constant is caught
        bad, _ = c.selftest_div_fma(1.0000001192092896, -1.1920928955078125e-07)
        assert bad > 0


@pytest.mark.parametrize("fma_div", [0, 1])
def test_gaussian_with_and_without_fma_division(hip, fma_div):
    """Same bits from both division forms, for sigmas whose full-window weight is 1, 1-2^-24 and 1+2^-23."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(300, 1000), dtype=np.uint8)
    with hip.Context(0) as c:
        c.set_option("gaussian_fma_div", fma_div)
        try:
            for sigma in (0.5, 1.0, 1.2, 1.4, 2.0, 2.5):
                assert np.array_equal(c.gaussian(img, sigma), oracle.gaussian(img, sigma)), sigma
        finally:
            c.set_option("gaussian_fma_div", 1)


@pytest.mark.parametrize("sigma", [0.05, 0.08, 0.1, 0.12, 0.13, 0.14, 0.15, 0.2])
def test_tiny_sigma_gaussian_is_still_bit_exact(hip, sigma):
    """Taps as small as exp(-1/(2 sigma^2)): denormal and underflowing weights; below the 2^-48 tap gate
    the host must fall back to the IEEE-divide kernels on its own."""
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(70, 300), dtype=np.uint8)
    img[rng.random(img.shape) < 0.5] = 0
    with hip.Context(0) as c:
        assert np.array_equal(c.gaussian(img, sigma), oracle.gaussian(img, sigma))
