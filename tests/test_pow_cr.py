"""The device code's powf (pow_cr, raytracer_amd/csrc/rt_device.hpp) against glibc's powf, which the reference and the
oracle call (ray_tracer.cpp:400 specular term, :47-51 tone curve).

pow_cr rounds a double-precision power to float once, which gives the correctly rounded float in all but vanishing
cases.  glibc's powf is NOT correctly rounded (its documented bound is about 0.52 ulp), so it differs from the correctly
rounded float by one unit in the last place in a few results per thousand, whatever the device does; which powf the
reference ran against is unknown besides (a 2017 glibc had another algorithm).  Bit-equality with "the reference's
powf" is therefore not defined, and the image tests carry a 4-ULP / 1-LSB tolerance for exactly this term.

What this test pins is that the shortcut pow_cr takes for whole-number exponents up to 1024 (a specular hardness as the
exporter writes it) adds nothing to that: square-and-multiply in double, up to 20 multiplications each within 2^-53.
The same loop (same order of operations, IEEE double) runs over a dense sweep of bases and every exponent 1..511; its
disagreements with glibc's powf are counted beside those of the plain double pow rounded to float, and the two counts
must be the same to within a few results."""
import ctypes
import ctypes.util

import numpy as np


def _square_and_multiply(x32, n):
    """(float)r of pow_cr's loop: double b = x, r = 1; for (; n; n >>= 1) { if (n & 1) r *= b; b *= b; }"""
    b = x32.astype(np.float64)
    r = np.ones_like(b)
    while n:
        if n & 1:
            r = r * b
        b = b * b
        n >>= 1
    with np.errstate(under="ignore"):
        return r.astype(np.float32)


def test_whole_number_powers_against_glibc_powf():
    libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    rng = np.random.default_rng(7)
    # bases in (0, 1]: the cosine of the specular term; dense near 1, where large exponents still give normal floats
    x = np.concatenate([rng.uniform(0.0, 1.0, 1500), 1.0 - rng.uniform(0.0, 1.0, 1500) ** 3 * 0.2,
                        np.float32(1.0) - np.arange(64, dtype=np.float32) * np.float32(2.0 ** -24), [1.0, 0.5, 0.25]]).astype(np.float32)
    x = x[x > 0]
    total = differ = worst = differ_generic = differ_between = 0
    for n in range(1, 512):
        got = _square_and_multiply(x, n)
        with np.errstate(under="ignore"):
            generic = np.power(x.astype(np.float64), float(n)).astype(np.float32)      # (float)pow((double)x, (double)n)
        ref = np.array([libm.powf(float(v), float(n)) for v in x], np.float32)
        keep = ref >= np.float32(1.2e-38)                    # normal results (a pixel's specular term below that is zero in RGB8)
        as_int = lambda a: a[keep].view(np.int32).astype(np.int64)
        d = np.abs(as_int(got) - as_int(ref))
        total += int(keep.sum()); differ += int((d != 0).sum()); worst = max(worst, int(d.max()) if d.size else 0)
        differ_generic += int((as_int(generic) != as_int(ref)).sum())
        differ_between += int((as_int(generic) != as_int(got)).sum())
    print(f"whole-number exponents 1..511, {total} results: square-and-multiply differs from glibc powf in {differ} ({differ / total:.2e}), "
          f"worst {worst} ulp; double pow rounded to float differs from glibc powf in {differ_generic} ({differ_generic / total:.2e}); "
          f"the two device forms differ from each other in {differ_between}")
    assert worst <= 1, "a difference of more than one unit in the last place is not a rounding-boundary effect"
    assert differ / total < 1e-2 and differ_generic / total < 1e-2
    assert differ_between <= 20 and abs(differ - differ_generic) <= 20, "the shortcut changes how often the device differs from powf"
