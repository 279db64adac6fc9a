"""Math.Acos / Math.Atan2 / Math.Sin of the texture maps (Sphere.fs:59-60, Texture.fs:58) are DEFINED here as the correctly
rounded values (csrc/rt_trig.h explains why).  Three independent routes must agree bit for bit:
  * csrc/rt_trig.h, the text the gfx950 kernel compiles (double-double Newton step), built here for the host;
  * the oracle's binary128 route (libquadmath);
  * mpmath at 250 bits, rounded to nearest -- the pin.
The C runtime's own results (what .NET would return on this machine) are within 1 ulp of them and equal for most inputs: that is
the residual distance to the unpinnable reference (DESIGN.md "Exactness")."""
import ctypes as C
import os
import subprocess

import mpmath as mp
import numpy as np
import pytest

import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
ACOS, SIN, ATAN2 = 7, 8, 9  # rt_dev_arith / orc_arith op numbers; +3 = the C runtime's version in the oracle
_dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def trig(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("trig") / "libtrig_cr.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", so,
                           os.path.join(HERE, "c", "trig_cr.cpp")])
    lib = C.CDLL(so)

    def run(op, a, b=None):
        a = np.ascontiguousarray(a, np.float64)
        bb = np.ascontiguousarray(b if b is not None else a, np.float64)
        out = np.zeros_like(a)
        lib.trig_cr(op, len(a), a.ctypes.data_as(_dp), bb.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
        return out

    def sincos_dd(a):
        a = np.ascontiguousarray(a, np.float64)
        s, c = np.zeros((len(a), 2)), np.zeros((len(a), 2))
        lib.trig_sincos_dd(len(a), a.ctypes.data_as(_dp), s.ctypes.data_as(_dp), c.ctypes.data_as(_dp))
        return s, c

    run.sincos_dd = sincos_dd
    return run


def _same(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint64), b[~nb].view(np.uint64))  # signs of zeros included


def _cases():
    rng = np.random.default_rng(2024)
    n = 12000
    x = np.concatenate([rng.uniform(-1, 1, n), 1 - 10.0 ** rng.uniform(-16, -1, 1500), -1 + 10.0 ** rng.uniform(-16, -1, 1500),
                        rng.choice([-1.0, 1.0], 500) * 10.0 ** rng.uniform(-300, -1, 500),
                        [0.0, -0.0, 0.5, -0.5, 1.0, -1.0, 1 - 2.0 ** -53, -1 + 2.0 ** -53, 2.0 ** -1022, np.sqrt(0.5)]])
    a = np.concatenate([rng.uniform(-100, 100, n), rng.uniform(-1e5, 1e5, 3000), rng.choice([-1.0, 1.0], 800) * 10.0 ** rng.uniform(-300, 0, 800),
                        np.arange(1, 1500) * (np.pi / 2), -np.arange(1, 300) * np.pi, np.nextafter(np.arange(1, 300) * np.pi, 0.0),
                        [0.0, -0.0, 2.0 ** 19, 1048575.9, 5e-324]])
    y = np.concatenate([rng.normal(size=n), rng.normal(size=2000) * 10.0 ** rng.uniform(-200, 200, 2000), rng.normal(size=1500) * 1e-12,
                        rng.normal(size=500)])
    xx = np.concatenate([rng.normal(size=n), rng.normal(size=2000) * 10.0 ** rng.uniform(-200, 200, 2000), rng.normal(size=1500),
                         rng.normal(size=500) * 1e-13])
    return x, a, y, xx


def test_three_routes_agree_bit_for_bit(trig):
    mp.mp.prec = 250
    x, a, y, xx = _cases()
    pin = np.array([float(mp.acos(mp.mpf(float(v)))) for v in x])
    assert _same(trig(ACOS, x), pin) and _same(orc.arith(ACOS, x), pin)
    pin = np.array([float(mp.sin(mp.mpf(float(v)))) for v in a])
    pin[a == 0.0] = a[a == 0.0]  # sin(-0) = -0 (C, IEEE 754); mpmath has no signed zero
    assert _same(trig(SIN, a), pin) and _same(orc.arith(SIN, a), pin)
    pin = np.array([float(mp.atan2(mp.mpf(float(u)), mp.mpf(float(v)))) for u, v in zip(y, xx)])
    assert _same(trig(ATAN2, y, xx), pin) and _same(orc.arith(ATAN2, y, xx), pin)


def test_special_values_follow_the_c_runtime(trig):
    """Math.Atan2's zeros, infinities and NaNs are C's; Math.Acos is NaN outside [-1, 1].  Both routes against libm, whose special
    cases are exact by definition."""
    inf, nan = np.inf, np.nan
    ys = np.array([0.0, -0.0, 0.0, -0.0, 0.0, -0.0, 1.0, -1.0, inf, -inf, inf, -inf, 1.0, -1.0, 1.0, -1.0, nan, 1.0, 0.0, -0.0, 3.0, -3.0])
    xs = np.array([1.0, 1.0, -1.0, -1.0, 0.0, -0.0, 0.0, -0.0, inf, inf, -inf, -inf, inf, inf, -inf, -inf, 1.0, nan, -0.0, 0.0, inf, -inf])
    want = np.arctan2(ys, xs)
    assert _same(trig(ATAN2, ys, xs), want) and _same(orc.arith(ATAN2, ys, xs), want)
    xa = np.array([1.0, -1.0, 1.0000000000000002, -1.0000000000000002, 2.0, nan, inf, 0.0, -0.0])
    with np.errstate(invalid="ignore"):
        want = np.arccos(xa)
    assert _same(trig(ACOS, xa), want) and _same(orc.arith(ACOS, xa), want)
    xs = np.array([0.0, -0.0, nan, inf, -inf])
    with np.errstate(invalid="ignore"):
        want = np.sin(xs)
    assert _same(trig(SIN, xs), want) and _same(orc.arith(SIN, xs), want)


def test_the_reference_kats_of_plane_map_inverse_stay_exact():
    """TestSphere.fs:197-204 asserts planeMapInverse with shouldEqual (exact): the correctly rounded functions give those values."""
    for p, uv in (((1, 0, 0), (0.5, 0.5)), ((-1, 0, 0), (0.0, 0.5)), ((0, 1, 0), (0.5, 1.0)), ((0, -1, 0), (0.5, 0.0)), ((0, 0, 1), (0.25, 0.5)),
                  ((0, 0, -1), (0.75, 0.5))):
        assert tuple(orc.plane_map_inverse(1.0, (0.0, 0.0, 0.0), tuple(float(c) for c in p))) == uv


def test_double_double_sine_and_cosine_are_good_to_2_pow_minus_100(trig):
    """The intermediate itself: sin and cos of a double as double-doubles, against mpmath -- also where the argument is within
    2^-50 of a multiple of pi/2 (the reduction is exact, so the tiny result keeps its RELATIVE accuracy)."""
    mp.mp.prec = 400
    rng = np.random.default_rng(7)
    a = np.concatenate([rng.uniform(-4, 4, 3000), rng.uniform(-1e5, 1e5, 1500), np.arange(1, 800) * (np.pi / 2), 10.0 ** rng.uniform(-30, 0, 500)])
    s, c = trig.sincos_dd(a)
    worst = 0.0
    for v, (sh, sl), (ch, cl) in zip(a, s, c):
        ms, mc = mp.sin(mp.mpf(float(v))), mp.cos(mp.mpf(float(v)))
        es = abs((mp.mpf(float(sh)) + mp.mpf(float(sl)) - ms) / ms)  # relative, for the sine
        ec = abs(mp.mpf(float(ch)) + mp.mpf(float(cl)) - mc)         # absolute, for the cosine
        worst = max(worst, float(es), float(ec) if abs(mc) > 0.5 else float(ec / abs(mc)))
    assert worst < 2.0 ** -100, worst


def test_distance_to_the_c_runtime_is_at_most_one_ulp():
    """What .NET's Math.* would return here is the C runtime's value: equal to the correctly rounded one for most inputs and never
    further than 1 ulp (measured on this machine's libm; the reference's own platform cannot be pinned)."""
    x, a, y, xx = _cases()
    for op, args in ((ACOS, (x,)), (SIN, (a,)), (ATAN2, (y, xx))):
        cr, crt = orc.arith(op, *args), orc.arith(op + 3, *args)
        ok = ~np.isnan(cr)
        assert np.all(np.abs(cr[ok] - crt[ok]) <= np.spacing(np.abs(crt[ok])))
        assert np.mean(cr[ok] == crt[ok]) > 0.9
