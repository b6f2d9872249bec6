"""csrc/fmt_g6.h -- the number formatter the GPU writes plane_refinement_inliers.xyz with (wass_stereo.cpp:2077-2085: "x y z" per line through a
default std::ofstream, i.e. printf("%g")) -- built for the host and compared with Python's '%g' (correctly rounded, the same characters as glibc's
printf) on the values where a formatter goes wrong: exact ties of dyadic rationals, the neighbours of every power of ten and of every kind of
six-digit midpoint, carries into the next decade, the switch between fixed and scientific notation.  No GPU involved: the same source compiles
for gfx950 (tests/test_cli.py and tests/test_batch_driver.py compare the files the device wrote with the host-formatted ones byte for byte)."""
import ctypes as C
import math
import random

import pytest

from wass_amd import _lib


@pytest.fixture(scope="module")
def fmt():
    lib = _lib.load()
    buf = C.create_string_buffer(32)

    def g(v):
        n = lib.wass_format_g6(C.c_double(v), buf)
        return None if n < 0 else buf.raw[:n].decode()
    return g


def _in_domain(v):
    a = abs(v)
    return v == v and a != math.inf and (a == 0 or 1e-22 <= a < 1e6)


def _check(fmt, values):
    for v in values:
        r = fmt(v)
        if r is None:
            assert not _in_domain(v), f"{v!r} is inside the domain and was refused"
        else:
            assert r == "%g" % v, f"{v!r}: device form {r!r}, printf {'%g' % v!r}"


def test_known_answers(fmt):
    assert fmt(0.0) == "0" and fmt(-0.0) == "-0"
    assert fmt(1.015625) == "1.01562"          # an exact tie (65/64): half to even, down
    assert fmt(1.046875) == "1.04688"          # 67/64: half to even, up
    assert fmt(100000.5) == "100000" and fmt(100001.5) == "100002"
    assert fmt(999999.5) == "1e+06"            # the rounding carries into scientific notation
    assert fmt(0.0001) == "0.0001" and fmt(0.00001) == "1e-05" and fmt(0.000099999951) == "0.0001"
    assert fmt(-123.456789) == "-123.457" and fmt(2.5) == "2.5" and fmt(100000.0) == "100000"
    for v in (float("inf"), float("nan"), 1e6, 1e300, 1e-23, 5e-324):
        assert fmt(v) is None                  # the caller falls back to the host's formatter


def test_random_and_adversarial_values(fmt):
    rnd = random.Random(20260930)
    _check(fmt, (rnd.choice((-1, 1)) * 10 ** rnd.uniform(-22, 6) for _ in range(60000)))
    _check(fmt, (rnd.uniform(-200, 200) for _ in range(60000)))            # what camera-frame coordinates look like
    for n in range(1, 30):                                                    # dyadic rationals: where exact ties live
        _check(fmt, (s * rnd.randrange(1, 1 << min(n + 8, 40)) / (1 << n) for _ in range(600) for s in (1, -1)))
    for e in range(-22, 6):                                                   # both sides of every power of ten
        for d in range(-3, 4):
            v = 10.0 ** e
            for _ in range(abs(d)):
                v = math.nextafter(v, math.inf if d > 0 else 0.0)
            _check(fmt, [v, -v])
    vals = []
    for _ in range(40000):                                                    # the neighbours of six-digit midpoints
        v = (rnd.randrange(100000, 1000000) + 0.5) * 10.0 ** rnd.randrange(-27, 1)
        for d in (-2, -1, 0, 1, 2):
            w = v
            for _ in range(abs(d)):
                w = math.nextafter(w, math.inf if d > 0 else 0.0)
            vals.append(w)
    _check(fmt, vals)
