"""Host-side helpers of the C++ drop-in (wass_amd/host/hostio.hpp) that must reproduce what the reference's iostreams print.

fmt_g6 writes plane_refinement_inliers.xyz (wass_stereo.cpp:2077-2085: `ofs << x << " " << y << " " << z`, the stream's
default format = printf("%g")); it is a fast path in front of std::to_chars and has to give the same characters."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fmt_check():
    src = os.path.join(HERE, "native", "fmt_check.cpp")
    hdr = os.path.join(HERE, "..", "wass_amd", "host", "hostio.hpp")
    out_dir = os.path.join(HERE, "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "fmt_check")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", src, "-o", exe, "-lz"])
    return exe


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fast_g_format_equals_printf(fmt_check, seed):
    """400 000 rounds x 9 values: random magnitudes 1e-7 .. 1e8, values on and one ulp beside the sixth-digit rounding
    boundaries, short decimals, integers, dyadic fractions, and the range limits of %g's fixed notation."""
    r = subprocess.run([fmt_check, "400000", str(seed)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("0 mismatches"), r.stdout[-2000:]
