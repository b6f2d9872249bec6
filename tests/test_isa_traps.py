"""Build-time guard for a hardware trap of gfx950 that two rounds ran into: a buffer store of more than 64
bits whose scalar offset sits in an SGPR fetches its data registers late; under memory back-pressure a later LDS read or
load that lands in the same registers corrupts what is stored (about 1e-4 of the pixels, different ones every run).  The
compiler MERGES neighbouring narrow stores, so the source cannot be trusted to show it: the generated ISA of the chain
kernels is scanned instead (hipcc cross-compiles without a GPU)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.parametrize("src", ["sgm_aggregate.hip", "sgm_cost.hip"])
def test_no_wide_buffer_store_with_sgpr_offset(tmp_path, src):
    from wass_amd import build
    out = tmp_path / (src + ".s")
    flags = [f for f in build.FLAGS if f != "-fPIC"]
    subprocess.check_call([HIPCC, *flags, "-S", "--cuda-device-only", "-c", os.path.join(ROOT, "wass_amd", "csrc", src), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    asm = out.read_text()
    assert "buffer_store_dword" in asm or "global_store_dword" in asm          # the scan looks at the right thing
    bad = re.findall(r"buffer_store_dwordx[34]\s+[^\n]*\],\s*s\d+[^\n]*", asm)
    assert not bad, f"{len(bad)} wide buffer stores with an SGPR offset, e.g. {bad[0].strip()}"


# The chain kernels are software-pipelined by hand and depend on the compiler leaving loads, waits and registers where the source
# put them; five compiler behaviours and one MISCOMPILE (the per-lane load of a minima record sunk past a loop exit in
# k_pair<2, 8, 1>, wrong S in short diagonal chains; held in place by the fence in Rec::load, NOTES/traps.md) were found on the
# toolchain below.  A different hipcc is therefore a correctness event, not a routine bump: this test names the versions the
# library has been validated on (GPU suite + scripts/selftest.py + tests/test_sgm_gpu.py::test_short_chains_every_instance) and
# fails on any other until someone has run those on a GPU box and added the new version string here.
VALIDATED_HIPCC = (
    "HIP version: 7.2.26015-fc0010cf6a",      # ROCm 7.2.0 image of rounds 1-5 (AMD clang 22.0.0git)
)


def test_hipcc_is_a_version_the_chain_kernels_were_validated_on():
    out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
    first = out.strip().splitlines()[0].strip() if out.strip() else ""
    assert first in VALIDATED_HIPCC, (
        f"hipcc reports {first!r}; the aggregation kernels were validated on {VALIDATED_HIPCC}.  Run `pytest -m gpu tests/test_sgm_gpu.py "
        "tests/test_fullsize_gpu.py` and `python scripts/selftest.py` on a GPU box with this compiler, then add the version string.")
