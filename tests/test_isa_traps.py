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
