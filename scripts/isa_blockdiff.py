#!/usr/bin/env python3
"""scripts/isa_blockdiff.py A.s B.s <mangled-name-prefix> -- two ISA listings of one kernel (hipcc -S --cuda-device-only), compared block by block:
same control-flow skeleton?  which blocks differ in their instruction mix?  where are the memory operations (LDS, buffer, global) of each block?
Written for the round-3 miscompile of k_pair<2, 8, 1> (NOTES/traps.md): registers differ throughout between two builds, the block structure
and the memory operations per block do not -- that is what to look at.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-rdc --cuda-device-only -S sgm_aggregate.hip -o a.s
    python scripts/isa_blockdiff.py a.s b.s _ZN4wass6k_pairILi2ELi8ELi1EEE [--mem]
"""
import re
import sys
from collections import Counter


def kernel(path, prefix):
    s = open(path).read().split("\n")
    start = [i for i, l in enumerate(s) if l.startswith(prefix) and ":" in l][0]
    end = [i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end")][0]
    blk, order, d = "entry", ["entry"], {"entry": []}
    for l in s[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; %(bb\.\d+):", l)
        if m:
            blk = m.group(1)
            order.append(blk)
            d[blk] = []
            continue
        t = l.strip()
        if l.startswith("\t") and t and not t.startswith((".", ";")):
            d[blk].append(t)
    return order, d


def main():
    a, b, prefix = sys.argv[1:4]
    (oa, da), (ob, db) = kernel(a, prefix), kernel(b, prefix)
    print("same block skeleton:", oa == ob, "| blocks:", len(oa), len(ob), "| instructions:", sum(map(len, da.values())), sum(map(len, db.values())))
    mem = ("ds_", "buffer_", "global_", "scratch_")
    for blk in oa:
        if blk not in db:
            print(blk, "only in A")
            continue
        ca, cb = Counter(x.split()[0] for x in da[blk]), Counter(x.split()[0] for x in db[blk])
        if ca != cb:
            print(blk, len(da[blk]), len(db[blk]), {k: (ca[k], cb[k]) for k in sorted(set(ca) | set(cb)) if ca[k] != cb[k]})
            if "--mem" in sys.argv:
                for name, dd in (("A", da), ("B", db)):
                    print("   ", name, [re.sub(r"\s+", " ", x)[:60] for x in dd[blk] if x.startswith(mem)])


if __name__ == "__main__":
    main()
