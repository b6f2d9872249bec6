"""ISA skeleton of one kernel: labels, branches, s_waitcnt, barriers and memory operations with their line numbers.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -ffp-contract=off -Iinclude -Iwass_amd/csrc wass_amd/csrc/sgm_aggregate.hip -o /tmp/agg.s
    python scripts/flow.py /tmp/agg.s k_pairxILi2ELi8
How the vmcnt(0) waits of DESIGN.md 4.3 were found."""
import re, sys
src=open(sys.argv[1]).read().split('\n')
key=sys.argv[2]
start=next(i for i,l in enumerate(src) if l.startswith('_Z') and key in l.split(':')[0])
end=next(i for i in range(start,len(src)) if '.Lfunc_end' in src[i])
body=src[start:end]
print("lines", len(body))
for i,l in enumerate(body):
    t=l.strip()
    if re.match(r'^\.LBB',t) or t.startswith('s_waitcnt') or t.startswith('s_barrier') or t.startswith('s_cbranch') or t.startswith('s_branch') or t.startswith('buffer_') or t.startswith('global_') or t.startswith('s_load') or t.startswith('s_buffer') or t.startswith('scratch_'):
        print(f"{i:5d} {t[:80]}")
