"""Instruction mix of every loop of one kernel in a .s file (see scripts/flow.py for how to produce it):
    python scripts/loopstat.py /tmp/agg.s k_rowsweepILi2E"""
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i,l in enumerate(src) if l.startswith('_ZN') and key in l.split(':')[0])
end = next(i for i in range(start, len(src)) if '.Lfunc_end' in src[i])
body = src[start:end]
labels = {}
for i,l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i,l in enumerate(body):
    m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'\s+s_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
def classify(op):
    if op.startswith('v_pk_'): return 'v_pk'
    if op.startswith('v_readlane') or op.startswith('v_readfirstlane') or op.startswith('v_writelane'): return 'lane'
    if op.startswith('v_'): return 'valu_other'
    if op.startswith('s_nop'): return 's_nop'
    if op.startswith('s_waitcnt'): return 's_waitcnt'
    if op.startswith('s_'): return 'salu'
    if op.startswith('global_load') or op.startswith('scratch_load') or op.startswith('buffer_load'): return 'vmem_ld'
    if op.startswith('global_store') or op.startswith('scratch_store') or op.startswith('buffer_store'): return 'vmem_st'
    if op.startswith('ds_'): return 'lds'
    return 'other'
print('kernel lines', len(body), 'loops', [(a,b,b-a) for a,b in loops])
for a,b in loops:
    cnt = collections.Counter(); ops = collections.Counter(); nops = 0
    for l in body[a:b+1]:
        m = re.match(r'^\s+([a-z_0-9]+)', l)
        if not m: continue
        op = m.group(1)
        if op == 's_nop':
            mm = re.search(r's_nop\s+(\d+)', l); nops += int(mm.group(1)) + 1
        if op.startswith('v_') and ('quad_perm' in l or 'row_' in l or 'wave_' in l):
            cnt['dpp'] += 1; ops[op+'_dpp'] += 1; continue
        cnt[classify(op)] += 1; ops[op] += 1
    tot = sum(cnt.values())
    if tot < 50: continue
    print('loop', a, b, 'instrs', tot, dict(cnt), 'nop_cycles', nops)
    print('   ', ops.most_common(30))
for l in src[end:end+40]:
    if any(k in l for k in ('num_vgpr','num_agpr','numbered_sgpr','private_seg_size','Occupancy')): print(l.strip().split('.')[-1])
