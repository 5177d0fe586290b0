#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (which block is the steady-state
loop, and what rides beside its MFMAs).  usage: isa_blocks.py file.s <kernel-substring> [min_mfma]"""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and pat in l.split(':')[0])
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, cur = [], ['entry', []]
for l in lines[start + 1:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur); cur = [m.group(1), []]
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur[1].append(t.split()[0] + ' ' + ' '.join(t.split()[1:]))
blocks.append(cur)
def cls(op):
    if 'mfma' in op: return 'mfma'
    if op.startswith('ds_read') or op.startswith('ds_load'): return 'ds_rd'
    if op.startswith('ds_write') or op.startswith('ds_store'): return 'ds_wr'
    if op.startswith('global_load') or op.startswith('buffer_load'): return 'gld'
    if op.startswith('global_store'): return 'gst'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_barrier'): return 'bar'
    if op.startswith('v_accvgpr'): return 'acc_mv'
    if op.startswith('scratch'): return 'scratch'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'): return 'salu'
    return 'other'
for name, ins in blocks:
    c = collections.Counter(cls(i.split()[0]) for i in ins)
    if c['mfma'] >= min_mfma:
        br = [i for i in ins if i.startswith('s_cbranch') or i.startswith('s_branch')]
        print(name, len(ins), dict(c), br[-2:])
        if '-v' in sys.argv:
            vc = collections.Counter(i.split()[0] for i in ins if cls(i.split()[0]) in ('valu', 'salu', 'nop', 'wait'))
            print('   ', dict(vc))
