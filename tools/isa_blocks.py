#!/usr/bin/env python
"""Instruction classes per basic block of one kernel in a hipcc -S listing:  python tools/isa_blocks.py file.s kernel_substring"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if sys.argv[2] in l and l.rstrip().split(';')[0].strip().endswith(':') and not l.startswith('\t')][0]
end = [i for i, l in enumerate(lines) if i > start and '.Lfunc_end' in l and l.strip().endswith(':')][0]
blk, stats, order = 'entry', {}, []
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blk = m.group(1)
    if blk not in stats:
        stats[blk] = dict(valu=0, mfma=0, ds_r=0, ds_w=0, salu=0, vmem=0, scratch=0, wait=0, br='')
        order.append(blk)
    t = l.strip().split(' ')[0].split('\t')[0] if l.startswith('\t') else ''
    if not t or t.startswith('.') or t.startswith(';'):
        continue
    s = stats[blk]
    if t.startswith('v_mfma'):
        s['mfma'] += 1
    elif t.startswith('v_'):
        s['valu'] += 1
    elif t.startswith(('ds_read', 'ds_bpermute', 'ds_swizzle')):
        s['ds_r'] += 1
    elif t.startswith('ds_write'):
        s['ds_w'] += 1
    elif t.startswith('scratch'):
        s['scratch'] += 1
    elif t.startswith(('s_waitcnt', 's_nop', 's_barrier')):
        s['wait'] += 1
    elif t.startswith(('s_cbranch', 's_branch')):
        s['br'] += ' ' + l.strip().split()[-1]
        s['salu'] += 1
    elif t.startswith('s_'):
        s['salu'] += 1
    elif t.startswith(('global', 'buffer')):
        s['vmem'] += 1
for b in order:
    print(b, {k: v for k, v in stats[b].items() if v})
