#!/usr/bin/env python3
"""Print the instruction-class sequence of the MFMA-heaviest loop of a kernel.  usage: asm_seq.py file.s key"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(_Z\w*%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm' % re.escape(sys.argv[2]), s, re.S | re.M)
lines = m.group(2).split('\n')
labels = {}
for i, l in enumerate(lines):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = i
best = None
for i, l in enumerate(lines):
    mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        span = (labels[mm.group(1)], i)
        nm = sum('mfma' in x for x in lines[span[0]:span[1]])
        if best is None or nm > best[2] or (nm == best[2] and span[1] - span[0] < best[1] - best[0]):
            best = (span[0], span[1], nm)
seq = []
for l in lines[best[0]:best[1] + 1]:
    l = l.strip()
    if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
        continue
    op = l.split()[0]
    t = ('M' if 'mfma' in op else 'e' if op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_sqrt')) else 'v' if op.startswith('v_')
         else 'r' if op.startswith('ds_read') else 'w' if op.startswith('ds_write') else 'G' if op.startswith('global_load')
         else '|' if op.startswith('s_waitcnt') else 'B' if op.startswith('s_barrier') else 'n' if op.startswith('s_nop') else 's')
    seq.append(t)
print(''.join(seq))
