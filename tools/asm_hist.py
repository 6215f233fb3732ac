#!/usr/bin/env python3
"""Histogram the instructions of the hottest loop of a kernel in a hipcc -save-temps .s file.
usage: asm_hist.py file.s mangled_name_substring"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(_Z\w*%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm' % re.escape(key), s, re.S | re.M)
body = m.group(2)
lines = body.split('\n')
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
loop = lines[best[0]:best[1] + 1]
cnt = collections.Counter()
for l in loop:
    l = l.strip()
    if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
        continue
    cnt[l.split()[0]] += 1
tot = sum(cnt.values())
valu = sum(v for k, v in cnt.items() if k.startswith('v_') and 'mfma' not in k)
print(m.group(1)[:60], 'loop lines', len(loop), 'instrs', tot, 'valu', valu, 'mfma', best[2])
print(cnt.most_common(45))
