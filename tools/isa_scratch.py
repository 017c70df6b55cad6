"""Where a kernel touches scratch memory, by source line (asm from -S -gline-tables-only): python tools/isa_scratch.py /tmp/k.s <mangled name>"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
st = next(i for i, l in enumerate(lines) if l.startswith(sys.argv[2] + ':'))
en = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split('/')[-1]
cur, c = None, collections.Counter()
for l in lines[st:en]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(m.group(1)), int(m.group(2)))
    if 'scratch_' in l:
        c[(cur, l.split()[0])] += 1
print(en - st, 'lines;', sum(c.values()), 'scratch instructions')
for k, v in c.most_common(40):
    print(k, v)
