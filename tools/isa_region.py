"""Static instruction count of one kernel restricted to a source-line range of one file (asm from -S -gline-tables-only):
python tools/isa_region.py /tmp/k.s <mangled kernel> <file> <first line> <last line>"""
import collections
import re
import sys

path, sym, fname, lo, hi = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
lines = open(path).read().split('\n')
st = next(i for i, l in enumerate(lines) if l.startswith(sym + ':'))
en = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split('/')[-1]
cur, kinds, n = None, collections.Counter(), 0
for l in lines[st:en]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(m.group(1)), int(m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
        continue
    if cur and cur[0] == fname and lo <= cur[1] <= hi:
        n += 1
        op = t.split()[0]
        kinds['_'.join(op.split('_')[:2])] += 1
print(n, 'instructions;', kinds.most_common(30))
