"""Static instruction-count profile of one kernel by source line (compile with -gline-tables-only -S first).

In the lane-state-machine kernels every instruction of the main loop is issued for the wave on (nearly) every
iteration, so code size per source region is a usable proxy for where the issue slots go.
  hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -gline-tables-only -o /tmp/k.s kernels.hip
  python tools/isa_profile.py /tmp/k.s _Z11k_mutate_v3ILi0EEv7DParamsjj [bucket]
"""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 10
lines = open(path).read().split('\n')
files, start = {}, None
for i, l in enumerate(lines):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
    if l.startswith(sym + ':'):
        start = i
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
cur = ('?', 0)
cnt, kinds = collections.Counter(), collections.Counter()
for l in lines[start:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
        continue
    cnt[cur] += 1
    kinds[t.split('_')[0]] += 1
tot = sum(cnt.values())
print('total instrs', tot, dict(kinds.most_common(6)))
agg = collections.Counter()
for (f, ln), c in cnt.items():
    agg[(f, ln // bucket * bucket)] += c
for (f, ln), c in sorted(agg.items(), key=lambda kv: -kv[1])[:50]:
    print('%-18s %5d-%-5d %5d  %4.1f%%' % (f, ln, ln + bucket - 1, c, 100 * c / tot))
