#!/bin/bash
# usage (on the GPU box): tools/sweep_soup1m.sh [config]   -- k_mutate_v5's trace_yield x mh_batch on a BVH scene, two steps each
cfg=${1:-soup1m}
run() { env "$@" python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-quality 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['roofline']['frac'])"; }
for ty in 4 8 12 16; do for mb in 4 8 16; do run DRMLT_TRACE_YIELD=$ty DRMLT_MH_BATCH=$mb; done; done
