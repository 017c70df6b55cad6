#!/bin/bash
# A/B of library builds that differ in code-generation flags (on the GPU box): build them first, e.g.
#   cd drmlt-mitsuba_amd/csrc && make -B OUT=../variants/libD.so DEVFLAGS="<the Makefile's DEVFLAGS> -mllvm -amdgpu-sched-strategy=max-ilp"
# (drmlt-mitsuba_amd/variants/ is git-ignored; DRMLT_LIBRARY makes the binding load another build of the same ABI).
# Round 2: -amdgpu-early-ifcvt, -amdgpu-sched-strategy=max-ilp and raised SimplifyCFG phi-folding thresholds all stay within
# +-1 % of the default build on config 2, config 3 and the soup.
for v in ${VARIANTS:-A B D E}; do
  if [ $v = A ]; then unset DRMLT_LIBRARY; else export DRMLT_LIBRARY=$GRAFT_REPO_ROOT/drmlt-mitsuba_amd/variants/lib$v.so; fi
  for sc in "cornell_c2 orbital" "door_c3 green" "triangle_soup orbital"; do set -- $sc
    r=$(SCENE=$1 TYPE=$2 SPP=1280 REPS=3 python tools/perf_ab.py 0 2>/dev/null | grep variant | cut -c1-60)
    echo "$v $1 $r"
  done
done
