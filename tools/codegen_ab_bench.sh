#!/bin/bash
# Per bench config: mutations/s of the default library (A) and of variant builds drmlt-mitsuba_amd/variants/lib<X>.so
# (see codegen_ab.sh for how to build one). On the GPU box:  VARIANTS="A W" CONFIGS="2 3 5 bdpt" bash tools/codegen_ab_bench.sh
for v in ${VARIANTS:-A S}; do
  if [ $v = A ]; then unset DRMLT_LIBRARY; else export DRMLT_LIBRARY=$GRAFT_REPO_ROOT/drmlt-mitsuba_amd/variants/lib$v.so; fi
  for c in ${CONFIGS:-2 3 5 bdpt soup50k}; do
    r=$(python bench.py --config $c --no-cpu-baseline --no-quality 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e' % d['value'])")
    echo "$v $c $r"
  done
done
