for v in ${VARIANTS:-A S}; do
  if [ $v = A ]; then unset DRMLT_LIBRARY; else export DRMLT_LIBRARY=$GRAFT_REPO_ROOT/drmlt-mitsuba_amd/variants/lib$v.so; fi
  for c in ${CONFIGS:-2 3 5 bdpt soup50k}; do
    r=$(python bench.py --config $c --no-cpu-baseline --no-quality 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e' % d['value'])")
    echo "$v $c $r"
  done
done
