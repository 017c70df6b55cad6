#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counter list>" ...   (one rocprofv3 --pmc pass per counter group)
# PMC_CMD="python3 <script> args" replaces the default workload (config 2, one 1.68e7-mutation launch)
set -e
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
i=0
for grp in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i
  echo "pass $i: $grp" >> $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_progress.txt
  # a counter group the hardware cannot collect in one pass aborts rocprofv3 and can leave the child hanging: bound it
  timeout -k 10 180 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -- ${PMC_CMD:-python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --spp 64 --no-cpu-baseline --no-quality} > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "k_mutate" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc: print("%-28s per-launch %.6g  (launches %d)" % (k, acc[k] / n[k], n[k]))
PY
