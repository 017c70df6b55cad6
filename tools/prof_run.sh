#!/bin/bash
# usage: tools/prof_run.sh <tag> <python script and args...>   -> gpurun_out/prof_<tag>/ (rocprofv3 --kernel-trace --stats)
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 "$@" > $R/gpurun_out/prof_$tag.log 2>&1
grep -v "rocprofv3\]" $R/gpurun_out/prof_$tag.log | tail -2
head -4 $(ls $R/gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1) | cut -c1-160
