#!/bin/bash
# usage (on the GPU box): tools/profile_config.sh <round tag, e.g. r02> <config> [<config> ...]
# Per config of bench.py: a rocprofv3 --kernel-trace --stats run of five bench steps without warm-up (its JSON line is kept
# beside the stats: HIP-event and profiler durations of the SAME launches), then one --pmc pass per counter group (tools/pmc_run.sh, 64
# mutations/pixel launches: counters are per mutation),
# then the summary bench.py reads. Outputs land in gpurun_out/; copy what is to be judged into profiles/.
set -e
R=$GRAFT_REPO_ROOT
rt=$1; shift
for cfg in "$@"; do
  spp=64
  case $cfg in
    2) name=c2;;
    3) name=c3; spp=60;;    # 196 608 chains: a whole number of mutations per chain
    2x) name=2x; spp=60;;
    5) name=c5; spp=256;;   # the bidirectional kernels cut a call into a short first launch and the rest (regrouping): a call long
    bdpt) name=bdpt; spp=256;; # enough that the chain state's load / store per launch is amortised as in a render
    soup|soup50k|soup1m) name=$cfg; spp=60;; # 196 608 chains: a whole number of mutations per chain
    *) name=$cfg;;
  esac
  muts=$((512*512*spp))
  cmd="python3 $R/bench.py --config $cfg --steps 1 --warmup 0 --spp $spp --no-cpu-baseline --no-quality"
  echo "== $cfg: kernel trace" >> $R/gpurun_out/profile_progress.txt
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_${rt}_$cfg -- python3 $R/bench.py --config $cfg --steps 5 --warmup 0 --no-cpu-baseline --no-quality > $R/gpurun_out/kt_${rt}_${cfg}_bench.json 2> $R/gpurun_out/kt_${rt}_$cfg.log )
  f=$(find $R/gpurun_out/kt_${rt}_$cfg -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/${rt}_${name}_kernel_stats.csv
  [ -n "$SKIP_PMC" ] && continue
  echo "== $cfg: counters" >> $R/gpurun_out/profile_progress.txt
  PMC_CMD="$cmd" $R/tools/pmc_run.sh ${rt}_$cfg \
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES" \
    "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
    "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" \
    "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" > $R/gpurun_out/pmc_${rt}_$cfg.txt
  python3 $R/tools/pmc_summary.py ${rt}_$cfg $muts $R/gpurun_out/${rt}_${name}_pmc.json "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- $cmd (tools/profile_config.sh; one pass per counter group, FETCH_SIZE and WRITE_SIZE in passes of their own)"
done
