#!/bin/bash
# usage (on the GPU box): tools/protocol_sweep.sh <group 1|2|3>  -- the parity protocol (N = 64, permutation test) over scene / technique /
# rule combinations beyond the named configurations; one JSON per run in gpurun_out/sweep_*.json, one summary line each on stdout
run() { name=$1; shift; python tools/parity_protocol.py --n 64 --threads 16 --no-reference --out gpurun_out/sweep_$name.json "$@" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', 'ratio %.3f p %.3f medians %+.1f%%' % (d['two_sample_ratio'], d['two_sample_permutation_p'], 100*d['equal_budget_rel_diff_of_medians']))"; }
case $1 in
1) run glass_path --config c2 --scene glass_sphere --set max_depth=6
   run glass_mmlt --config c5 --scene glass_sphere
   run glass_bdpt --config bdpt --scene glass_sphere --set max_depth=6
   run door_mmlt --config c5 --scene door_c3 --set fix_emitter_path=0;;
2) run door_bdpt --config bdpt --scene door_c3 --set max_depth=6
   run cornell_mira --config c2 --set type=mira
   run cornell_green_mixture --config c2 --set type=green,use_mixture=1
   run cornell_timid --config c2 --set timid_after_large=1;;
3) run cornell_pssmlt --config c2 --set algo=1
   run cornell_mmlt_green --config c5 --scene cornell_c2 --set type=green,fix_emitter_path=0
   run cornell_bdpt_nodirect --config bdpt --set no_direct_sampling=1
   run caustic_mmlt_mira --config c5 --set type=mira;;
esac
case $1 in
4) run door_bdpt_rep --config bdpt --scene door_c3 --set max_depth=6 --seed-offset 7000 --save-means gpurun_out/door_bdpt_rep_means.npz
   run door_mmlt_rep --config c5 --scene door_c3 --set fix_emitter_path=0 --seed-offset 7000;;
esac
case $1 in
5) run door_bdpt_fp32oracle --config bdpt --scene door_c3 --set max_depth=6 --oracle-precision 32 --save-means gpurun_out/door_bdpt_fp32_means.npz
   run door_bdpt_fp32oracle_rep --config bdpt --scene door_c3 --set max_depth=6 --oracle-precision 32 --seed-offset 7000;;
esac
case $1 in
6) run cornell_direct16 --config c2 --set direct_samples=16
   run door_orbital --config c3 --set type=orbital
   run door_mira --config c3 --set type=mira
   run glass_green --config c2 --scene glass_sphere --set type=green,max_depth=6;;
7) run cornell_pssmlt_gauss --config c2 --set algo=1,kelemen_style_mutation=0
   run cornell_pssmlt_noweights --config c2 --set algo=1,kelemen_style_weights=0
   run cornell_short --config c2 --set max_depth=3,rr_depth=2,p_large=0.5
   run glass_pssmlt --config c2 --scene glass_sphere --set algo=1,max_depth=6;;
esac
case $1 in
8) run cornell_direct16 --config c2 --set direct_samples=16
   run cornell_mmlt_direct16 --config c5 --scene cornell_c2 --set direct_samples=16,fix_emitter_path=0
   run cornell_bdpt_mixture --config bdpt --set use_mixture=1;;
esac
case $1 in
9) run cornell_amap --config c2 --set acceptance_map=1
   run caustic_mmlt_amap --config c5 --set acceptance_map=1
   run cornell_bdpt_amap --config bdpt --set acceptance_map=1;;
esac
case $1 in
10) run soup300_path --config c2 --scene triangle_soup:300 --n 32;;
esac
case $1 in
11) run caustic_bdpt_green --config bdpt_c5 --set type=green
    run cornell_bdpt_nolightimage --config bdpt --set no_light_image=1
    run caustic_mmlt_nolightimage --config c5 --set no_light_image=1;;
esac
