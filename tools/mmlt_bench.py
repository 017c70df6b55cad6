"""Throughput of the technique=mmlt chain kernel on BASELINE config 5 (glass caustic, orbital, fixEmitterPath)."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
abi = pkg.abi
res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
per_chain = int(sys.argv[3]) if len(sys.argv) > 3 else 256
amap = int(sys.argv[4]) if len(sys.argv) > 4 else 1
sd = pkg.scenes.SCENES[sys.argv[5] if len(sys.argv) > 5 else 'caustic_c5'](res)
cfg = abi.make_config(technique='mmlt', type='orbital', max_depth=6, direct_samples=-1, fix_emitter_path=1,
                      acceptance_map=amap, work_units=chains, sample_count=1, luminance_samples=1000, p_large=float(os.environ.get('PLARGE', 0.3)))
ctx = pkg.Context(cfg, sd)
t0 = time.time(); b = ctx.seed(0x5EED); t_seed = time.time() - t0
ctx.run(chains * 16)
ctx.kernel_time(reset=True)
t0 = time.time(); ctx.run(chains * per_chain); dt = time.time() - t0
ms, n = ctx.kernel_time()
st = ctx.stats()
print('mmlt config5 res=%d chains=%d: b=%.4f seed %.2fs; %.3e mutations/s wall, kernel %.2f ms x %d; evals/mut %.2f rays/mut %.2f acc1 %.3f acc2 %.3f' % (
    res, chains, b, t_seed, chains * per_chain / dt, ms, n, st.path_evals / st.mutations, st.rays / st.mutations,
    st.first_acc / st.first_base, st.second_acc / max(st.second_base, 1)))
