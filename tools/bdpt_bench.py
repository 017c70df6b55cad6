"""Throughput of the technique=bdpt chain kernel (Cornell config-2 scene, orbital); DIRECT=0/1 selects directSampling."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
abi = pkg.abi
res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
per_chain = int(sys.argv[3]) if len(sys.argv) > 3 else 64
scene = sys.argv[4] if len(sys.argv) > 4 else 'cornell_c2'
sd = pkg.scenes.SCENES[scene](res)
cfg = abi.make_config(technique='bdpt', type='orbital', max_depth=8, rr_depth=5, direct_samples=-1, no_direct_sampling=0 if os.environ.get('DIRECT', '1') == '1' else 1,
                      work_units=chains, sample_count=1, luminance_samples=100000)
ctx = pkg.Context(cfg, sd)
t0 = time.time(); b = ctx.seed(0x5EED); t_seed = time.time() - t0
ctx.run(chains * 8)
ctx.kernel_time(reset=True)
t0 = time.time(); ctx.run(chains * per_chain); dt = time.time() - t0
ms, n = ctx.kernel_time()
st = ctx.stats()
print('bdpt %s res=%d chains=%d: b=%.4f seed %.2fs; %.3e mutations/s wall, kernel %.2f ms x %d; rays/mut %.2f acc1 %.3f acc2 %.3f' % (
    scene, res, chains, b, t_seed, chains * per_chain / dt, ms, n, st.rays / st.mutations,
    st.first_acc / st.first_base, st.second_acc / max(st.second_base, 1)))
