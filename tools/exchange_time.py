"""Diagnostic: wall time of drmlt_exchange_tiled alone (one rank, RCCL world size 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
abi = pkg.abi
res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sd = pkg.scenes.cornell_c2(res)
ctx = pkg.Context(abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=65536, sample_count=16), sd)
b = ctx.seed_pool(1, 0, 65536)
ctx.comm_init(pkg.comm_unique_id(), 0, 1)
ctx.run(res * res * 16)
for k in range(3):
    ctx.exchange_tiled(b, want_tile=False)
t = time.perf_counter()
for k in range(20):
    ctx.exchange_tiled(b, want_tile=False)
print("exchange_tiled at %dx%d, world 1: %.3f ms per call" % (res, res, (time.perf_counter() - t) / 20 * 1e3))
