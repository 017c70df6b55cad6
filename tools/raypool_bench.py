"""Quick A/B of the chain kernels on a BVH scene (GPU box): mutations/s and traversal occupancy for k_mutate_v4 / v5.
  python tools/raypool_bench.py [n_tris] [chains] [steps]      (env: DRMLT_KERNEL, DRMLT_MH_BATCH, DRMLT_TRACE_YIELD ...)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
res = 512
sd = pkg.scenes.triangle_soup(n_tris, res)
cfg = pkg.abi.make_config(technique="path", type="orbital", max_depth=8, rr_depth=5, direct_samples=-1, work_units=chains, luminance_samples=100000, sample_count=256)
ctx = pkg.Context(cfg, sd)
ctx.seed_pool(0x5EED, 0, chains)
M = res * res * 256
ctx.run(M)
s0 = ctx.stats()
t = time.perf_counter(); ctx.run(steps * M); dt = time.perf_counter() - t
s1 = ctx.stats()
muts = s1.mutations - s0.mutations
print("kernel=%s tris=%d chains=%d: %.4g mutations/s  (%.1f ms/step)  rays/mut %.2f  nodes/mut %.1f prims/mut %.1f" % (
    os.environ.get("DRMLT_KERNEL", "default"), n_tris, chains, muts / dt, 1e3 * dt / steps, (s1.rays - s0.rays) / muts,
    (s1.bvh_node_visits - s0.bvh_node_visits) / muts, (s1.bvh_prim_tests - s0.bvh_prim_tests) / muts), flush=True)
