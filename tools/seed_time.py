"""Diagnostic: wall time of drmlt_seed (bootstrap + seed selection + replay) for a bench configuration and chain count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import bench
pkg = g.load_package()
name, chains = sys.argv[1], int(sys.argv[2])
conf = bench.CONFIGS[name]
sd = bench.build_scene(pkg, conf, conf["res"])
cfg = pkg.abi.make_config(work_units=chains, luminance_samples=100000, direct_samples=-1, sample_count=conf["spp"], **conf["cfg"])
ctx = pkg.Context(cfg, sd)
t = time.perf_counter()
b = ctx.seed_pool(0x5EED, 0, chains)
print("config %s, %d chains: seed %.2f s (b = %.4g)" % (name, chains, time.perf_counter() - t, b))
