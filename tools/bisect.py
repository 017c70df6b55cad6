import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
abi, scenes = pkg.abi, pkg.scenes
sd = scenes.cornell_c1(64)
t = sys.argv[1] if len(sys.argv) > 1 else "orbital"
pl = float(os.environ.get("PLARGE", "0.3"))
cfg = abi.make_config(type=t, max_depth=8, direct_samples=-1, work_units=64, luminance_samples=2000, sample_count=1, p_large=pl)
ctx = pkg.Context(cfg, sd)
ctx.seed(1)
ctx.run(64 * int(sys.argv[2]) if len(sys.argv) > 2 else 64)
print("OK", os.environ.get("DRMLT_DEBUG"), pl, sys.argv[1:], ctx.stats().ratios())
