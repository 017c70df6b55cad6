"""Diagnosis (GPU box): the luminance estimate b = mean f(u) of the bootstrap over many seeds, device against oracle.
  python tools/b_probe.py --scene door_c3 --technique bdpt --seeds 512"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="door_c3"); ap.add_argument("--technique", default="bdpt"); ap.add_argument("--seeds", type=int, default=512)
ap.add_argument("--max-depth", type=int, default=6); ap.add_argument("--precision", type=int, default=64)
a = ap.parse_args()
pkg, ob = g.load_package(), g.load_oracle()
ob.build(native=True)
abi = pkg.abi
sd = pkg.scenes.SCENES[a.scene](res=64)
cfg = abi.make_config(technique=a.technique, type="orbital", max_depth=a.max_depth, rr_depth=5, direct_samples=-1, work_units=4096, sample_count=1024, luminance_samples=100000)
bg, bo = [], []
for i in range(a.seeds):
    c = pkg.Context(cfg, sd); bg.append(c.seed(1000 + i)); c.close()
    o = ob.Oracle(abi, cfg, sd, precision=a.precision, native=True); bo.append(o.seed(501000 + i)); o.close()
    if (i + 1) % 64 == 0:
        g_, o_ = np.array(bg), np.array(bo)
        se = np.hypot(g_.std(ddof=1), o_.std(ddof=1)) / np.sqrt(len(g_))
        print("%d seeds: b device %.6f oracle %.6f  diff %+.3f%% +- %.3f%% (%.1f sigma)" % (i + 1, g_.mean(), o_.mean(), 100 * (g_.mean() - o_.mean()) / o_.mean(), 100 * se / o_.mean(), (g_.mean() - o_.mean()) / se), flush=True)
