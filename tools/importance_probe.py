"""Chains under a SYNTHETIC importance map against plain chains and a path-traced reference (device only): any positive map must
leave the expectation alone (SplatList::normalize divides by it, develop multiplies it back).
  python tools/importance_probe.py --scene cornell_c2 --map ramp"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402
LUMW = np.array([0.212671, 0.715160, 0.072169])
def rel_mse(img, ref):
    li, lr = img @ LUMW, ref @ LUMW
    return float(np.mean((li - lr) ** 2 / (lr ** 2 + 1e-2 * lr.mean() ** 2)))
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell_c2"); ap.add_argument("--map", default="ramp"); ap.add_argument("--n", type=int, default=32)
ap.add_argument("--spp", type=int, default=1024); ap.add_argument("--chains", type=int, default=4096); ap.add_argument("--type", default="orbital"); ap.add_argument("--p-large", type=float, default=0.3); ap.add_argument("--algo", type=int, default=0); ap.add_argument("--kelemen-weights", type=int, default=1)
a = ap.parse_args()
pkg = g.load_package(); abi = pkg.abi
res = 64
sd = pkg.scenes.SCENES[a.scene](res=res)
cfg = abi.make_config(technique="path", type=a.type, max_depth=6, rr_depth=5, direct_samples=-1, work_units=a.chains, sample_count=a.spp, luminance_samples=100000, p_large=a.p_large, algo=a.algo, kelemen_style_weights=a.kelemen_weights)
xs = (np.arange(res) + 0.5) / res
imp = {"ramp": np.tile(0.02 + xs, (res, 1)), "const": np.full((res, res), 0.37), "steps": np.tile(np.where(xs < 0.5, 0.01, 1.0), (res, 1))}[a.map].astype(np.float32)
rc = pkg.Context(cfg, sd); ref = 0.5 * (rc.render_pt(65536, seed=11).astype(np.float64) + rc.render_pt(65536, seed=22).astype(np.float64)); rc.close()
imgs = []
for i in range(a.n):
    c = pkg.Context(cfg, sd); c.set_importance_map(imp); c.seed(100 + i); c.run(res * res * a.spp); imgs.append(c.develop().astype(np.float64)); c.close()
imgs = np.array(imgs); m = imgs.mean(0)
noise = np.mean([rel_mse(x - m + ref, ref) for x in imgs]) * a.n / (a.n - 1)
np.set_printoptions(precision=3, suppress=True, linewidth=200)
blk = lambda x: (x @ LUMW).reshape(8, 8, 8, 8).sum((1, 3))
print("algo %d kelemen weights %d" % (a.algo, a.kelemen_weights), end=" "); print("%s map %s type %s chains %d spp %d pLarge %.2f: mean of %d vs reference rel. MSE %.4g (noise of the mean %.4g)" % (a.scene, a.map, a.type, a.chains, a.spp, a.p_large, a.n, rel_mse(m, ref), noise / a.n))
print("column profile mean/reference:", (blk(m).sum(0) / blk(ref).sum(0)))
