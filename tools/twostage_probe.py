"""Two-stage MLT against plain MLT on the device (both estimate the same image): N renders each, distance between the two mean
images against their own noise, permutation test (as tools/parity_protocol.py). Device only.
  python tools/twostage_probe.py --scene cornell_c2 --technique path --n 64"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

LUMW = np.array([0.212671, 0.715160, 0.072169])


def rel_mse(img, ref):
    li, lr = img @ LUMW, ref @ LUMW
    return float(np.mean((li - lr) ** 2 / (lr ** 2 + 1e-2 * lr.mean() ** 2)))


ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell_c2"); ap.add_argument("--technique", default="path"); ap.add_argument("--type", default="orbital")
ap.add_argument("--ref-spp", type=int, default=0, help="technique=path: device path-traced reference with this many samples per pixel"); ap.add_argument("--map", default="", help="steps: a synthetic map of contrast 100 (left half 0.01, right half 1) instead of a first-stage render"); ap.add_argument("--algo", type=int, default=0, help="1: pssmlt"); ap.add_argument("--n", type=int, default=64); ap.add_argument("--res", type=int, default=64); ap.add_argument("--spp", type=int, default=1024)
a = ap.parse_args()
pkg = g.load_package(); abi = pkg.abi
sd = pkg.scenes.SCENES[a.scene](res=a.res)
cfg = abi.make_config(technique=a.technique, type=a.type, algo=a.algo, max_depth=6, rr_depth=5, direct_samples=-1, work_units=4096, sample_count=a.spp, luminance_samples=100000)
steps = np.tile(np.where((np.arange(a.res) + 0.5) / a.res < 0.5, 0.01, 1.0), (a.res, 1)).astype(np.float32)
if a.map == 'const':
    steps[:] = 0.37
one, two = [], []
for i in range(a.n):
    c = pkg.Context(cfg, sd); c.seed(3000 + i); c.run(a.res * a.res * a.spp); one.append(c.develop().astype(np.float64)); c.close()
    if a.map:
        c = pkg.Context(cfg, sd); c.set_importance_map(steps); c.seed(9000 + i); c.run(a.res * a.res * a.spp); two.append(c.develop().astype(np.float64)); c.close()
    else:
        two.append(pkg.binding.render_two_stage(cfg, sd, 9000 + i, size_reduction=8)[0].astype(np.float64))
one, two = np.array(one), np.array(two)
N = a.n
m1, m2 = one.mean(0), two.mean(0)
pooled = 0.5 * (m1 + m2)
s1 = np.array([rel_mse(x - m1 + pooled, pooled) for x in one]) * N / (N - 1)
s2 = np.array([rel_mse(x - m2 + pooled, pooled) for x in two]) * N / (N - 1)
between, expected = rel_mse(m1 - m2 + pooled, pooled), (s1.mean() + s2.mean()) / N
rng = np.random.default_rng(1); both = np.concatenate([one, two]); null = []
for _ in range(400):
    idx = rng.permutation(2 * N)
    null.append(rel_mse(both[idx[:N]].mean(0) - both[idx[N:]].mean(0) + pooled, pooled))
if a.ref_spp:
    rc = pkg.Context(cfg, sd); ra, rb = rc.render_pt(a.ref_spp // 2, seed=11).astype(np.float64), rc.render_pt(a.ref_spp // 2, seed=22).astype(np.float64); rc.close()
    ref = 0.5 * (ra + rb)
    print("against a %d-spp path-traced reference (own noise %.3g): plain mean %.4g (noise of the mean %.3g), two-stage mean %.4g (%.3g)" %
          (a.ref_spp, rel_mse(ra, rb) / 4, rel_mse(m1, ref), s1.mean() / N, rel_mse(m2, ref), s2.mean() / N))
if a.ref_spp:
    np.set_printoptions(precision=2, suppress=True, linewidth=200)
    B = a.res // 8
    blk = lambda x: (x @ LUMW).reshape(8, B, 8, B).sum((1, 3))
    print("two-stage mean / reference per block:"); print(blk(m2) / blk(ref))
    print("plain mean / reference per block:"); print(blk(m1) / blk(ref))
    lum = pkg.binding.render_two_stage(cfg, sd, 9000, size_reduction=8)[1]
    print("importance map per block (mean), and its minimum:", lum.min()); print(lum.reshape(8, B, 8, B).mean((1, 3)))
    print("reference per block (mean luminance):"); print(blk(ref) / (B * B))
print("%s %s %s: plain vs two-stage, N = %d: ratio %.3f, permutation p %.3f; noise per render plain %.4g two-stage %.4g (medians %.4g / %.4g); mean luminance %.5f / %.5f" %
      (a.scene, a.technique, a.type, N, between / expected, float((np.array(null) >= between).mean()), s1.mean(), s2.mean(), np.median(s1), np.median(s2), (m1 @ LUMW).mean(), (m2 @ LUMW).mean()))
