import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n_mut = int(sys.argv[1]) if len(sys.argv) > 1 else 1
typ = sys.argv[2] if len(sys.argv) > 2 else "orbital"
sd = pkg.scenes.triangle_soup(2000, 32)
n = 1000
cfg = pkg.abi.make_config(type=typ, max_depth=8, direct_samples=-1, luminance_samples=20000, work_units=n, sample_count=1)
res = []
for k in (4, 5):
    os.environ["DRMLT_KERNEL"] = str(k)
    ctx = pkg.Context(cfg, sd)
    ctx.seed(0x5005)
    c0, u0 = ctx.chain_state(34)
    ctx.run(n * n_mut)
    c, u = ctx.chain_state(34)
    res.append((u0, u, c, ctx.stats()))
(a0, a, ca, sa), (b0, b, cb, sb) = res
print("init equal:", np.array_equal(a0, b0))
diff = np.any(a != b, axis=1)
print("chains differing: %d / %d" % (diff.sum(), n), " moved v4 %d v5 %d" % (np.any(a != a0, axis=1).sum(), np.any(b != b0, axis=1).sum()))
print("accepted", sa.accepted, sb.accepted, "rays", sa.rays, sb.rays, "evals", sa.path_evals, sb.path_evals, "second", sa.second_base, sb.second_base)
if diff.any():
    i = np.nonzero(diff)[0][0]
    d = np.nonzero(a[i] != b[i])[0]
    print("chain", i, "dims differing", d[:10], "v4", a[i][d[:6]], "v5", b[i][d[:6]], "init", a0[i][d[:6]])
    md = np.abs(a - b)[diff]
    print("max abs diff quantiles over differing chains:", np.quantile(md.max(axis=1), [0.1, 0.5, 0.9]))
