"""Diagnostic: device vs oracle f(u) for technique=mmlt, broken down by depth and strategy."""
import sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package(); ob = g.load_oracle()
abi = pkg.abi
name = sys.argv[1] if len(sys.argv) > 1 else 'cornell_c2'
sd = pkg.scenes.SCENES[name](res=64)
cfg = abi.make_config(technique='mmlt', type='orbital', max_depth=6, direct_samples=-1, work_units=64)
ctx = pkg.Context(cfg, sd); orc = ob.Oracle(abi, cfg, sd, 64)
rng = np.random.default_rng(7)
n = 8192
us, ue, ud = rng.random((n, 14), dtype=np.float32), rng.random((n, 14), dtype=np.float32), rng.random(n, dtype=np.float32)
for depth in range(1, 7):
    gs, stg = ctx.eval_paths_mmlt(depth, us, ue, ud)
    o, sto = orc.mmlt_eval(depth, us, ue, ud)
    for s in range(depth + 1):
        m = sto[:, 0] == s
        if not m.any():
            continue
        rays = (gs['n_rays'] == o['n_rays'])[m].mean()
        posq = ((gs['luminance'] > 0) == (o['luminance'] > 0))[m].mean()
        pos = m & (gs['luminance'] > 0) & (o['luminance'] > 0)
        rel = np.abs(gs['luminance'] - o['luminance'])[pos] / o['luminance'][pos] if pos.any() else np.zeros(1)
        dx = np.abs(gs['x'] - o['x'])[pos].max() if pos.any() else 0
        dy = np.abs(gs['y'] - o['y'])[pos].max() if pos.any() else 0
        print('dx=%.2e dy=%.2e ' % (dx, dy), end='')
        print('d=%d s=%d n=%4d rays_eq=%.4f pos_eq=%.4f dims_eq=%.4f rel99=%.2e relmax=%.2e sum_g=%.5f sum_o=%.5f' % (
            depth, s, m.sum(), rays, posq, (gs['n_dims'] == o['n_dims'])[m].mean(), np.quantile(rel, 0.99), rel.max(),
            gs['luminance'][m].sum(), o['luminance'][m].sum()))
