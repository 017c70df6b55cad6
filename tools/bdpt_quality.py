"""Diagnostic: technique=bdpt image against the device's path tracer (same integrand), per option set."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
pkg = importlib.import_module('drmlt-mitsuba_amd')
abi = pkg.abi
L = np.array([0.212671, 0.715160, 0.072169])
res = int(os.environ.get("RES", "256"))
chains = int(os.environ.get("CHAINS", "65536"))
sd = pkg.scenes.cornell_c2(res)
for ds in (-1, 16):
    for direct in (1, 0):
        cfg = abi.make_config(technique='bdpt', type='orbital', max_depth=8, rr_depth=5, work_units=chains, sample_count=64,
                              no_direct_sampling=1 - direct, direct_samples=ds)
        ctx = pkg.Context(cfg, sd)
        b = ctx.seed_pool(0x5EED, 0, chains)
        for _ in range(3):
            ctx.run(res * res * 64)
        img = ctx.develop()
        ref = ctx.render_pt(1024, seed=3)
        li, lr = img @ L, ref @ L
        blk = lambda a: a.reshape(8, res // 8, 8, res // 8).mean((1, 3))
        e = (li - lr) ** 2 / (lr ** 2 + 1e-2 * lr.mean() ** 2)
        print('direct_samples', ds, 'directSampling', direct, 'b', b, 'mean img', li.mean(), 'mean ref', lr.mean(), 'blk err', np.abs(blk(li) - blk(lr)).mean() / lr.mean(),
              'rmse', float(e.mean()), 'median', float(np.median(e)), 'max', li.max(), lr.max(), 'nonfinite', int((~np.isfinite(li)).sum()),
              'worst', np.unravel_index(np.argmax(e), e.shape), float(e.max()))
        ctx.close()
