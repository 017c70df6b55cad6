"""k_mutate_v4 against k_mutate_v3 on the GPU: identical chains, equal films (sum order aside), all kernel types."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
abi, scenes = pkg.abi, pkg.scenes
LUMW = np.array([0.212671, 0.715160, 0.072169])


def ctx_env(cfg, sd, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return pkg.Context(cfg, sd)
    finally:
        for k, v in old.items():
            if v is None: del os.environ[k]
            else: os.environ[k] = v


ok = True
cases = [("cornell_c2", dict(type="orbital")), ("cornell_c2", dict(type="green")), ("cornell_c2", dict(type="mira")),
         ("cornell_c2", dict(type="orbital", use_mixture=1)), ("cornell_c2", dict(type="green", timid_after_large=1)),
         ("cornell_c2", dict(type="orbital", timid_after_large=1)), ("cornell_c2", dict(type="orbital", acceptance_map=1)),
         ("door_c3", dict(type="green")), ("glass_sphere", dict(type="orbital")), ("caustic_c5", dict(type="mira"))]
for scene, kw in cases:
    sd = scenes.SCENES[scene](res=32)
    n_chains, n_mut = 1000, 48   # 1000: the last wave is ragged
    cfg = abi.make_config(max_depth=8, direct_samples=-1, luminance_samples=20000, work_units=n_chains, sample_count=1, **kw)
    res = []
    for env in (dict(DRMLT_KERNEL=3), dict(DRMLT_KERNEL=4, DRMLT_MH_BATCH=1), dict(DRMLT_KERNEL=4, DRMLT_MH_BATCH=4),
                dict(DRMLT_KERNEL=4, DRMLT_MH_BATCH=32)):
        c = ctx_env(cfg, sd, **env)
        c.seed(0x77)
        c.run(n_chains * n_mut)
        res.append((c.chain_state(34), c.stats(), c.film()))
        c.close()
    (c0, u0), s0, f0 = res[0]
    for i, ((c, u), s, f) in enumerate(res[1:]):
        same_u = np.array_equal(u, u0)
        if not same_u:
            print("      max |du| %.3g, chains with any difference %.4f" % (np.abs(u - u0).max(), np.any(u != u0, axis=1).mean()))
        same_s = s.accepted == s0.accepted and s.rays == s0.rays and s.path_evals == s0.path_evals
        l, l0 = f @ LUMW, f0 @ LUMW
        dsum = abs(l.sum() - l0.sum()) / l0.sum()
        dpix = np.abs(f - f0).max() / max(f0.max(), 1e-30)
        good = same_u and same_s and dsum < 1e-5 and dpix < 1e-4
        ok &= good
        print("%-12s %-40s variant %d: chains equal %s stats equal %s film sum rel diff %.2e max pixel rel diff %.2e %s"
              % (scene, kw, i, same_u, same_s, dsum, dpix, "ok" if good else "FAIL"), flush=True)
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
