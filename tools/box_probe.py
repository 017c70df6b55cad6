import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as e
pkg = e.load_package()
def ctx(cfg, sd, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try: return pkg.Context(cfg, sd)
    finally:
        for k, v in old.items():
            if v is None: del os.environ[k]
            else: os.environ[k] = v
for scene in ("cornell_c2", "door_c3", "caustic_c5"):
    sd = pkg.scenes.SCENES[scene](res=48)
    cfg = pkg.abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=64)
    u = np.random.default_rng(17).random((32768, 50), dtype=np.float32)
    a = ctx(cfg, sd, DRMLT_NO_BOX_MERGE=1).eval_paths(u)
    b = ctx(cfg, sd).eval_paths(u)
    same = (a["n_dims"] == b["n_dims"]) & (a["n_rays"] == b["n_rays"])
    rel = np.abs(a["luminance"] - b["luminance"])[same] / np.maximum(a["luminance"][same], 1e-6)
    print(scene, "same", same.mean(), "q50 %.3g q99 %.3g q999 %.3g max %.3g" % tuple(np.quantile(rel, [0.5, 0.99, 0.999, 1.0])), "n>2e-4:", (rel > 2e-4).sum(), "sum ratio", a["luminance"].sum() / b["luminance"].sum())
    bad = np.where(same)[0][np.argsort(rel)[-5:]]
    for i in bad: print("   ", i, a["luminance"][i], b["luminance"][i], a["n_dims"][i], a["n_rays"][i])
