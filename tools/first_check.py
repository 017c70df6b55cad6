"""Eyeball check on a GPU box: f(u) parity, seeding, a short run, throughput."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); ob = g.load_oracle()
abi, scenes = pkg.abi, pkg.scenes

for name, sd in (("c1", scenes.cornell_c1(64)), ("c2", scenes.cornell_c2(64)), ("glass", scenes.glass_sphere(64))):
    cfg = abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=1024, luminance_samples=20000, sample_count=16)
    ctx = pkg.Context(cfg, sd); orc = ob.Oracle(abi, cfg, sd, 64)
    u = np.random.default_rng(1).random((4096, 64), dtype=np.float32)
    a, b = ctx.eval_paths(u), orc.eval_paths(u)
    same = a["n_dims"] == b["n_dims"]
    rel = np.abs(a["luminance"] - b["luminance"]) / np.maximum(b["luminance"], 1e-3)
    print(name, "topology match %.4f" % same.mean(), "rel err q50 %.2e q99 %.2e max %.2e" % (np.quantile(rel[same], .5), np.quantile(rel[same], .99), rel[same].max()),
          "mean lum gpu %.6f oracle %.6f" % (a["luminance"].mean(), b["luminance"].mean()), "rays", a["n_rays"].sum(), b["n_rays"].sum())
    bad = np.where(~same)[0][:5]
    for i in bad: print("   mismatch", i, a[i], b[i])
    t = time.time(); bg = ctx.seed(0x5EED); tg = time.time() - t
    bo = orc.seed(0x5EED)
    print("   b gpu %.6f oracle %.6f  seed %.3fs" % (bg, bo, tg))
    cg, ug = ctx.chain_state(34); co, uo = orc.chain_state(34)
    print("   init state: max |x diff|", np.abs(ug - uo).max(), "lum rel", np.max(np.abs(cg["luminance"] - co["luminance"]) / co["luminance"]))
    total = 64 * 64 * 16
    ctx.run(total); orc.run(total, 8)
    sg, so = ctx.stats(), orc.stats()
    print("   gpu   ", {k: round(v, 4) for k, v in sg.ratios().items()}, sg.mutations, sg.path_evals, sg.rays)
    print("   oracle", {k: round(v, 4) for k, v in so.ratios().items()}, so.mutations, so.path_evals, so.rays)
    cg, ug = ctx.chain_state(34); co, uo = orc.chain_state(34)
    close = np.all(np.abs(ug - uo) < 1e-3, axis=1)
    print("   chains with identical trajectory after %d mutations: %.4f" % (total // 1024, close.mean()))
    ig, io = ctx.develop(), orc.develop()
    print("   image mean gpu %.6f oracle %.6f" % (ig.mean(), io.mean()))
    ctx.close(); orc.close()

# throughput: config 2
sd = scenes.cornell_c2(512)
cfg = abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=65536, luminance_samples=655360, sample_count=256)
ctx = pkg.Context(cfg, sd)
t = time.time(); b = ctx.seed(0x5EED); print("C2 seed %.3fs b=%.6f" % (time.time() - t, b))
for rep in range(3):
    n = 65536 * 256
    t = time.time(); ctx.run(n); dt = time.time() - t
    st = ctx.stats()
    print("C2 run %d: %.3fs  %.3e mutations/s  kernel_ms %.1f  ratios %s" % (rep, dt, n / dt, st.kernel_ms, {k: round(v, 3) for k, v in st.ratios().items()}))
print("rays/mutation", st.rays / st.mutations, "evals/mutation", st.path_evals / st.mutations, "accepted frac", st.accepted / st.mutations)
