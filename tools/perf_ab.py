"""Interleaved A/B timing of k_mutate variants in ONE process (DRMLT_DEBUG bit masks / env knobs per context)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
abi, scenes = pkg.abi, pkg.scenes

variants = [v for v in (sys.argv[1] if len(sys.argv) > 1 else "0,64").split(",")]
chains = int(os.environ.get("CHAINS", 65536))
spp = int(os.environ.get("SPP", 64))
res = int(os.environ.get("RES", 512))
typ = os.environ.get("TYPE", "orbital")
sd = scenes.triangle_soup(int(os.environ.get("N_TRIS", 2000)), res) if os.environ.get("SCENE") == "triangle_soup" else scenes.SCENES[os.environ.get("SCENE", "cornell_c2")](res=res)
ctxs = {}
for v in variants:
    env = dict(kv.split("=") for kv in v.split("+") if "=" in kv)
    dbg = [kv for kv in v.split("+") if "=" not in kv]
    os.environ["DRMLT_DEBUG"] = dbg[0] if dbg else "0"
    for k, val in env.items():
        os.environ[k] = val
    n = int(env.get("CHAINS", chains))
    cfg = abi.make_config(type=typ, max_depth=8, direct_samples=-1, work_units=n, luminance_samples=10 * n, sample_count=spp)
    c = pkg.Context(cfg, sd)
    c.seed(0x5EED)
    for k in env:
        del os.environ[k]
    ctxs[v] = c
total = res * res * spp
for v, c in ctxs.items():
    c.run(total)  # warmup
rates = {v: [] for v in variants}
for rep in range(int(os.environ.get("REPS", 5))):
    for v, c in ctxs.items():
        t = time.perf_counter(); c.run(total); dt = time.perf_counter() - t
        rates[v].append(total / dt)
for v in variants:
    r = np.array(rates[v])
    st = ctxs[v].stats()
    if os.environ.get("ALL"): print("   rates:", " ".join("%.3e" % x for x in r))
    print("variant %-24s median %.4e  min %.4e  max %.4e mut/s   rays/mut %.2f  acc %.3f  bvh nodes/ray %.1f prims/ray %.1f" % (v, np.median(r), r.min(), r.max(), st.rays / st.mutations, st.accepted / st.mutations, st.bvh_node_visits / max(st.rays, 1), st.bvh_prim_tests / max(st.rays, 1)))
