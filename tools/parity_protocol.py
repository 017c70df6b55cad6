"""SURVEY 8(d) parity protocol, items (1)-(3), on the GPU box:

  (1) unbiasedness: N independent GPU renders and N oracle renders (CPU restatement, fp64) of config 2's scene at the same
      budget, against one high-spp plain path-traced reference of the same integrand; the rMSE of the mean of the first n
      renders must fall ~ 1/n;
  (2) equal budget: |rMSE_gpu - rMSE_oracle| / rMSE_oracle < 10 % (mean over the N single renders);
  (3) the budget at which BOTH meet north_star's rMSE < 1e-3 (mean of n renders of budget B ~ budget n B).

  python tools/parity_protocol.py [--res 64] [--spp 512] [--n 16] [--ref-spp 65536] [--out profiles/r02_parity_protocol.json]

rMSE = mean((I - R)^2 / (R^2 + eps)), eps = 1e-2 mean(R)^2, on luminance. The reference is path traced on the
device (the oracle's own path tracer agrees with it, tests/test_gpu_parity.py); its residual noise ~ 1 / ref-spp is reported.
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

LUMW = np.array([0.212671, 0.715160, 0.072169])


def rel_mse(img, ref):
    li, lr = img @ LUMW, ref @ LUMW
    return float(np.mean((li - lr) ** 2 / (lr ** 2 + 1e-2 * lr.mean() ** 2)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--spp", type=int, default=512)
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--ref-spp", type=int, default=65536)
    ap.add_argument("--chains", type=int, default=4096)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    pkg, ob = g.load_package(), g.load_oracle()
    ob.build(native=True)   # -O3 -march=native build of the restatement for this host
    abi = pkg.abi
    sd = pkg.scenes.cornell_c2(a.res)
    kw = dict(type="orbital", max_depth=8, rr_depth=5, direct_samples=-1, work_units=a.chains, sample_count=a.spp, luminance_samples=100000)
    cfg = abi.make_config(**kw)
    total = a.res * a.res * a.spp
    t0 = time.time()
    ref_ctx = pkg.Context(cfg, sd)
    half = a.ref_spp // 2
    ref_a, ref_b = ref_ctx.render_pt(half, seed=101), ref_ctx.render_pt(half, seed=202)   # two halves: the reference's own noise
    ref = 0.5 * (ref_a + ref_b)
    ref_noise = rel_mse(ref_a, ref_b) / 4.0            # Var(mean of halves) = Var(difference) / 4
    gpu, orc = [], []
    for i in range(a.n):
        c = pkg.Context(cfg, sd)
        c.seed(1000 + i); c.run(total); gpu.append(c.develop()); c.close()
        o = ob.Oracle(abi, cfg, sd, precision=64, native=True)
        o.seed(1000 + i); o.run(total, a.threads); orc.append(o.develop()); o.close()
        print("render %d/%d  gpu rMSE %.4g  oracle rMSE %.4g  (%.0f s)" % (i + 1, a.n, rel_mse(gpu[-1], ref), rel_mse(orc[-1], ref), time.time() - t0), flush=True)
    gpu, orc = np.array(gpu), np.array(orc)
    single_g = np.array([rel_mse(x, ref) for x in gpu]); single_o = np.array([rel_mse(x, ref) for x in orc])
    ns = [n for n in (1, 2, 4, 8, 16, 32) if n <= a.n]
    # mean over disjoint groups of n renders, so that every render is used at every n
    def curve(imgs):
        return [float(np.mean([rel_mse(imgs[k:k + n].mean(0), ref) for k in range(0, a.n - n + 1, n)])) for n in ns]
    cg, co = curve(gpu), curve(orc)
    slope = lambda c: float(np.polyfit(np.log(ns), np.log(np.maximum(np.array(c) - ref_noise, 1e-12)), 1)[0])
    equal_budget = abs(single_g.mean() - single_o.mean()) / single_o.mean()
    meet = [n * a.spp for n, x, y in zip(ns, cg, co) if x < 1e-3 and y < 1e-3]
    out = {
        "command": "python tools/parity_protocol.py " + " ".join(sys.argv[1:]),
        "scene": "cornell_c2 %dx%d, drmlt technique=path type=orbital, %d chains, %d mutations/pixel per render" % (a.res, a.res, a.chains, a.spp),
        "n_renders": a.n, "reference": "device path tracer, %d spp (two independent halves); residual rMSE of the reference itself %.3g" % (a.ref_spp, ref_noise),
        "rmse_single_render": {"gpu_mean": float(single_g.mean()), "gpu_std": float(single_g.std()), "oracle_mean": float(single_o.mean()), "oracle_std": float(single_o.std())},
        "rmse_of_mean_of_n": {"n": ns, "gpu": cg, "oracle": co},
        "loglog_slope_vs_n_after_subtracting_reference_noise": {"gpu": slope(cg), "oracle": slope(co), "ideal": -1.0},
        "equal_budget_relative_difference": float(equal_budget),
        "mean_image_gpu_vs_oracle_rmse": rel_mse(gpu.mean(0), orc.mean(0)),
        "budget_mutations_per_pixel_where_both_meet_1e-3": (min(meet) if meet else None),
        "summary": {"equal_budget_rel_diff": float(equal_budget), "bound": 0.10, "slope_gpu": slope(cg), "slope_oracle": slope(co),
                    "rmse_lt_1e-3_at_mutations_per_pixel": (min(meet) if meet else None),
                    "rmse_gpu_at_that_budget": (cg[ns.index(min(meet) // a.spp)] if meet else None),
                    "rmse_oracle_at_that_budget": (co[ns.index(min(meet) // a.spp)] if meet else None),
                    "source": "profiles/r02_parity_protocol.json (tools/parity_protocol.py, N = %d renders each)" % a.n},
        "seconds": time.time() - t0,
    }
    print(json.dumps(out["summary"]))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
