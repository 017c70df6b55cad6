"""SURVEY 8(d) parity protocol on the GPU box, for every measured configuration:

  python tools/parity_protocol.py --config c2|c3|c5|bdpt [--res 64] [--spp 1024] [--n 16] [--out profiles/r03_parity_protocol_<config>.json]

N device renders (HIP, fp32, through the C-ABI) and N oracle renders (CPU restatement, fp64) of the same scene / config /
budget. Unlike round 2's run the two sides use DIFFERENT seeds (device 1000 + i, oracle 501000 + i): they are independent
estimators, so the comparisons below are statistics about the two implementations, not a restatement of chain-by-chain
tracking (that is what tests/test_gpu_*.py::test_chains_track_the_oracle cover).

  (1) same expectation, reference-free: rMSE between the mean of the N device renders and the mean of the N oracle renders,
      against the level two unbiased estimators of ONE image would show, (s_gpu + s_oracle) / N with s = mean rMSE of a
      single render about its own side's mean (x N / (N - 1)); reported as a ratio (1 = indistinguishable; a bias of the
      size of one render's noise would give ~ N);
  (2) equal budget: per-render noise s_gpu vs s_oracle, relative difference (bound 10 %), with its standard error;
  (3) against an independent reference of the same integrand -- technique=path: device path tracer, two halves (its own
      residual reported); technique=bdpt: the oracle's independent-sample BDPT image (fp64, CPU); technique=mmlt: the sum over
      depths 1..maxDepth of the oracle's independent-sample multiplexed estimator (mmlt drops directly visible emitters,
      pathsampler.cpp:84-320, so a BDPT image is NOT its expectation) -- rMSE of the mean of n renders for n = 1, 2, 4 ...,
      log-log slope (ideal -1), and the budget at which both sides meet north_star's 1e-3. The same curve is also computed
      with the OTHER side's N-render mean as the reference (its noise s / N subtracted): no external estimator involved.

rMSE = mean((I - R)^2 / (R^2 + eps)), eps = 1e-2 mean(R)^2, on luminance.
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

LUMW = np.array([0.212671, 0.715160, 0.072169])

CONFIGS = {
    "c2": dict(scene="cornell_c2", cfg=dict(technique="path", type="orbital", max_depth=8, rr_depth=5), ref="pt",
               what="Cornell box, drmlt technique=path type=orbital (BASELINE configs[1])"),
    "c3": dict(scene="door_c3", cfg=dict(technique="path", type="green", max_depth=8, rr_depth=5), ref="pt",
               what="door scene (occluded light, rough-conductor floor), drmlt technique=path type=green (BASELINE configs[2])"),
    "c5": dict(scene="caustic_c5", cfg=dict(technique="mmlt", type="orbital", max_depth=6, fix_emitter_path=1), ref="mmlt",
               what="glass caustic, drmlt technique=mmlt type=orbital fixEmitterPath, RADIANCE output (BASELINE configs[4] without acceptanceMap)"),
    "bdpt": dict(scene="cornell_c2", cfg=dict(technique="bdpt", type="orbital", max_depth=8, rr_depth=5), ref="bdpt",
                 what="Cornell box, drmlt technique=bdpt type=orbital, directSampling=true"),
    # the specular scene under the other two techniques (glints, caustics: what the Cornell box cannot show)
    "path_c5": dict(scene="caustic_c5", cfg=dict(technique="path", type="orbital", max_depth=6, rr_depth=5), ref="pt",
                    what="glass caustic scene, drmlt technique=path type=orbital"),
    "bdpt_c5": dict(scene="caustic_c5", cfg=dict(technique="bdpt", type="orbital", max_depth=6, rr_depth=5), ref="bdpt",
                    what="glass caustic scene, drmlt technique=bdpt type=orbital, directSampling=true"),
}


def rel_mse(img, ref):
    li, lr = img @ LUMW, ref @ LUMW
    return float(np.mean((li - lr) ** 2 / (lr ** 2 + 1e-2 * lr.mean() ** 2)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--ref-spp", type=int, default=65536, help="device path tracer (technique=path)")
    ap.add_argument("--ref-samples-per-pixel", type=int, default=8192, help="oracle BDPT samples per pixel (bdpt / mmlt)")
    ap.add_argument("--chains", type=int, default=4096)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--oracle-precision", type=int, default=64, choices=(32, 64), help="the oracle chains' arithmetic: 64 = the reference's default build, 32 = its SINGLE_PRECISION build (Epsilon 1e-4 / ShadowEpsilon 1e-3, as the device)")
    ap.add_argument("--seed-offset", type=int, default=0, help="added to every chain seed: an independent repetition of the whole protocol")
    ap.add_argument("--save-means", default="", help=".npz: the two sides' mean images, the reference and every render's luminance estimate b")
    ap.add_argument("--scene", default="", help="override the configuration's scene (a name of drmlt-mitsuba_amd/scenes.py: SCENES)")
    ap.add_argument("--set", default="", help="override configuration fields: k=v,k=v (make_config names, e.g. technique=mmlt,type=green,use_mixture=1)")
    ap.add_argument("--side", default="both", choices=("both", "gpu", "oracle"), help="render one side only and store its renders (--renders): the oracle needs no GPU, so it can run elsewhere; --combine joins the two")
    ap.add_argument("--renders", default="", help=".npz of one side's renders (written with --side gpu|oracle)")
    ap.add_argument("--combine", nargs=2, default=None, metavar=("GPU_NPZ", "ORACLE_NPZ"), help="statistics of two stored sides (implies --no-reference)")
    ap.add_argument("--no-reference", action="store_true", help="skip the independent reference (two-sample statistics only; the slopes against it are then meaningless)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    if a.combine or a.side != "both":
        a.no_reference = True
    conf = dict(CONFIGS[a.config])
    conf["cfg"] = dict(conf["cfg"])
    if a.scene:
        conf["scene"] = a.scene
    for kv in filter(None, a.set.split(",")):
        k, v = kv.split("=")
        conf["cfg"][k] = v if not v.lstrip("-").replace(".", "", 1).isdigit() else (float(v) if "." in v else int(v))
    if a.scene or a.set:
        conf["what"] = "%s with scene %s and %s" % (a.config, conf["scene"], conf["cfg"])
        conf["ref"] = {"path": "pt", "bdpt": "bdpt", "mmlt": "mmlt"}[conf["cfg"].get("technique", "path")]
    pkg, ob = g.load_package(), g.load_oracle()
    if not a.combine and a.side != "gpu":
        ob.build(native=True)   # -O3 -march=native build of the restatement for this host
    abi = pkg.abi
    # "triangle_soup:200": a scene builder with its first argument (here the triangle count: a BVH scene the brute-force oracle can still afford)
    sname, _, sarg = conf["scene"].partition(":")
    sd = pkg.scenes.SCENES[sname](int(sarg), res=a.res) if sarg else pkg.scenes.SCENES[sname](res=a.res)
    kw = dict(direct_samples=-1, work_units=a.chains, sample_count=a.spp, luminance_samples=100000)
    kw.update(conf["cfg"])
    cfg = abi.make_config(**kw)
    total = a.res * a.res * a.spp
    t0 = time.time()
    if a.no_reference:
        ref_a = ref_b = None
        ref_what = "none"
    elif conf["ref"] == "pt":
        ref_ctx = pkg.Context(cfg, sd)
        half = a.ref_spp // 2
        ref_a, ref_b = ref_ctx.render_pt(half, seed=101), ref_ctx.render_pt(half, seed=202)
        ref_ctx.close()
        ref_what = "device path tracer, 2 x %d spp" % half
    elif conf["ref"] == "bdpt":
        ro = ob.Oracle(abi, cfg, sd, precision=64, native=True)
        nsamp = a.res * a.res * a.ref_samples_per_pixel // 2
        ref_a, ref_b = ro.bdpt_render(nsamp, seed=101, nthreads=a.threads), ro.bdpt_render(nsamp, seed=202, nthreads=a.threads)
        ro.close()
        ref_what = "oracle's independent-sample BDPT image (fp64, CPU), 2 x %d samples per pixel" % (a.ref_samples_per_pixel // 2)
    else:
        ro = ob.Oracle(abi, cfg, sd, precision=64, native=True)
        nsamp = a.res * a.res * a.ref_samples_per_pixel // 2
        md = conf["cfg"]["max_depth"]
        ref_a = sum(ro.mmlt_render(d, nsamp, seed=101 + d, nthreads=a.threads)[0].astype(np.float64) for d in range(1, md + 1))
        ref_b = sum(ro.mmlt_render(d, nsamp, seed=202 + d, nthreads=a.threads)[0].astype(np.float64) for d in range(1, md + 1))
        ro.close()
        ref_what = "oracle's independent-sample multiplexed estimator summed over depths 1..%d (fp64, CPU), 2 x %d samples per pixel and depth" % (md, a.ref_samples_per_pixel // 2)
    if a.combine:
        ga, oa = np.load(a.combine[0]), np.load(a.combine[1])
        gpu, orc, b_gpu, b_orc = list(ga["renders"]), list(oa["renders"]), list(ga["b"]), list(oa["b"])
        a.n = min(len(gpu), len(orc)); gpu, orc = gpu[:a.n], orc[:a.n]
        ref = 0.5 * (np.mean(gpu, 0) + np.mean(orc, 0)); ref_noise = 0.0; ref_what = "none (pooled mean)"; t_gpu = t_orc = 0.0
    else:
        if a.no_reference and a.side != "oracle":   # stand-in so that the per-render progress lines mean something: a longer device render
            c = pkg.Context(cfg, sd); c.seed(77); c.run(4 * total); ref_a = ref_b = c.develop().astype(np.float64); c.close()
        elif a.no_reference:
            o = ob.Oracle(abi, cfg, sd, precision=a.oracle_precision, native=True); o.seed(77); o.run(total, a.threads); ref_a = ref_b = o.develop().astype(np.float64); o.close()
        ref = 0.5 * (ref_a.astype(np.float64) + ref_b.astype(np.float64))
        ref_noise = rel_mse(ref_a, ref_b) / 4.0            # Var(mean of halves) = Var(difference) / 4
        print("reference: %s, own rMSE %.3g (%.0f s)" % (ref_what, ref_noise, time.time() - t0), flush=True)
        gpu, orc, b_gpu, b_orc = [], [], [], []
        t_gpu = t_orc = 0.0
        for i in range(a.n):
            if a.side != "oracle":
                t = time.time()
                c = pkg.Context(cfg, sd)
                b_gpu.append(c.seed(1000 + a.seed_offset + i)); c.run(total); gpu.append(c.develop().astype(np.float64)); c.close()
                t_gpu += time.time() - t
            if a.side != "gpu":
                t = time.time()
                o = ob.Oracle(abi, cfg, sd, precision=a.oracle_precision, native=True)
                b_orc.append(o.seed(501000 + a.seed_offset + i)); o.run(total, a.threads); orc.append(o.develop().astype(np.float64)); o.close()
                t_orc += time.time() - t
            if (i + 1) % 8 == 0:
                print("render %d/%d (%.0f s)" % (i + 1, a.n, time.time() - t0), flush=True)
        if a.side != "both":
            rs, bs = (gpu, b_gpu) if a.side == "gpu" else (orc, b_orc)
            np.savez_compressed(a.renders, renders=np.array(rs, dtype=np.float32), b=np.array(bs))
            print("stored %d %s renders in %s" % (len(rs), a.side, a.renders))
            return
    gpu, orc = np.array(gpu), np.array(orc)
    N = a.n
    if a.save_means:
        np.savez(a.save_means, gpu_mean=gpu.mean(0), orc_mean=orc.mean(0), gpu_var=gpu.var(0, ddof=1), orc_var=orc.var(0, ddof=1), ref=ref, b_gpu=np.array(b_gpu), b_orc=np.array(b_orc))
    mg, mo = gpu.mean(0), orc.mean(0)
    pooled = 0.5 * (mg + mo)
    # (2) per-render noise about the side's own mean (unbiased: x N / (N - 1)), measured with the pooled mean as denominator image
    sg = np.array([rel_mse(x - mg + pooled, pooled) for x in gpu]) * N / (N - 1)
    so = np.array([rel_mse(x - mo + pooled, pooled) for x in orc]) * N / (N - 1)
    equal_budget = (sg.mean() - so.mean()) / so.mean()
    equal_budget_se = float(np.sqrt(sg.var(ddof=1) / N + so.var(ddof=1) / N) / so.mean())
    # (1) two-sample: rMSE between the two means vs what two unbiased estimators of one image would show
    between = rel_mse(mg - mo + pooled, pooled)
    expected = (sg.mean() + so.mean()) / N
    # ... and the same statistic under the null hypothesis, by permutation: the 2N renders dealt into two groups of N at random. MLT noise
    # is heavy-tailed (a chain parked on a caustic), so the ratio's spread about 1 is wide: the p-value says where the observed one sits
    rng = np.random.default_rng(12345)
    both = np.concatenate([gpu, orc])
    null = []
    for _ in range(400):
        idx = rng.permutation(2 * N)
        null.append(rel_mse(both[idx[:N]].mean(0) - both[idx[N:]].mean(0) + pooled, pooled))
    null = np.array(null)
    p_value = float((null >= between).mean())
    # (3) against the reference
    single_g = np.array([rel_mse(x, ref) for x in gpu]); single_o = np.array([rel_mse(x, ref) for x in orc])
    ns = [n for n in (1, 2, 4, 8, 16, 32) if n <= N]
    def curve(imgs):
        return [float(np.mean([rel_mse(imgs[k:k + n].mean(0), ref) for k in range(0, N - n + 1, n)])) for n in ns]
    cg, co = curve(gpu), curve(orc)
    slope = lambda c: float(np.polyfit(np.log(ns), np.log(np.maximum(np.array(c) - ref_noise, 1e-12)), 1)[0])
    meet = [n * a.spp for n, x, y in zip(ns, cg, co) if x < 1e-3 and y < 1e-3]
    # the same curves against the other side's N-render mean (noise of that mean: s / N, subtracted before the fit)
    def curve_x(imgs, other_mean):
        return [float(np.mean([rel_mse(imgs[k:k + n].mean(0), other_mean) for k in range(0, N - n + 1, n)])) for n in ns]
    xg, xo = curve_x(gpu, mo), curve_x(orc, mg)
    slope_x = lambda c, noise: float(np.polyfit(np.log(ns), np.log(np.maximum(np.array(c) - noise, 1e-12)), 1)[0])
    out_name = a.out or ""
    out = {
        "command": "python tools/parity_protocol.py " + " ".join(sys.argv[1:]),
        "scene": "%s %dx%d: %s, %d chains, %d mutations/pixel per render" % (conf["scene"], a.res, a.res, conf["what"], a.chains, a.spp),
        "n_renders": N, "seeds": "device %d + i, oracle %d + i (independent estimators)" % (1000 + a.seed_offset, 501000 + a.seed_offset), "oracle_chain_precision": a.oracle_precision,
        "reference": "%s; residual rMSE of the reference itself %.3g" % (ref_what, ref_noise),
        "two_sample": {"rmse_between_means": between, "expected_if_same_expectation": expected, "ratio": between / expected,
                       "permutation_test": {"n_permutations": 400, "p_value": p_value, "null_median": float(np.median(null)), "null_95th_percentile": float(np.quantile(null, 0.95)),
                                            "note": "the statistic recomputed with the 2N renders dealt at random into two groups: p = share of deals with a distance at least the observed one"},
                       "note": "ratio ~ 1: the two means differ by no more than their own noise; a bias as large as one render's noise would give ~ %d" % N},
        "noise_single_render_about_own_mean": {"gpu_mean": float(sg.mean()), "gpu_std": float(sg.std(ddof=1)), "oracle_mean": float(so.mean()), "oracle_std": float(so.std(ddof=1))},
        "equal_budget_relative_difference": float(equal_budget), "equal_budget_standard_error": equal_budget_se, "equal_budget_bound": 0.10,
        "rmse_single_render_vs_reference": {"gpu_mean": float(single_g.mean()), "gpu_std": float(single_g.std()), "oracle_mean": float(single_o.mean()), "oracle_std": float(single_o.std())},
        "rmse_of_mean_of_n_vs_reference": {"n": ns, "gpu": cg, "oracle": co},
        "loglog_slope_vs_n_after_subtracting_reference_noise": {"gpu": slope(cg), "oracle": slope(co), "ideal": -1.0},
        "rmse_of_mean_of_n_vs_other_sides_mean": {"n": ns, "gpu_vs_oracle_mean": xg, "oracle_vs_gpu_mean": xo,
                                                  "slope_gpu": slope_x(xg, so.mean() / N), "slope_oracle": slope_x(xo, sg.mean() / N)},
        "noise_median_single_render": {"gpu": float(np.median(sg)), "oracle": float(np.median(so)),
                                       "relative_difference_of_medians": float((np.median(sg) - np.median(so)) / np.median(so))},
        "budget_mutations_per_pixel_where_both_meet_1e-3": (min(meet) if meet else None),
        "summary": {"config": a.config, "two_sample_ratio": between / expected, "two_sample_permutation_p": p_value, "equal_budget_rel_diff": float(equal_budget), "equal_budget_se": equal_budget_se,
                    "equal_budget_rel_diff_of_medians": float((np.median(sg) - np.median(so)) / np.median(so)),
                    "bound": 0.10, "slope_gpu": slope(cg), "slope_oracle": slope(co),
                    "slope_gpu_vs_oracle_mean": slope_x(xg, so.mean() / N), "slope_oracle_vs_gpu_mean": slope_x(xo, sg.mean() / N),
                    "rmse_lt_1e-3_at_mutations_per_pixel": (min(meet) if meet else None),
                    "rmse_gpu_at_that_budget": (cg[ns.index(min(meet) // a.spp)] if meet else None),
                    "rmse_oracle_at_that_budget": (co[ns.index(min(meet) // a.spp)] if meet else None),
                    "source": "%s (tools/parity_protocol.py --config %s, N = %d independent renders per side at %dx%d)" % (os.path.basename(out_name) or "stdout", a.config, N, a.res, a.res)},
        "seconds": {"total": time.time() - t0, "gpu_renders": t_gpu, "oracle_renders": t_orc},
    }
    print(json.dumps(out["summary"]))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
