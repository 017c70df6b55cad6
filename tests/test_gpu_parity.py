"""GPU parity tests: every call goes through the C-ABI (libdrmlt_amd.so) and is compared with the CPU oracle
on identical inputs. Floating-point path => tolerances are stated per test:
  * f(u) on identical PSS points: same path topology for >= 99.5 % of points, luminance within 1e-3 relative
    at the 99th percentile (fp32 device vs fp64 oracle; discontinuities of f account for the rest);
  * chains: the addressed RNG makes device and oracle chains comparable mutation by mutation;
  * images: relative MSE with BASELINE.md's epsilon."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LUMW = np.array([0.212671, 0.715160, 0.072169])


def lum(img):
    return img @ LUMW


def rel_mse(img, ref):
    li, lr = lum(img), lum(ref)
    return float(np.mean((li - lr) ** 2 / (lr ** 2 + 1e-2 * lr.mean() ** 2)))


def make(pkg, ob, sd, precision=64, **kw):
    abi = pkg.abi
    base = dict(max_depth=8, rr_depth=5, direct_samples=-1, luminance_samples=20000)
    base.update(kw)
    cfg = abi.make_config(**base)
    return cfg, pkg.Context(cfg, sd), ob.Oracle(abi, cfg, sd, precision)


SCENES = ["cornell_c1", "cornell_c2", "glass_sphere", "door_c3", "door_ggx", "triangle_soup", "caustic_c5"]


@pytest.mark.parametrize("name", SCENES)
def test_eval_paths_matches_oracle(pkg, ob, name, native_lib):
    if name == "triangle_soup": sd = pkg.scenes.triangle_soup(600, 64)
    elif name == "door_ggx": sd = pkg.scenes.door_c3(64, ggx=True)      # the GGX branch of the microfacet sampler
    else: sd = pkg.scenes.SCENES[name](res=64)
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", work_units=64)
    u = np.random.default_rng(1).random((8192, 50), dtype=np.float32)
    g, o = ctx.eval_paths(u), orc.eval_paths(u)
    same = g["n_dims"] == o["n_dims"]
    assert same.mean() >= 0.995, same.mean()
    # the device skips the shadow ray when the BSDF value is already zero (back-face hits), the reference
    # tests visibility first (scene.cpp:890-895): never more rays, identical counts on one-sided scenes
    assert np.all(g["n_rays"][same] <= o["n_rays"][same])
    if name in ("cornell_c1", "cornell_c2", "glass_sphere"):  # one-sided diffuse scenes: the BSDF value is never 0
        assert (g["n_rays"] == o["n_rays"])[same].mean() > 0.995
    assert np.allclose(g["x"], o["x"], atol=1e-3) and np.allclose(g["y"], o["y"], atol=1e-3)
    rel = np.abs(g["luminance"] - o["luminance"])[same] / np.maximum(o["luminance"][same], 1e-3)
    assert np.quantile(rel, 0.99) < 1e-3, np.quantile(rel, 0.99)
    assert g["luminance"].mean() == pytest.approx(o["luminance"].mean(), rel=5e-3)
    assert np.allclose(g["rgb"][same], o["rgb"][same], rtol=5e-2, atol=1e-3)


def test_bvh_and_brute_force_agree_on_device(pkg, ob, native_lib):
    sd = pkg.scenes.cornell_c2(64)
    u = np.random.default_rng(3).random((8192, 50), dtype=np.float32)
    cfg = pkg.abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=64)
    os.environ["DRMLT_BVH_THRESHOLD"] = "1000000"
    os.environ["DRMLT_NO_BOX_MERGE"] = "1"   # the loop over the separate faces, as the leaves hold them (cuboid records: tests/test_gpu_boxes.py)
    a = pkg.Context(cfg, sd).eval_paths(u)
    os.environ["DRMLT_BVH_THRESHOLD"] = "0"
    try:
        b = pkg.Context(cfg, sd).eval_paths(u)
    finally:
        del os.environ["DRMLT_BVH_THRESHOLD"], os.environ["DRMLT_NO_BOX_MERGE"]
    same = a["n_dims"] == b["n_dims"]
    assert same.mean() > 0.999
    assert np.allclose(a["luminance"][same], b["luminance"][same], rtol=1e-4, atol=1e-6)


def test_deep_bvh_spills_its_traversal_stack(pkg, ob, native_lib, capfd):
    """Geometrically shrinking triangles make SAH peel a few primitives per level (a chain > 20 levels deep for this
    scene). The kernels' stack keeps 24 entries in LDS and a 4-wide node pushes up to 3: with the builder left free the
    4-wide tree is deeper than 8 levels and the stacks must spill to (and refill from) their overflow area in memory;
    DRMLT_BVH_MAX_DEPTH bounds the binary depth instead (median-split fallback). Either way the traversal must agree
    with the brute-force loop."""
    import re
    sd = pkg.scenes.cornell_c2(64)
    white = 0
    for i in range(160):
        s = 0.7 * 0.5 ** (i * 0.3)        # down to 2e-15: areas stay representable in fp32
        x = 0.9 * 0.5 ** (i * 0.3)        # clustered towards the origin, where fp32 keeps resolving them
        sd.triangle((x, 0.0, 0.0), (x + 0.3 * s, 0.0, 0.0), (x, 0.3 * s, 0.1 * s), white)
    u = np.random.default_rng(5).random((8192, 50), dtype=np.float32)
    cfg = pkg.abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=64)
    os.environ["DRMLT_BVH_THRESHOLD"] = "1000000"
    os.environ["DRMLT_NO_BOX_MERGE"] = "1"
    try:
        a = pkg.Context(cfg, sd).eval_paths(u)
    finally:
        del os.environ["DRMLT_NO_BOX_MERGE"]
    os.environ["DRMLT_BVH_THRESHOLD"] = "0"
    os.environ["DRMLT_VERBOSE"] = "1"
    try:
        for bound, want_median in (("64", False), ("9", True)):
            os.environ["DRMLT_BVH_MAX_DEPTH"] = bound
            b = pkg.Context(cfg, sd).eval_paths(u)
            log = capfd.readouterr().err
            m = re.search(r"BVH: (\d+) primitives, (\d+) binary / (\d+) 4-wide nodes, 4-wide depth (\d+) \(stack 24 in LDS \+ (\d+) in memory\), (\d+) median splits", log)
            assert m and int(m.group(3)) < int(m.group(2)) and (int(m.group(6)) > 0) == want_median, log
            if not want_median:
                assert 3 * int(m.group(4)) > 24 and int(m.group(5)) > 0, log          # deeper than the LDS column: the overflow area is in use
            same = a["n_dims"] == b["n_dims"]
            assert same.mean() > 0.999
            assert np.allclose(a["luminance"][same], b["luminance"][same], rtol=1e-4, atol=1e-6)
    finally:
        del os.environ["DRMLT_BVH_THRESHOLD"], os.environ["DRMLT_VERBOSE"], os.environ["DRMLT_BVH_MAX_DEPTH"]


def test_bootstrap_and_seed_replay(pkg, ob, native_lib):
    sd = pkg.scenes.cornell_c2(64)
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", work_units=4096, sample_count=1)
    bg, bo = ctx.seed(0x5EED), orc.seed(0x5EED)       # seed() raises DRMLT_E_REPLAY on a luminance mismatch
    assert bg == pytest.approx(bo, rel=2e-4)
    (cg, ug), (co, uo) = ctx.chain_state(34), orc.chain_state(34)
    # (1) the picks, deterministically: the oracle's resampler (the second half of generateSeeds, pathsampler.cpp:936-957)
    # run on the DEVICE's own bootstrap luminances must pick exactly the device's seeds -- same CDF arithmetic, same
    # lower_bound / zero-mass handling, same TAG_SEEDSEL stream. An off-by-one or a wrong stream cannot pass this.
    n_boot = 10 * 4096                                 # max(luminanceSamples = 20000, 10 x workUnits), drmlt.cpp:454-466
    lum_dev = ctx.bootstrap_luminances(0x5EED, 0, n_boot)
    picks = ob.select_seeds(lum_dev, 0x5EED, 0, 4096)
    assert np.array_equal(picks, ctx.seed_indices())
    assert np.array_equal(cg["luminance"] > 0, np.ones(4096, bool)) and np.all(lum_dev[picks] > 0)
    # (2) device (fp32) and oracle (fp64) luminances agree to ~1e-6, except for a few samples per thousand that sit on a
    # discontinuity of f (another surface hit, another roulette outcome) and differ by O(their own size). Each of those
    # moves the CDF under every later pick by a sizeable part of one sample's mass -- about one CDF slot -- so the two seed
    # SETS differ widely (measured 0.67 overlap) although nothing is wrong. Shown, not assumed: give the oracle's luminance
    # array the device's values for just those samples and the picks coincide again.
    lum_orc = orc.bootstrap_lum(0x5EED, 0, n_boot)
    rel = np.abs(lum_dev - lum_orc) / np.maximum(np.maximum(lum_orc, lum_dev), 1e-6)
    jumps = rel > 1e-3
    assert np.quantile(rel, 0.99) < 1e-4 and 0 < jumps.sum() < 3e-3 * n_boot, (np.quantile(rel, 0.99), jumps.sum())
    picks_orc = ob.select_seeds(lum_orc, 0x5EED, 0, 4096)
    overlap = np.intersect1d(picks, picks_orc).size / np.unique(picks_orc).size
    assert overlap > 0.5, overlap
    picks_fixed = ob.select_seeds(np.where(jumps, lum_dev, lum_orc), 0x5EED, 0, 4096)
    overlap_fixed = np.intersect1d(picks, picks_fixed).size / np.unique(picks_fixed).size
    assert overlap_fixed >= 0.95, (overlap, overlap_fixed, int(jumps.sum()))
    keys_g = {r.tobytes() for r in ug}
    keys_o = {r.tobytes() for r in uo}
    assert len(keys_g & keys_o) / len(keys_o) > 0.5    # the oracle's own seeding (its own luminances): same effect
    # replay: the stored current state is f(u) of the stored vector (drmlt_proc.cpp:481,509-512)
    chk = orc.eval_paths(np.pad(ug, ((0, 0), (0, 16))))
    ok = np.abs(chk["luminance"] - cg["luminance"]) <= 1e-3 * cg["luminance"]
    assert ok.mean() > 0.995
    assert np.allclose(chk["x"][ok], cg["x"][ok], atol=1e-3)
    # luminance-proportional resampling (pathsampler.cpp:946-954): E[lum of a seed] = sum(l^2) / sum(l)
    assert cg["luminance"].mean() == pytest.approx((lum_orc ** 2).sum() / lum_orc.sum(), rel=0.06)
    st = ctx.stats()
    assert st.n_chains == 4096 and st.max_dim == 50


VARIANTS = [
    dict(type="orbital"), dict(type="green"), dict(type="mira"),
    dict(type="orbital", use_mixture=1), dict(type="green", use_mixture=1),
    dict(type="orbital", timid_after_large=1), dict(type="mira", timid_after_large=1),
    dict(type="orbital", p_large=0.05), dict(type="orbital", direct_samples=16),
]


@pytest.mark.parametrize("kw", VARIANTS, ids=lambda k: "-".join("%s=%s" % i for i in k.items()))
def test_chains_track_the_oracle(pkg, ob, kw, native_lib):
    sd = pkg.scenes.cornell_c2(32)
    n_chains, n_mut = 2048, 48
    cfg, ctx, orc = make(pkg, ob, sd, work_units=n_chains, sample_count=1, **kw)
    ctx.seed(0xABCD), orc.seed(0xABCD)
    (c0g, u0g), (c0o, u0o) = ctx.chain_state(34), orc.chain_state(34)
    same0 = np.all(u0g == u0o, axis=1)
    ctx.run(n_chains * n_mut)
    orc.run(n_chains * n_mut, 8)
    (cg, ug), (co, uo) = ctx.chain_state(34), orc.chain_state(34)
    tracked = np.all(np.abs(ug - uo) < 2e-3, axis=1) & same0
    # a chain leaves the oracle's trajectory only when an acceptance test is decided differently in fp32
    assert tracked.sum() / same0.sum() > 0.97, tracked.sum() / same0.sum()
    sg, so = ctx.stats(), orc.stats()
    assert sg.mutations == so.mutations == n_chains * n_mut
    for k in ("first", "large", "bold", "second", "second_large", "second_bold", "overall"):
        bg, bo = getattr(sg, k + "_base"), getattr(so, k + "_base")
        assert abs(bg - bo) <= 0.01 * max(bo, 1) + 20, (k, bg, bo)
        if bo > 200:
            pg, po = getattr(sg, k + "_acc") / bg, getattr(so, k + "_acc") / bo
            assert abs(pg - po) < 4 * np.sqrt(po * (1 - po) / bo) + 0.01, (k, pg, po)
    assert abs(sg.path_evals - so.path_evals) <= 0.01 * so.path_evals
    assert abs(sg.rays - so.rays) <= 0.02 * so.rays
    fg, fo = ctx.film(), orc.film()
    assert lum(fg).sum() == pytest.approx(lum(fo).sum(), rel=2e-3)
    # same chains splat the same pixels: block-averaged films agree
    bgk, bok = (lum(f).reshape(8, 4, 8, 4).sum(axis=(1, 3)) for f in (fg, fo))
    assert np.abs(bgk - bok).sum() / bok.sum() < 0.06
    ig, io = ctx.develop(), orc.develop()
    assert lum(ig).mean() == pytest.approx(lum(io).mean(), rel=2e-3)


def test_config3_door_scene_green(pkg, ob, native_lib):
    """BASELINE config 3: occluded light behind an ajar partition, rough-conductor floor, type=green."""
    sd = pkg.scenes.door_c3(48)
    n_chains, n_mut = 2048, 32
    cfg, ctx, orc = make(pkg, ob, sd, type="green", work_units=n_chains, sample_count=1, luminance_samples=100000)
    bg, bo = ctx.seed(0xD00D), orc.seed(0xD00D)
    assert bg == pytest.approx(bo, rel=5e-3)
    (c0g, u0g), (c0o, u0o) = ctx.chain_state(34), orc.chain_state(34)
    same0 = np.all(u0g == u0o, axis=1)
    ctx.run(n_chains * n_mut), orc.run(n_chains * n_mut, 8)
    (cg, ug), (co, uo) = ctx.chain_state(34), orc.chain_state(34)
    tracked = np.all(np.abs(ug - uo) < 2e-3, axis=1) & same0
    assert tracked.sum() / max(same0.sum(), 1) > 0.95, tracked.sum() / max(same0.sum(), 1)
    rg, ro = ctx.stats().ratios(), orc.stats().ratios()
    for k in ("first", "bold", "second", "overall"):
        assert abs(rg[k] - ro[k]) < 0.02, (k, rg[k], ro[k])
    assert lum(ctx.film()).sum() == pytest.approx(lum(orc.film()).sum(), rel=5e-3)


def test_acceptance_map(pkg, ob, native_lib):
    sd = pkg.scenes.cornell_c2(32)
    n_chains, n_mut = 2048, 32
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", work_units=n_chains, sample_count=1, acceptance_map=1)
    assert ctx.seed(1) == 1.0 and orc.seed(1) == 1.0       # luminance forced to 1 (drmlt.cpp:550-552)
    ctx.run(n_chains * n_mut), orc.run(n_chains * n_mut, 8)
    fg, fo = ctx.develop(), orc.develop()                  # factor 1: raw bins
    sg = ctx.stats()
    w = 0.99998 ** 2                                       # box-filter table weight
    assert fg[..., 2].max() == 0
    assert fg[..., 0].sum() == pytest.approx(sg.bold_acc * w, rel=1e-3)      # red: accepted bold first stages
    assert fg[..., 1].sum() == pytest.approx(sg.second_acc * w, rel=1e-3)    # green: accepted second stages
    assert fg[..., 0].sum() == pytest.approx(fo[..., 0].sum(), rel=0.02)
    assert fg[..., 1].sum() == pytest.approx(fo[..., 1].sum(), rel=0.05)


def test_gaussian_filter_film(pkg, ob, native_lib):
    sd = pkg.scenes.cornell_c2(32, filt=pkg.abi.FILTER_GAUSSIAN)
    n_chains, n_mut = 1024, 32
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", work_units=n_chains, sample_count=1)
    ctx.seed(3), orc.seed(3)
    ctx.run(n_chains * n_mut), orc.run(n_chains * n_mut, 8)
    fg, fo = ctx.film(), orc.film()
    assert lum(fg).sum() == pytest.approx(lum(fo).sum(), rel=5e-3)
    assert np.abs(lum(fg) - lum(fo)).sum() / lum(fo).sum() < 0.08


def test_mlt_image_is_unbiased_against_device_path_tracing(pkg, ob, native_lib):
    sd = pkg.scenes.cornell_c2(32)
    spp = 2048
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", work_units=4096, sample_count=spp, luminance_samples=200000)
    ref = ctx.render_pt(8192, seed=5)
    ref_cpu = orc.render_pt(256, seed=5, nthreads=8)
    assert lum(ref).mean() == pytest.approx(lum(ref_cpu).mean(), rel=0.02)   # same integrand on both sides
    b = ctx.seed(9)
    assert b == pytest.approx(lum(ref).mean(), rel=0.02)
    ctx.run(32 * 32 * spp)
    img = ctx.develop()
    assert lum(img).mean() == pytest.approx(b, rel=1e-3)
    # equal-budget protocol: the device image is as close to the reference as the oracle's
    orc.seed(9)
    orc.run(32 * 32 * spp, 16)
    e_gpu, e_cpu = rel_mse(img, ref), rel_mse(orc.develop(), ref)
    assert abs(e_gpu - e_cpu) < 0.10 * e_cpu, (e_gpu, e_cpu)   # within 10 % at equal budget
    assert e_gpu < 1e-2


def test_config2_full_size_invariants(pkg, ob, native_lib):
    """BASELINE config 2 at full size: size-independent properties only."""
    sd = pkg.scenes.cornell_c2(512)
    n_chains = 65536
    cfg = pkg.abi.make_config(type="orbital", max_depth=8, rr_depth=5, direct_samples=-1, work_units=n_chains,
                              luminance_samples=655360, sample_count=256)
    ctx = pkg.Context(cfg, sd)
    b = ctx.seed(0x5EED)
    total = 512 * 512 * 64
    ctx.run(total)
    st = ctx.stats()
    M = st.mutations
    assert M == total
    assert st.first_base == M and st.large_base + st.bold_base == M
    assert st.second_base == st.bold_base - st.bold_acc and st.second_large_base == 0
    assert st.overall_base == M + st.second_base and st.overall_acc == st.first_acc + st.second_acc == st.accepted
    assert st.path_evals == M + st.second_base
    assert abs(st.large_base / M - 0.3) < 2e-3
    film = ctx.film()
    assert np.all(np.isfinite(film)) and film.min() >= 0
    # every mutation deposits unit luminance (three expectation weights sum to one)
    assert lum(film).sum() == pytest.approx(M * 0.99998 ** 2, rel=2e-3)
    cur, u = ctx.chain_state(34)
    assert np.all((u >= 0) & (u <= 1)) and np.all(cur["luminance"] > 0)
    img = ctx.develop()
    assert lum(img).mean() == pytest.approx(b, rel=1e-3)
    direct = np.full_like(img, 0.25)
    assert np.allclose(ctx.develop(direct), img + 0.25, atol=1e-5)          # develop adds the direct image


def test_call_order_and_cancellation(pkg, ob, native_lib):
    abi = pkg.abi
    sd = pkg.scenes.cornell_c1(16)
    cfg = abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=256, sample_count=64)
    ctx = pkg.Context(cfg, sd)
    with pytest.raises(pkg.DrmltError) as e:
        ctx.run(1000)
    assert e.value.code == abi.E_STATE
    ctx.seed(1)
    seen = []
    ctx.run(256 * 600, progress=lambda d, t: seen.append((d, t)))
    assert seen and seen[-1][0] == seen[-1][1] == 256 * 600 and all(a[0] < b[0] for a, b in zip(seen, seen[1:]))
    stop = C.c_int(1)
    with pytest.raises(pkg.DrmltError) as e:
        ctx.run(256 * 600, stop=stop)
    assert e.value.code == abi.E_CANCELLED
    assert ctx.stats().mutations == 256 * 600


def test_black_scene_reports_zero_luminance(pkg, ob, native_lib):
    sc = pkg.scenes
    sd = sc.SceneData("dark")
    grey, black = sd.diffuse(0.5), sd.diffuse(0.0)
    sd.rectangle(sc.translate(0, -1, 0) @ sc.rotate("x", -90), grey)
    sd.rectangle(sc.translate(0, 1.5, 0) @ sc.rotate("x", -90) @ sc.scale(0.25), black, radiance=10.0)  # faces away
    sd.set_camera(sc.lookat((0, 0, 3.9), (0, 0, 0), (0, 1, 0)), 39.0, 16, 16)
    ctx = pkg.Context(pkg.abi.make_config(type="orbital", max_depth=4, direct_samples=-1, work_units=64), sd)
    with pytest.raises(pkg.DrmltError) as e:
        ctx.seed(1)
    assert e.value.code == pkg.abi.E_ZERO_LUM and "luminance appears to be zero" in str(e.value)


def _ctx_with_env(pkg, cfg, sd, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return pkg.Context(cfg, sd)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("kw", [dict(type="orbital"), dict(type="green"), dict(type="mira"),
                                dict(type="orbital", use_mixture=1), dict(type="green", timid_after_large=1)],
                         ids=lambda k: "-".join("%s=%s" % i for i in k.items()))
def test_chain_kernel_generations_run_the_same_chains(pkg, ob, kw, native_lib):
    """k_mutate_v4 (default) against its predecessor k_mutate_v3, kept as the bit-equality cross-check, at several
    bookkeeping batch sizes: same addressed draws, same arithmetic per chain -> identical states, statistics, films."""
    sd = pkg.scenes.cornell_c2(32)
    n_chains, n_mut = 1024, 40
    cfg = pkg.abi.make_config(max_depth=8, direct_samples=-1, luminance_samples=20000, work_units=n_chains,
                              sample_count=1, **kw)
    results = []
    for env in (dict(DRMLT_KERNEL=3, DRMLT_MH_BATCH=12), dict(DRMLT_KERNEL=4, DRMLT_MH_BATCH=1), dict(DRMLT_KERNEL=4, DRMLT_MH_BATCH=12),
                dict(DRMLT_KERNEL=4, DRMLT_MH_BATCH=32)):
        ctx = _ctx_with_env(pkg, cfg, sd, **env)
        ctx.seed(0x77)
        ctx.run(n_chains * n_mut)
        results.append((ctx.chain_state(34), ctx.stats(), ctx.film()))
    (c0, u0), s0, f0 = results[0]
    for (c, u), s, f in results[1:]:
        # states AND luminances are bit-equal: kernels.hip is compiled with -ffp-contract=on (contraction as the source is written,
        # not as each kernel's inlined copy of the path step happens to be optimised) -- round 3 had relaxed this to 2e-6
        assert np.array_equal(u, u0) and np.array_equal(c["luminance"], c0["luminance"])
        for k in ("first", "large", "bold", "second", "second_large", "second_bold", "overall"):
            assert getattr(s, k + "_base") == getattr(s0, k + "_base") and getattr(s, k + "_acc") == getattr(s0, k + "_acc")
        assert s.rays == s0.rays and s.path_evals == s0.path_evals and s.accepted == s0.accepted
        # v4 splats the current state once per residence with its summed weight, v3 once per mutation: same sum, other association
        assert lum(f).sum() == pytest.approx(lum(f0).sum(), rel=1e-5)
        assert np.abs(lum(f) - lum(f0)).sum() / lum(f0).sum() < 1e-4


def test_large_scene_short_stack_column_spills_and_refills(pkg, ob, native_lib, capfd):
    """40 000 triangles: leaf references need 32 bits, the SAH tree is 11+ four-wide levels deep. k_mutate_v4 keeps 12
    stack entries per lane in LDS and spills the rest to memory; k_mutate_v3 traverses with the full 24-entry column (and
    the same overflow area). Same chains, bit for bit -- and f(u) through the BVH equals f(u) through the brute-force loop."""
    import re
    sd = pkg.scenes.triangle_soup(40000, 32)
    n_chains, n_mut = 2048, 12
    cfg = pkg.abi.make_config(type="orbital", max_depth=8, direct_samples=-1, luminance_samples=20000, work_units=n_chains, sample_count=1)
    os.environ["DRMLT_VERBOSE"] = "1"
    try:
        res = []
        for env in (dict(DRMLT_KERNEL=4), dict(DRMLT_KERNEL=3)):
            ctx = _ctx_with_env(pkg, cfg, sd, **env)
            ctx.seed(0x99)
            ctx.run(n_chains * n_mut)
            res.append((ctx.chain_state(34), ctx.stats(), ctx.film()))
        log = capfd.readouterr().err
    finally:
        del os.environ["DRMLT_VERBOSE"]
    m = re.search(r"4-wide depth (\d+) \(stack 24 in LDS \+ (\d+) in memory\), (\d+) median splits, 32-bit stack entries", log)
    assert m and int(m.group(1)) > 8 and int(m.group(2)) > 0 and int(m.group(3)) == 0, log
    ((c4, u4), s4, f4), ((c3, u3), s3, f3) = res
    assert np.array_equal(u4, u3) and s4.accepted == s3.accepted and s4.rays == s3.rays
    assert lum(f4).sum() == pytest.approx(lum(f3).sum(), rel=1e-5)
    u = np.random.default_rng(3).random((4096, 50), dtype=np.float32)
    b = pkg.Context(cfg, sd).eval_paths(u)
    os.environ["DRMLT_BVH_THRESHOLD"] = "1000000"
    os.environ["DRMLT_NO_BOX_MERGE"] = "1"
    try:
        a = pkg.Context(cfg, sd).eval_paths(u)
    finally:
        del os.environ["DRMLT_BVH_THRESHOLD"], os.environ["DRMLT_NO_BOX_MERGE"]
    same = a["n_dims"] == b["n_dims"]
    assert same.mean() > 0.995
    assert np.allclose(a["luminance"][same], b["luminance"][same], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("scene,kw", [("cornell_c2", dict(type="orbital")), ("door_c3", dict(type="green")), ("glass_sphere", dict(type="mira"))],
                         ids=["c2-orbital", "door-green", "glass-mira"])
def test_run_ahead_between_launches_changes_no_chain(pkg, ob, scene, kw, native_lib):
    """drmlt_run cuts a call into launches of DRMLT_SLICE mutations per chain; between them chains that have reached a launch's
    target run ahead (per-chain mutation counters) instead of idling until the slowest chain of the grid is there. Every
    chain still runs exactly its count, with the random numbers of its own mutation indices: states, statistics and film
    equal those of fixed-count launches (and of k_mutate_v3)."""
    sd = pkg.scenes.SCENES[scene](res=32)
    n_chains, per_chain = 2048, 700                     # slice 128: 6 launches, the last one short
    cfg = pkg.abi.make_config(max_depth=8, direct_samples=-1, luminance_samples=20000, work_units=n_chains, sample_count=1, **kw)
    res = []
    for env in (dict(DRMLT_SLICE=128), dict(DRMLT_SLICE=128, DRMLT_NO_RUN_AHEAD=1), dict(DRMLT_SLICE=128, DRMLT_KERNEL=3)):
        ctx = _ctx_with_env(pkg, cfg, sd, **env)
        ctx.seed(0x1234)
        for k in env:
            os.environ[k] = str(env[k])                 # the slice and the switch are read by drmlt_run as well
        try:
            ctx.run(n_chains * per_chain)
            ctx.run(n_chains * 40)                      # a second, single-launch call continues from the counters
        finally:
            for k in env:
                del os.environ[k]
        res.append((ctx.chain_state(34), ctx.stats(), ctx.film()))
    (c0, u0), s0, f0 = res[0]
    assert np.array_equal(res[1][0][0]["luminance"], c0["luminance"])          # with / without run-ahead: the same kernel, bit for bit
    for (c, u), s, f in res[1:]:
        assert np.array_equal(u, u0) and np.array_equal(c["luminance"], c0["luminance"])   # (v3 included: -ffp-contract=on)
        assert s.mutations == s0.mutations == n_chains * (per_chain + 40)
        assert s.accepted == s0.accepted and s.rays == s0.rays and s.first_acc == s0.first_acc and s.second_base == s0.second_base
        assert lum(f).sum() == pytest.approx(lum(f0).sum(), rel=1e-5)
        assert np.abs(lum(f) - lum(f0)).sum() / lum(f0).sum() < 1e-3
    assert res[0][1].launches == 7


@pytest.mark.parametrize("scene,kw", [("caustic_c5", dict(type="orbital")), ("glass_sphere", dict(type="mira")), ("door_c3", dict(type="green"))],
                         ids=["caustic-spheres", "glass-sphere", "door-conductor"])
def test_forced_bvh_on_scenes_with_spheres_and_glossy_surfaces(pkg, ob, scene, kw, native_lib):
    """The general build of the traversal (k_mutate_v4<15>: leaves may hold spheres, surfaces may be dielectric or rough
    conductors) on scenes small enough for the brute-force loop: with the BVH forced on, f(u) equals the brute-force
    loop's, and the free-running kernel runs the chains of the lock-step one."""
    sd = pkg.scenes.SCENES[scene](res=32)
    n_chains = 2048
    cfg = pkg.abi.make_config(max_depth=8, direct_samples=-1, luminance_samples=20000, work_units=n_chains, sample_count=1, **kw)
    u = np.random.default_rng(11).random((8192, 50), dtype=np.float32)
    brute = _ctx_with_env(pkg, cfg, sd, DRMLT_BVH_THRESHOLD=1000000, DRMLT_NO_BOX_MERGE=1).eval_paths(u)
    bvh = _ctx_with_env(pkg, cfg, sd, DRMLT_BVH_THRESHOLD=0).eval_paths(u)
    same = brute["n_dims"] == bvh["n_dims"]
    assert same.mean() > 0.995
    assert np.allclose(brute["luminance"][same], bvh["luminance"][same], rtol=1e-4, atol=1e-6)
    res = []
    for env in (dict(DRMLT_BVH_THRESHOLD=0), dict(DRMLT_BVH_THRESHOLD=0, DRMLT_KERNEL=3)):
        ctx = _ctx_with_env(pkg, cfg, sd, **env)
        ctx.seed(0x4242)
        ctx.run(n_chains * 120)
        res.append((ctx.chain_state(34), ctx.stats(), ctx.film()))
    ((c4, u4), s4, f4), ((c3, u3), s3, f3) = res
    assert s4.bvh_node_visits > 0 and s4.bvh_prim_tests > 0
    assert np.array_equal(u4, u3) and s4.accepted == s3.accepted and s4.rays == s3.rays
    assert lum(f4).sum() == pytest.approx(lum(f3).sum(), rel=1e-5)
