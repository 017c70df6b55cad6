"""GPU parity for technique=bdpt, directSampling=false and true (the reference's default): through
the C-ABI, against the oracle's restatement of PathSampler::sampleSplats(EBidirectional) and of the chain loop over
multi-splat lists."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LUMW = np.array([0.212671, 0.715160, 0.072169])


def lum(img):
    return img @ LUMW


def make(pkg, ob, sd, **kw):
    abi = pkg.abi
    base = dict(technique="bdpt", max_depth=6, rr_depth=4, direct_samples=-1, no_direct_sampling=1, luminance_samples=20000)
    base.update(kw)
    cfg = abi.make_config(**base)
    return cfg, pkg.Context(cfg, sd), ob.Oracle(abi, cfg, sd, 64)


@pytest.mark.parametrize("direct", [0, 1], ids=["nodirect", "direct"])
@pytest.mark.parametrize("name", ["cornell_c2", "glass_sphere", "door_c3", "caustic_c5"])
def test_lists_match_oracle(pkg, ob, name, direct, native_lib):
    """f(u) = a splat list: same number of light-image splats, same dims / rays, luminance and every splat within 2e-3.
    direct: the s = 1 / t = 1 strategies by direct sampling and the sampleDirect terms of miWeight."""
    sd = pkg.scenes.SCENES[name](res=64)
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", work_units=1024, no_direct_sampling=0 if direct else 1)
    rng = np.random.default_rng(11)
    n = 6000
    us, ue, ud = (rng.random((n, 24), dtype=np.float32) for _ in range(3))
    g, o = (ctx.eval_lists_bdpt(us, ue, ud), orc.bdpt_eval(us, ue, ud)) if direct else (ctx.eval_lists_bdpt(us, ue), orc.bdpt_eval(us, ue))
    same = (g[:, 1] == o[:, 1]) & (g[:, 7] == o[:, 7]) & (g[:, 8] == o[:, 8]) & (g[:, 9] == o[:, 9])
    assert same.mean() > 0.99, same.mean()
    rel = np.abs(g[:, 0] - o[:, 0])[same] / np.maximum(o[:, 0][same], 1e-3)
    assert np.quantile(rel, 0.99) < 2e-3, np.quantile(rel, 0.99)
    assert g[:, 0].sum() == pytest.approx(o[:, 0].sum(), rel=2e-3)
    ok = same & (rel_full(g, o) < 1e-2)
    assert ok.mean() > 0.985
    np.testing.assert_allclose(g[ok][:, 2:4], o[ok][:, 2:4], atol=1e-3)                 # main splat position
    np.testing.assert_allclose(g[ok][:, 4:7], o[ok][:, 4:7], rtol=2e-2, atol=3e-4)      # main splat value (one connection of 18 000 sees its ray change sides of an edge in fp32)
    mg, mo = g[ok][:, 10:].reshape(ok.sum(), -1, 5), o[ok][:, 10:].reshape(ok.sum(), -1, 5)
    np.testing.assert_allclose(mg[:, :, :2], mo[:, :, :2], atol=3e-2)                   # light-image positions (fp32 light paths)
    np.testing.assert_allclose(mg[:, :, 2:], mo[:, :, 2:], rtol=3e-2, atol=3e-4)


@pytest.mark.parametrize("direct", [0, 1], ids=["nodirect", "direct"])
def test_lists_of_deep_paths(pkg, ob, direct, native_lib):
    """maxDepth 12 without russian roulette in a closed box: both walks run to full length, a chain has up to 90 (s, t) cells --
    more than the 64 lanes of a connection round, so it gets rounds of its own, 64 cells at a time (device_bdpt.h) -- and 46
    components of the direct sampler. Same lists as the oracle."""
    sd = pkg.scenes.cornell_c2(64)
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", work_units=256, max_depth=12, rr_depth=-1, no_direct_sampling=0 if direct else 1)
    rng = np.random.default_rng(3)
    n, w = 3000, 48
    us, ue, ud = (rng.random((n, w), dtype=np.float32) for _ in range(3))
    g, o = (ctx.eval_lists_bdpt(us, ue, ud), orc.bdpt_eval(us, ue, ud)) if direct else (ctx.eval_lists_bdpt(us, ue), orc.bdpt_eval(us, ue))
    same = (g[:, 1] == o[:, 1]) & (g[:, 7] == o[:, 7]) & (g[:, 8] == o[:, 8]) & (g[:, 9] == o[:, 9])
    assert same.mean() > 0.98, same.mean()
    deep = o[:, 9] >= 80                                                        # rays: the walks and one per connected cell
    assert deep.sum() >= 30, (deep.sum(), np.quantile(o[:, 9], [0.5, 0.9, 0.99]), o[:, 9].max())
    assert same[deep].mean() > 0.95
    rel = np.abs(g[:, 0] - o[:, 0])[same] / np.maximum(o[:, 0][same], 1e-3)
    assert np.quantile(rel, 0.99) < 3e-3, np.quantile(rel, 0.99)
    assert g[:, 0].sum() == pytest.approx(o[:, 0].sum(), rel=2e-3)
    mg, mo = g[same][:, 10:].reshape(same.sum(), -1, 5), o[same][:, 10:].reshape(same.sum(), -1, 5)
    np.testing.assert_allclose(mg[:, :, :2], mo[:, :, :2], atol=5e-2)
    np.testing.assert_allclose(mg[:, :, 2:], mo[:, :, 2:], rtol=5e-2, atol=5e-4)


@pytest.mark.parametrize("direct", [0, 1], ids=["nodirect", "direct"])
def test_lists_at_max_depth_20(pkg, ob, direct, native_lib):
    """maxDepth 20 (round 3 refused more than 15: two flag bits per stored vertex in one 64-bit word; now one bit per vertex in
    each of two words, up to maxDepth 24): 41 stored vertices, 60 KB of sampler and density rows per wave. Same lists as the oracle,
    and a short chain run stays finite."""
    sd = pkg.scenes.cornell_c2(64)
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", work_units=256, max_depth=20, rr_depth=-1, no_direct_sampling=0 if direct else 1)
    rng = np.random.default_rng(4)
    n, w = 1500, 80
    us, ue, ud = (rng.random((n, w), dtype=np.float32) for _ in range(3))
    g, o = (ctx.eval_lists_bdpt(us, ue, ud), orc.bdpt_eval(us, ue, ud)) if direct else (ctx.eval_lists_bdpt(us, ue), orc.bdpt_eval(us, ue))
    same = (g[:, 1] == o[:, 1]) & (g[:, 7] == o[:, 7]) & (g[:, 8] == o[:, 8]) & (g[:, 9] == o[:, 9])
    assert same.mean() > 0.97, same.mean()
    assert (o[:, 9] >= 150).sum() >= 20                                         # paths that do use the depth
    rel = np.abs(g[:, 0] - o[:, 0])[same] / np.maximum(o[:, 0][same], 1e-3)
    assert np.quantile(rel, 0.99) < 4e-3, np.quantile(rel, 0.99)
    assert g[:, 0].sum() == pytest.approx(o[:, 0].sum(), rel=3e-3)
    b = ctx.seed(5)
    ctx.run(256 * 32)
    st = ctx.stats()
    img = ctx.develop()
    assert st.mutations == 256 * 32 and np.isfinite(img).all() and lum(img).mean() == pytest.approx(b, rel=1e-3)


def rel_full(g, o):
    return np.abs(g[:, 0] - o[:, 0]) / np.maximum(o[:, 0], 1e-3)


VARIANTS = [dict(type="orbital"), dict(type="green"), dict(type="mira"), dict(type="orbital", use_mixture=1),
            dict(type="orbital", no_light_image=1), dict(type="green", direct_samples=16),
            dict(type="orbital", no_direct_sampling=0), dict(type="green", no_direct_sampling=0), dict(type="mira", no_direct_sampling=0),
            dict(type="orbital", no_direct_sampling=0, use_mixture=1), dict(type="orbital", no_direct_sampling=0, no_light_image=1),
            # timidAfterLarge under bdpt (round 3 refused it; the reference gates the second stage on it for every technique,
            # drmlt_proc.cpp:553-558): a rejected large step's second stage is another uniform proposal of all three samplers
            dict(type="orbital", no_direct_sampling=0, timid_after_large=1), dict(type="green", no_direct_sampling=0, timid_after_large=1),
            dict(type="mira", timid_after_large=1)]


@pytest.mark.parametrize("kw", VARIANTS, ids=lambda k: "-".join("%s=%s" % i for i in k.items()))
def test_chains_track_the_oracle(pkg, ob, kw, native_lib):
    sd = pkg.scenes.glass_sphere(32)
    n_chains, n_mut = 2048, 32
    cfg, ctx, orc = make(pkg, ob, sd, work_units=n_chains, sample_count=1, **kw)
    bg, bo = ctx.seed(0xABCD), orc.seed(0xABCD)
    assert bg == pytest.approx(bo, rel=1e-3)
    dim = ctx.stats().max_dim
    (c0g, u0g), (c0o, u0o) = ctx.chain_state(dim), orc.chain_state(dim)
    same0 = np.abs(c0g["luminance"] - c0o["luminance"]) <= 1e-3 * c0o["luminance"]
    assert same0.mean() > 0.5
    ctx.run(n_chains * n_mut); orc.run(n_chains * n_mut, 8)
    (cg, ug), (co, uo) = ctx.chain_state(dim), orc.chain_state(dim)
    tracked = same0 & (np.abs(cg["luminance"] - co["luminance"]) <= 3e-3 * co["luminance"]) & (cg["n_rays"] == co["n_rays"])
    assert tracked.sum() / same0.sum() > 0.93, tracked.sum() / same0.sum()
    sg, so = ctx.stats(), orc.stats()
    assert sg.mutations == so.mutations == n_chains * n_mut
    if kw.get("timid_after_large"):
        assert sg.second_large_base > 0.1 * sg.large_base and so.second_large_base > 0          # second stages after rejected large steps do happen
        assert abs(sg.second_large_base - so.second_large_base) <= 0.03 * so.second_large_base + 20
    for k in ("first", "large", "bold", "second", "overall"):
        bg_, bo_ = getattr(sg, k + "_base"), getattr(so, k + "_base")
        assert abs(bg_ - bo_) <= 0.02 * max(bo_, 1) + 20, (k, bg_, bo_)
        if bo_ > 200:
            pg, po = getattr(sg, k + "_acc") / bg_, getattr(so, k + "_acc") / bo_
            assert abs(pg - po) < 4 * np.sqrt(po * (1 - po) / bo_) + 0.015, (k, pg, po)
    assert abs(sg.rays - so.rays) <= 0.03 * so.rays
    fg, fo = ctx.film(), orc.film()
    assert lum(fg).sum() == pytest.approx(lum(fo).sum(), rel=5e-3)
    bgk, bok = (lum(f).reshape(8, 4, 8, 4).sum(axis=(1, 3)) for f in (fg, fo))
    assert np.abs(bgk - bok).sum() / bok.sum() < 0.1


def test_bdpt_image_and_acceptance_map(pkg, ob, native_lib):
    sd = pkg.scenes.glass_sphere(32)
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", work_units=8192, sample_count=512, luminance_samples=200000)
    ref = orc.bdpt_render(32 * 32 * 3000, seed=9, nthreads=8)
    b = ctx.seed(0x5EED)
    ctx.run(32 * 32 * 512)
    img = ctx.develop()
    assert b == pytest.approx(lum(ref).mean(), rel=0.02)
    blk = lambda a: a.reshape(8, 4, 8, 4, 3).mean((1, 3))
    assert np.abs(blk(img) - blk(ref)).mean() / ref.mean() < 0.04
    # README's acceptance-map example uses technique=bdpt: every splat of an accepted list marks its pixel
    cfg2, ctx2, orc2 = make(pkg, ob, sd, type="orbital", work_units=4096, sample_count=1, acceptance_map=1)
    assert ctx2.seed(3) == 1.0
    orc2.seed(3)
    ctx2.run(4096 * 32); orc2.run(4096 * 32, 8)
    fg, fo = ctx2.film().astype(np.float64), orc2.film().astype(np.float64)
    assert fg[..., 2].max() == 0 and fg[..., 0].sum() > 0 and fg[..., 1].sum() > 0
    assert fg[..., 0].sum() == pytest.approx(fo[..., 0].sum(), rel=0.03)
    assert fg[..., 1].sum() == pytest.approx(fo[..., 1].sum(), rel=0.05)


def test_direct_sampling_image_matches_the_plain_estimator(pkg, ob, native_lib):
    """directSampling=true (the reference default) is another weighting of the same integrand: chains with it converge to the
    image of the oracle's independent bdpt samples drawn WITHOUT it, and their b agrees."""
    sd = pkg.scenes.cornell_c2(32)
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", work_units=8192, sample_count=512, luminance_samples=200000, no_direct_sampling=0)
    _, _, plain = make(pkg, ob, sd, type="orbital", work_units=4, no_direct_sampling=1)
    ref = plain.bdpt_render(32 * 32 * 3000, seed=9, nthreads=8)
    b = ctx.seed(0x5EED)
    assert b == pytest.approx(lum(ref).mean(), rel=0.02)
    ctx.run(32 * 32 * 512)
    img = ctx.develop()
    blk = lambda a: a.reshape(8, 4, 8, 4, 3).mean((1, 3))
    assert np.abs(blk(img) - blk(ref)).mean() / ref.mean() < 0.04


def test_refusals(pkg, native_lib):
    sd = pkg.scenes.cornell_c2(16)
    ctx = pkg.Context(pkg.abi.make_config(technique="bdpt", max_depth=5, work_units=64), sd)   # directSampling=true: the default
    assert ctx.stats().max_dim == 2 * 14 + 2 * (2 * 5 - 1)
    with pytest.raises(ValueError, match="direct sampler"):
        ctx.eval_lists_bdpt(np.zeros((4, 24), dtype=np.float32), np.zeros((4, 24), dtype=np.float32))
    ctx = pkg.Context(pkg.abi.make_config(technique="bdpt", max_depth=5, work_units=64, no_direct_sampling=1), sd)
    with pytest.raises(pkg.DrmltError, match="drmlt_eval_lists"):
        ctx.eval_paths(np.zeros((4, 64), dtype=np.float32))


@pytest.mark.parametrize("tech", ["bdpt", "mmlt"])
def test_bvh_builds_of_the_bidirectional_kernels_run_the_same_chains(pkg, tech, native_lib, monkeypatch, capfd):
    """Scenes above the BVH threshold run other builds of the bidirectional kernels (traversal with its LDS stack instead of
    the brute-force loop -- in bdpt's connection phase under per-cell control flow): forcing the BVH onto a small scene must
    leave every chain where the flat build puts it, bit for bit (the same intersection routines find the same hits)."""
    sd = pkg.scenes.glass_sphere(32)
    n_chains, n_mut = 1000, 40
    kw = dict(technique=tech, type="orbital", max_depth=6, rr_depth=4, direct_samples=-1, luminance_samples=20000, work_units=n_chains, sample_count=1)
    if tech == "bdpt":
        kw["no_direct_sampling"] = 0
    cfg = pkg.abi.make_config(**kw)
    res = []
    monkeypatch.setenv("DRMLT_NO_BOX_MERGE", "1")   # the flat side loops over the separate faces, as the leaves do (cuboid records: tests/test_gpu_boxes.py)
    for thr in (None, "0"):
        if thr is None:
            monkeypatch.delenv("DRMLT_BVH_THRESHOLD", raising=False)
        else:
            monkeypatch.setenv("DRMLT_BVH_THRESHOLD", thr)
        monkeypatch.setenv("DRMLT_VERBOSE", "1")
        capfd.readouterr()
        ctx = pkg.Context(cfg, sd)
        built_bvh = "[drmlt] BVH:" in capfd.readouterr().err
        monkeypatch.delenv("DRMLT_VERBOSE")
        ctx.seed(0xB7)
        ctx.run(n_chains * n_mut)
        st = ctx.stats()
        res.append((ctx.chain_state(st.max_dim if tech == "bdpt" else 27), st, ctx.film(), built_bvh))
        ctx.close()
    ((cf, uf), sf, ff, bvh_f), ((cb, ub), sb, fb, bvh_b) = res
    assert not bvh_f and bvh_b
    assert np.array_equal(ub, uf) and np.array_equal(cb["luminance"], cf["luminance"])
    assert sb.accepted == sf.accepted and sb.rays == sf.rays and sb.path_evals == sf.path_evals
    assert lum(fb).sum() == pytest.approx(lum(ff).sum(), rel=1e-5)
