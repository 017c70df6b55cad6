"""Edge cases through the C-ABI: ragged chain counts (last wave partly empty), a single chain, non-square and one-pixel
films, the shortest paths, no roulette dimensions, pLarge at both ends, more chains than luminance samples asked for."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LUMW = np.array([0.212671, 0.715160, 0.072169])


def lum(img):
    return img @ LUMW


def rect_scene(pkg, w, h):
    sc = pkg.scenes
    sd = sc.cornell_c2(64)
    cam = sd.camera
    sd.set_camera(np.array(cam.to_world[:]).reshape(4, 4), cam.fov_x_deg, w, h)
    return sd


@pytest.mark.parametrize("tech", ["path", "mmlt", "bdpt"])
@pytest.mark.parametrize("n_chains", [1, 63, 65, 100])
def test_ragged_chain_counts_track_the_oracle(pkg, ob, tech, n_chains, native_lib):
    sd = pkg.scenes.cornell_c2(16)
    kw = dict(technique=tech, type="orbital", max_depth=5, direct_samples=-1, work_units=n_chains, sample_count=1,
              luminance_samples=3000)
    cfg = pkg.abi.make_config(**kw)
    ctx, orc = pkg.Context(cfg, sd), ob.Oracle(pkg.abi, cfg, sd, 64)
    bg, bo = ctx.seed(42), orc.seed(42)
    assert bg == pytest.approx(bo, rel=2e-3)
    ctx.run(n_chains * 24); orc.run(n_chains * 24, 4)
    sg, so = ctx.stats(), orc.stats()
    assert sg.mutations == so.mutations == n_chains * 24 and sg.n_chains == n_chains
    fg, fo = ctx.film(), orc.film()
    assert np.isfinite(fg).all() and lum(fg).sum() == pytest.approx(lum(fo).sum(), rel=2e-2)
    dim = ctx.stats().max_dim
    (cg, ug), (co, uo) = ctx.chain_state(dim), orc.chain_state(dim)
    same = np.abs(cg["luminance"] - co["luminance"]) <= 5e-3 * co["luminance"]
    assert same.mean() > 0.6          # short run: most chains still on the oracle's trajectory
    assert ((ug >= 0) & (ug <= 1)).all()


@pytest.mark.parametrize("wh", [(96, 32), (20, 50), (1, 1)])
def test_non_square_and_tiny_films(pkg, ob, wh, native_lib):
    w, h = wh
    sd = rect_scene(pkg, w, h)
    cfg = pkg.abi.make_config(type="orbital", max_depth=6, direct_samples=-1, work_units=256, sample_count=64,
                              luminance_samples=20000)
    ctx, orc = pkg.Context(cfg, sd), ob.Oracle(pkg.abi, cfg, sd, 64)
    rng = np.random.default_rng(3)
    u = rng.random((4096, 50), dtype=np.float32)
    g, o = ctx.eval_paths(u), orc.eval_paths(u)
    same = g["n_dims"] == o["n_dims"]
    assert same.mean() > 0.995
    assert np.allclose(g["x"], o["x"], atol=1e-3) and np.allclose(g["y"], o["y"], atol=1e-3)
    assert g["x"].max() <= w and g["y"].max() <= h
    assert g["luminance"].mean() == pytest.approx(o["luminance"].mean(), rel=5e-3)
    b = ctx.seed(5)
    ctx.run(max(w * h * 64, 256 * 8))
    img = ctx.develop()
    assert img.shape == (h, w, 3) and np.isfinite(img).all()
    assert lum(img.astype(np.float64)).mean() == pytest.approx(b, rel=2e-3)
    # mmlt on the same film: light-tracing splats use the film's aspect as well
    cfgm = pkg.abi.make_config(technique="mmlt", type="orbital", max_depth=4, direct_samples=-1, work_units=256)
    ctxm, orcm = pkg.Context(cfgm, sd), ob.Oracle(pkg.abi, cfgm, sd, 64)
    us, ue, ud = rng.random((2048, 10), dtype=np.float32), rng.random((2048, 10), dtype=np.float32), rng.random(2048, dtype=np.float32)
    for depth in (2, 3):
        gm, _ = ctxm.eval_paths_mmlt(depth, us, ue, ud)
        om, _ = orcm.mmlt_eval(depth, us, ue, ud)
        pos = (gm["luminance"] > 0) & (om["luminance"] > 0)
        assert ((gm["luminance"] > 0) == (om["luminance"] > 0)).mean() > 0.995
        assert np.allclose(gm["x"][pos], om["x"][pos], atol=2e-2) and np.allclose(gm["y"][pos], om["y"][pos], atol=2e-2)
        assert gm["luminance"].sum() == pytest.approx(om["luminance"].sum(), rel=5e-3)


@pytest.mark.parametrize("kw", [dict(max_depth=2, rr_depth=5), dict(max_depth=3, rr_depth=3), dict(max_depth=8, rr_depth=8),
                                dict(max_depth=8, rr_depth=1), dict(max_depth=6, p_large=0.0), dict(max_depth=6, p_large=1.0)],
                         ids=lambda k: "-".join("%s=%s" % i for i in k.items()))
def test_depth_roulette_and_plarge_extremes(pkg, ob, kw, native_lib):
    sd = pkg.scenes.cornell_c2(16)
    base = dict(type="orbital", direct_samples=-1, work_units=1024, sample_count=1, luminance_samples=20000)
    base.update(kw)
    cfg = pkg.abi.make_config(**base)
    ctx, orc = pkg.Context(cfg, sd), ob.Oracle(pkg.abi, cfg, sd, 64)
    assert ctx.stats().max_dim == orc.stats().max_dim
    bg, bo = ctx.seed(9), orc.seed(9)
    assert bg == pytest.approx(bo, rel=1e-3)
    ctx.run(1024 * 32); orc.run(1024 * 32, 8)
    sg, so = ctx.stats(), orc.stats()
    assert sg.large_base == so.large_base                       # the large-step coin is the same addressed draw
    if kw.get("p_large") == 1.0:
        assert sg.large_base == sg.mutations and sg.second_base == 0
    if kw.get("p_large") == 0.0:
        assert sg.large_base == 0
    assert abs(sg.first_acc - so.first_acc) <= 0.02 * so.first_base + 20
    assert lum(ctx.film()).sum() == pytest.approx(lum(orc.film()).sum(), rel=5e-3)


def test_more_chains_than_requested_luminance_samples(pkg, native_lib):
    sd = pkg.scenes.cornell_c2(16)
    cfg = pkg.abi.make_config(type="orbital", max_depth=5, direct_samples=-1, work_units=5000, sample_count=1, luminance_samples=10)
    ctx = pkg.Context(cfg, sd)
    b = ctx.seed(1)                                             # floor: 10 samples per chain (drmlt.cpp:454-466)
    assert b > 0
    ctx.run(5000 * 4)
    assert ctx.stats().mutations == 20000


def test_derived_work_units_fill_the_device(pkg, native_lib):
    """workUnits = -1 (the reference default): a device-filling chain count, not the reference's few hundred CPU work units."""
    sd = pkg.scenes.cornell_c2(64)
    for tech, typ, want in (("path", "orbital", 196608), ("path", "green", 196608), ("path", "mira", 196608), ("mmlt", "orbital", 262144), ("bdpt", "orbital", 131072)):
        # (path: 64 chains per wave of the ray-pool kernel, three waves per SIMD with its proposal rows in device memory; mmlt: two rounds of waves)
        ctx = pkg.Context(pkg.abi.make_config(technique=tech, type=typ, max_depth=5, work_units=-1, sample_count=16384), sd)
        assert ctx.stats().n_chains == want
        ctx.close()
    # mmlt in a long render (2^35 mutations and more: a 2048^2 film at 8192 mutations per pixel): eight rounds of waves -- the bootstrap set
    # that seeds a million chains costs seconds, the shorter launch tails are worth 14 % of the rest
    big = pkg.scenes.cornell_c2(2048)
    ctx = pkg.Context(pkg.abi.make_config(technique="mmlt", type="orbital", max_depth=5, work_units=-1, sample_count=8192), big)
    assert ctx.stats().n_chains == 1048576
    ctx.close()
    ctx = pkg.Context(pkg.abi.make_config(technique="path", type="orbital", max_depth=5, work_units=-1, sample_count=16), sd)
    assert ctx.stats().n_chains == 64 * 64 * 16 // 64            # never chains shorter than 64 mutations
    ctx.close()
