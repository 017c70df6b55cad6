"""GPU parity for two-stage MLT (`twoStage`) and the equal-time mode (`timeout`), through the C-ABI."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LUMW = np.array([0.212671, 0.715160, 0.072169])


def test_luminance_map_matches_oracle(pkg, ob, native_lib):
    rng = np.random.default_rng(5)
    small = rng.random((8, 8, 3)).astype(np.float32) * 3
    a, b = pkg.binding.luminance_map(small, 128, 128), ob.luminance_map(small, 128, 128)
    np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-6)          # fp32 host statement vs fp64 oracle
    a, b = pkg.binding.luminance_map(small, 20, 50), ob.luminance_map(small, 20, 50)
    np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-6)


def _bdpt_dims(max_depth, rr_depth=5, direct_sampling=True):
    """[sensor S | emitter E | direct Dd] of a bdpt chain (device_bdpt.h / binding.eval_lists_bdpt)."""
    rr = max_depth + 1 - max(rr_depth, 0)
    S = 2 * (max_depth + 1) + max(rr, 0); S += S & 1
    E = 2 * max_depth + max(rr - 1, 0); E += E & 1
    return S + E + (2 * (2 * max_depth - 1) if direct_sampling else 0)


RULES = ["target", "reference"]   # drmlt_config.seed_rule: the product's default and pathsampler.cpp:901-905; the oracle follows the same field


@pytest.mark.parametrize("rule", RULES)
@pytest.mark.parametrize("tech", ["path", "mmlt", "bdpt"])
def test_weighted_chains_track_the_oracle(pkg, ob, tech, rule, native_lib):
    sd = pkg.scenes.cornell_c2(32)
    abi = pkg.abi
    ref = pkg.Context(abi.make_config(max_depth=6, rr_depth=100, direct_samples=-1, work_units=64), sd).render_pt(256, seed=3)
    imp = np.maximum(pkg.binding.luminance_map(ref.reshape(4, 8, 4, 8, 3).mean((1, 3)), 32, 32), 1e-3)
    n_chains, n_mut = 2048, 40
    cfg = abi.make_config(technique=tech, type="orbital", max_depth=6, direct_samples=-1, work_units=n_chains,
                          sample_count=1, luminance_samples=20000, seed_rule=rule)
    ctx, orc = pkg.Context(cfg, sd), ob.Oracle(abi, cfg, sd, 64)
    ctx.set_importance_map(imp); orc.set_importance_map(imp)
    bg, bo = ctx.seed(0xABCD), orc.seed(0xABCD)
    assert bg == pytest.approx(bo, rel=1e-3)                    # b is the mean of f under both rules
    # the two sides pick (nearly) the same bootstrap samples: the rule is the same rule on both sides
    ig_, io_ = ctx.seed_indices(), orc.seed_indices()
    assert len(np.intersect1d(ig_, io_)) > 0.5 * len(np.unique(io_))
    dim = 34 if tech == "path" else (27 if tech == "mmlt" else _bdpt_dims(6))
    (c0g, u0g), (c0o, u0o) = ctx.chain_state(dim), orc.chain_state(dim)
    same0 = np.abs(c0g["luminance"] - c0o["luminance"]) <= 1e-3 * c0o["luminance"]   # same seed, same weighted f
    assert same0.mean() > 0.5
    ctx.run(n_chains * n_mut); orc.run(n_chains * n_mut, 8)
    (cg, ug), (co, uo) = ctx.chain_state(dim), orc.chain_state(dim)
    tracked = same0 & (np.abs(cg["luminance"] - co["luminance"]) <= 2e-3 * co["luminance"]) & \
        (np.abs(cg["x"] - co["x"]) < 1e-2) & (np.abs(cg["y"] - co["y"]) < 1e-2)
    assert tracked.sum() / same0.sum() > 0.95, tracked.sum() / same0.sum()
    sg, so = ctx.stats(), orc.stats()
    for k in ("first", "second", "overall"):
        bg_, bo_ = getattr(sg, k + "_base"), getattr(so, k + "_base")
        pg, po = getattr(sg, k + "_acc") / bg_, getattr(so, k + "_acc") / bo_
        assert abs(pg - po) < 4 * np.sqrt(po * (1 - po) / bo_) + 0.015, (k, pg, po)
    ig, io = ctx.develop(), orc.develop()
    assert (ig @ LUMW).mean() == pytest.approx((io @ LUMW).mean(), rel=5e-3)
    blk = lambda a: (a @ LUMW).reshape(8, 4, 8, 4).mean((1, 3))
    assert np.abs(blk(ig) - blk(io)).sum() / blk(io).sum() < 0.1
    with pytest.raises(pkg.DrmltError, match="before drmlt_seed"):
        ctx.set_importance_map(imp)


@pytest.mark.parametrize("rule", RULES)
@pytest.mark.parametrize("scene", ["caustic_c5", "door_c3"])
def test_importance_map_with_unlit_regions_matches_oracle(pkg, ob, scene, rule, native_lib):
    """A first-stage image with black regions gives exact zeros in the map (mltLuminancePass applies no floor,
    util.cpp:190-196): SplatList::normalize divides by them, the list luminance is inf and the proposal is rejected
    (drmlt_proc.cpp:428). No 1e-3 floor here; part of the map is forced to zero so that chains do propose into it."""
    sd = pkg.scenes.SCENES[scene](res=32)
    abi = pkg.abi
    ref = pkg.Context(abi.make_config(max_depth=6, rr_depth=100, direct_samples=-1, work_units=64), sd).render_pt(256, seed=3)
    imp = pkg.binding.luminance_map(ref.reshape(4, 8, 4, 8, 3).mean((1, 3)), 32, 32)
    imp[:, :6] = 0.0                                           # an unlit band that the image plane does cover
    assert (imp == 0).any() and (imp > 0).any()
    n_chains, n_mut = 2048, 40
    cfg = abi.make_config(technique="path", type="orbital", max_depth=6, direct_samples=-1, work_units=n_chains,
                          sample_count=1, luminance_samples=20000, seed_rule=rule)
    ctx, orc = pkg.Context(cfg, sd), ob.Oracle(abi, cfg, sd, 64)
    ctx.set_importance_map(imp); orc.set_importance_map(imp)   # zeros accepted; only NaN / negative values are refused
    bg, bo = ctx.seed(0xABCD), orc.seed(0xABCD)
    assert bg == pytest.approx(bo, rel=1e-3)
    (c0g, _), (c0o, _) = ctx.chain_state(34), orc.chain_state(34)
    # seeds that fall into the zero band have an infinite (or NaN) weighted luminance on both sides and never move. The two
    # seed lists are not index-aligned (fp32 vs fp64 bootstrap CDFs), so the dead chains are compared as a fraction
    dead_g, dead_o = ~np.isfinite(c0g["luminance"]), ~np.isfinite(c0o["luminance"])
    if rule == "target":   # a sample on a zero of the map has no finite weight: it seeds nothing, on either side
        assert not dead_g.any() and not dead_o.any()
    else:                  # the reference's rule seeds from f itself: chains do start in the band, and stay there
        assert dead_g.any() and abs(dead_g.mean() - dead_o.mean()) < 0.03, (dead_g.mean(), dead_o.mean())
    fin = ~dead_g & ~dead_o
    with np.errstate(invalid="ignore"):
        same0 = fin & (np.abs(c0g["luminance"] - c0o["luminance"]) <= 1e-3 * np.where(fin, c0o["luminance"], 1.0))
    assert same0.sum() > 0.4 * fin.sum()
    xg0 = np.stack([c0g["x"], c0g["y"]], 1)
    ctx.run(n_chains * n_mut); orc.run(n_chains * n_mut, 8)
    (cg, _), (co, _) = ctx.chain_state(34), orc.chain_state(34)
    with np.errstate(invalid="ignore"):
        tracked = same0 & (np.abs(cg["luminance"] - co["luminance"]) <= 2e-3 * np.where(fin, co["luminance"], 1.0)) & \
            (np.abs(cg["x"] - co["x"]) < 1e-2) & (np.abs(cg["y"] - co["y"]) < 1e-2)
    assert tracked.sum() / same0.sum() > 0.95, tracked.sum() / same0.sum()
    assert np.array_equal(np.stack([cg["x"], cg["y"]], 1)[dead_g], xg0[dead_g])       # dead chains did not move
    assert (~np.isfinite(cg["luminance"]) == dead_g).all()                             # and no live chain died: proposals into the band are rejected
    sg, so = ctx.stats(), orc.stats()
    for k in ("first", "second", "overall"):
        bg_, bo_ = getattr(sg, k + "_base"), getattr(so, k + "_base")
        pg, po = getattr(sg, k + "_acc") / bg_, getattr(so, k + "_acc") / bo_
        assert abs(pg - po) < 4 * np.sqrt(po * (1 - po) / bo_) + 0.03, (k, pg, po)    # the dead fractions differ slightly, see above
    ig, io = ctx.develop(), orc.develop()
    assert np.isfinite(ig).all() and np.all(ig[:, :6] == 0) and np.all(io[:, :6] == 0)   # develop multiplies the map back
    assert (ig @ LUMW).mean() == pytest.approx((io @ LUMW).mean(), rel=1e-2)
    with pytest.raises(pkg.DrmltError, match="non-negative"):
        bad = imp.copy(); bad[3, 3] = -1.0
        pkg.Context(cfg, sd).set_importance_map(bad)


@pytest.mark.parametrize("rule", RULES)
def test_pssmlt_with_importance_map_uses_veach_weights(pkg, ob, rule, native_lib):
    """pssmlt_proc.cpp:203: with an importance map the accepted-branch weights are Veach's expectations even when
    kelemenStyleWeights is set ("these don't work for 2-stage MLT"); the a <= 0 branch keeps the Kelemen form."""
    sd = pkg.scenes.cornell_c1(32)
    abi = pkg.abi
    ref = pkg.Context(abi.make_config(max_depth=6, rr_depth=100, direct_samples=-1, work_units=64), sd).render_pt(256, seed=3)
    imp = np.maximum(pkg.binding.luminance_map(ref.reshape(4, 8, 4, 8, 3).mean((1, 3)), 32, 32), 1e-3)
    n_chains, n_mut = 2048, 48
    cfg = abi.make_config(algo=abi.ALGO_PSSMLT, technique="path", type="orbital", max_depth=8, rr_depth=5, direct_samples=-1,
                          luminance_samples=20000, work_units=n_chains, sample_count=1, kelemen_style_weights=1, seed_rule=rule)
    ctx, orc = pkg.Context(cfg, sd), ob.Oracle(abi, cfg, sd, 64)
    ctx.set_importance_map(imp); orc.set_importance_map(imp)
    bg, bo = ctx.seed(0xABCD), orc.seed(0xABCD)
    assert bg == pytest.approx(bo, rel=1e-3)
    ctx.run(n_chains * n_mut); orc.run(n_chains * n_mut, 8)
    fg, fo = ctx.film(), orc.film()
    # the Kelemen weights would put b-scaled sums into the film: an order-of-magnitude difference, not a tolerance question
    assert (fg @ LUMW).sum() == pytest.approx((fo @ LUMW).sum(), rel=1e-2)
    blk = lambda a: (a @ LUMW).reshape(8, 4, 8, 4).sum((1, 3))
    assert np.abs(blk(fg) - blk(fo)).sum() / blk(fo).sum() < 0.1
    ig, io = ctx.develop(), orc.develop()
    assert (ig @ LUMW).mean() == pytest.approx((io @ LUMW).mean(), rel=5e-3)


def test_two_stage_render_is_unbiased_and_flatter(pkg, native_lib):
    sd = pkg.scenes.glass_sphere(64)
    abi = pkg.abi
    ref = pkg.Context(abi.make_config(max_depth=6, rr_depth=100, direct_samples=-1, work_units=64), sd).render_pt(4096, seed=3)
    # (chains long enough that the seeding rule does not matter: the next test is about that)
    cfg = abi.make_config(technique="path", type="orbital", max_depth=6, direct_samples=-1, work_units=2048,
                          sample_count=2048, luminance_samples=200000)
    img, imp, b = pkg.binding.render_two_stage(cfg, sd, 0x5EED, size_reduction=16)
    assert imp.shape == (64, 64) and imp.min() > 0
    assert b == pytest.approx((ref @ LUMW).mean(), rel=0.02)
    blk = lambda a: a.reshape(8, 8, 8, 8, 3).mean((1, 3))
    assert np.abs(blk(img) - blk(ref)).mean() / ref.mean() < 0.03
    plain = pkg.Context(cfg, sd)
    plain.seed(0x5EED); plain.run(64 * 64 * 2048)
    ip = plain.develop()
    # relative error in the darkest quarter of the image drops (what two-stage MLT is for, drmlt.cpp:270-277)
    lr = ref @ LUMW
    dark = (lr > 0) & (lr < np.quantile(lr[lr > 0], 0.25))      # directly visible emitter pixels are 0 in this estimator
    err = lambda a: np.mean(((a @ LUMW) - lr)[dark] ** 2 / lr[dark] ** 2)
    assert err(img) < err(ip)


def test_seeds_are_drawn_from_the_weighted_target(pkg, native_lib):
    """Two-stage chains sample f / importance. The reference draws their seeds in proportion to f (pathsampler.cpp:903-905: the
    luminance is taken before SplatList::normalize(importanceMap)), i.e. outside the chains' stationary distribution; over its
    work units of 1e5 mutations that start-up bias vanishes, over a device's many short chains it does not. The device draws the
    seeds from f / importance (deviation 19): a map of contrast 100 across the Cornell box, 4096 chains of 1024
    mutations -- columns within 3 % of a path-traced reference; with the reference's rule (seed_rule = DRMLT_SEED_REFERENCE) the
    bright half comes out 13 % high and the dark half 15 % low."""
    abi = pkg.abi
    sd = pkg.scenes.cornell_c2(64)
    kw = dict(technique="path", type="orbital", max_depth=6, rr_depth=5, direct_samples=-1, work_units=4096, sample_count=1024, luminance_samples=100000)
    cfg = abi.make_config(**kw)
    rc = pkg.Context(cfg, sd)
    ref = 0.5 * (rc.render_pt(16384, seed=11).astype(np.float64) + rc.render_pt(16384, seed=22))
    rc.close()
    imp = np.tile(np.where((np.arange(64) + 0.5) / 64 < 0.5, 0.01, 1.0), (64, 1)).astype(np.float32)

    def halves(n_renders, cfg):
        acc = np.zeros((64, 64, 3))
        for i in range(n_renders):
            c = pkg.Context(cfg, sd)
            c.set_importance_map(imp)
            b = c.seed(100 + i)
            c.run(64 * 64 * 1024)
            acc += c.develop()
            c.close()
        m, r = (acc / n_renders) @ LUMW, ref @ LUMW
        return m[:, :32].sum() / r[:, :32].sum(), m[:, 32:].sum() / r[:, 32:].sum(), b
    dark, bright, b = halves(6, cfg)
    assert abs(dark - 1) < 0.03 and abs(bright - 1) < 0.03, (dark, bright)
    assert b == pytest.approx((ref @ LUMW).mean(), rel=0.02)                 # b stays the mean of f itself
    dark, bright, _ = halves(6, abi.make_config(seed_rule="reference", **kw))
    assert dark < 0.92 and bright > 1.07, (dark, bright)


def test_a_map_that_is_zero_wherever_the_scene_contributes_has_its_own_error(pkg, native_lib):
    """mean(f) > 0 but no bootstrap sample has a finite weight under the map: not "average luminance is zero"."""
    sd = pkg.scenes.cornell_c2(32)
    cfg = pkg.abi.make_config(technique="path", type="orbital", max_depth=6, direct_samples=-1, work_units=256, luminance_samples=5000)
    ctx = pkg.Context(cfg, sd)
    ctx.set_importance_map(np.zeros((32, 32), dtype=np.float32))
    with pytest.raises(pkg.DrmltError, match="under the importance map"):
        ctx.seed(1)
    ctx.close()
    ctx = pkg.Context(pkg.abi.make_config(technique="path", type="orbital", max_depth=6, direct_samples=-1, work_units=256,
                                          luminance_samples=5000, seed_rule="reference"), sd)
    ctx.set_importance_map(np.zeros((32, 32), dtype=np.float32))
    assert ctx.seed(1) > 0     # the reference's rule seeds from f itself (its chains then never move: every state is invalid)
    ctx.close()


def test_timeout_stops_the_run(pkg, native_lib):
    sd = pkg.scenes.cornell_c2(64)
    cfg = pkg.abi.make_config(technique="path", type="orbital", max_depth=8, direct_samples=-1, work_units=65536,
                              sample_count=1, luminance_samples=1000, timeout_s=1)
    ctx = pkg.Context(cfg, sd)
    ctx.seed(1)
    t0 = time.time()
    ctx.run(65536 * 200000)                      # minutes of work without the deadline
    dt = time.time() - t0
    st = ctx.stats()
    assert 0.9 < dt < 3.0
    assert 0 < st.mutations < 65536 * 200000 and st.mutations % 65536 == 0
    img = ctx.develop()
    assert np.isfinite(img).all() and img.mean() > 0
