"""technique=mmlt in the oracle: the multiplexed estimator (pathsampler.cpp:84-320 with libbidir's vertex / edge /
miWeight code) and the delayed-rejection chain over its three samplers (drmlt_proc.cpp:84-141, 452-771).

No golden vectors of the reference exist for this path, so the restatement is checked through properties
the reference's estimator must have: every depth's multiplexed estimate equals the same depth of the unidirectional
estimator (an independent code path: path.cpp:123-315) in expectation, which holds only if the vertex densities, the
geometric terms and Path::miWeight are all right; and chains over it converge to that image for every kernel type."""
import numpy as np
import pytest


def lum(img):
    return img @ np.array([0.212671, 0.715160, 0.072169])


def blocks(img, n=4):
    h, w, _ = img.shape
    return img.reshape(n, h // n, n, w // n, 3).mean((1, 3))


@pytest.fixture(scope="module")
def pt_by_depth(pkg, abi, ob):
    """Unidirectional images truncated at maxDepth = 1..5 (differences give single depths)."""
    out = {}
    for name in ("cornell_c2", "glass_sphere"):
        sd = pkg.scenes.SCENES[name](16)
        imgs = {1: 0.0}
        for d in range(2, 6):
            cfg = abi.make_config(max_depth=d, rr_depth=100, work_units=4, direct_samples=-1)
            imgs[d] = ob.Oracle(abi, cfg, sd, 64).render_pt(3000, seed=7, nthreads=8)
        out[name] = (sd, imgs)
    return out


@pytest.mark.parametrize("name", ["cornell_c2", "glass_sphere"])
@pytest.mark.parametrize("depth", [2, 3, 4, 5])
def test_depth_estimate_matches_unidirectional(pt_by_depth, abi, ob, name, depth):
    sd, imgs = pt_by_depth[name]
    want = imgs[depth] - imgs[depth - 1]
    cfg = abi.make_config(max_depth=8, rr_depth=100, work_units=4, direct_samples=-1)
    o = ob.Oracle(abi, cfg, sd, 64)
    got, strat = o.mmlt_render(depth, 16 * 16 * 3000, seed=3, nthreads=8)
    # strategies s = 0 .. depth exist with a light image; s = depth + 1 never (t >= 1)
    assert strat[depth + 1] == 0 and strat[1:depth + 1].min() > 0
    # the unidirectional estimate of caustic depths is heavy-tailed at this sample count: looser for glass
    rel, blk = (0.03, 0.1) if name == "cornell_c2" else (0.08, 0.2)
    assert lum(got).mean() == pytest.approx(lum(want).mean(), rel=rel)
    err = np.abs(blocks(got) - blocks(want)).mean() / want.mean()
    assert err < blk


def test_depth_one_and_direct_exclusion(pkg, abi, ob):
    """depth == 1 returns no splat (pathsampler.cpp:131-135); separate direct drops depth <= 2 (:274-280)."""
    sd = pkg.scenes.cornell_c2(16)
    rng = np.random.default_rng(5)
    us, ue, ud = (rng.random((2000, 24), dtype=np.float32) for _ in range(3))
    o = ob.Oracle(abi, abi.make_config(max_depth=8, direct_samples=-1, work_units=4), sd, 64)
    sp, st = o.mmlt_eval(1, us, ue, ud[:, 0])
    assert (sp["luminance"] == 0).all() and (sp["n_rays"] == 0).all()
    assert set(map(tuple, st)) == {(0, 2), (1, 1)}
    sp2, _ = o.mmlt_eval(2, us, ue, ud[:, 0])
    assert (sp2["luminance"] > 0).any()
    o_sep = ob.Oracle(abi, abi.make_config(max_depth=8, direct_samples=16, work_units=4), sd, 64)
    sp3, _ = o_sep.mmlt_eval(2, us, ue, ud[:, 0])
    assert (sp3["luminance"] == 0).all()
    sp4, _ = o_sep.mmlt_eval(3, us, ue, ud[:, 0])
    np.testing.assert_array_equal(sp4["luminance"], o.mmlt_eval(3, us, ue, ud[:, 0])[0]["luminance"])


def test_strategy_selection_and_dimensions(pkg, abi, ob):
    """s = min(int(nStrats * xi), nStrats - 1), t = nStrats - s with nStrats = depth + 1 (:107-113); without the
    light image nStrats = depth and t >= 2 (:114-124). Dimensions: 1 (direct) + 2 per sampled vertex."""
    sd = pkg.scenes.cornell_c2(16)
    rng = np.random.default_rng(9)
    n = 4000
    us, ue = rng.random((n, 24), dtype=np.float32), rng.random((n, 24), dtype=np.float32)
    ud = rng.random(n, dtype=np.float32)
    o = ob.Oracle(abi, abi.make_config(max_depth=8, direct_samples=-1, work_units=4), sd, 64)
    for depth in (2, 4, 7):
        sp, st = o.mmlt_eval(depth, us, ue, ud)
        s_want = np.minimum(((depth + 1) * ud.astype(np.float64)).astype(int), depth)
        np.testing.assert_array_equal(st[:, 0], s_want)
        np.testing.assert_array_equal(st[:, 1], depth + 1 - s_want)
        assert sp["n_dims"].max() <= 1 + 2 * (depth + 1)
        ok = sp["luminance"] > 0
        np.testing.assert_array_equal(sp["n_dims"][ok], 1 + 2 * (depth + 1))   # complete walks: 2 per step
        sp_n, st_n = o.mmlt_eval(depth, us, ue, ud, light_image=False)
        assert st_n[:, 1].min() >= 2 and (st_n.sum(1) == depth + 1).all()
    assert ob.lib().oracle_find_max_dim_mmlt(3) == 16 and ob.lib().oracle_find_max_dim_mmlt(4) == 18


@pytest.mark.parametrize("kw", [
    dict(type="orbital"), dict(type="green"), dict(type="mira"),
    dict(type="orbital", fix_emitter_path=1), dict(type="mira", fix_emitter_path=1),
    dict(type="green", fix_emitter_path=1), dict(type="orbital", use_mixture=1),
], ids=lambda kw: "-".join("%s=%s" % kv for kv in kw.items()))
def test_chains_converge_to_unidirectional_image(pkg, abi, ob, kw):
    sd = pkg.scenes.glass_sphere(16)
    ref = ob.Oracle(abi, abi.make_config(max_depth=6, rr_depth=100, work_units=4, direct_samples=-1), sd, 64) \
        .render_pt(3000, seed=7, nthreads=8)
    cfg = abi.make_config(technique="mmlt", max_depth=6, work_units=4096, direct_samples=-1,
                          luminance_samples=200000, **kw)
    o = ob.Oracle(abi, cfg, sd, 64)
    b = o.seed(1234)
    o.run(16 * 16 * 6000, nthreads=8)
    img = o.develop()
    assert b == pytest.approx(lum(ref).mean(), rel=0.05)
    assert np.abs(blocks(img) - blocks(ref)).mean() / ref.mean() < 0.05
    st = o.stats()
    assert st.mutations == 16 * 16 * 6000 // 4096 * 4096
    if kw.get("use_mixture"):
        assert st.second_base > 0
    else:
        assert st.second_base == st.first_base - st.first_acc - (st.large_base - st.large_acc)


def test_refused_configurations(pkg, abi, ob):
    sd = pkg.scenes.cornell_c2(16)
    for kw, msg in ((dict(technique="mmlt", max_depth=-1), "no max depth"),
                    (dict(technique="path", max_depth=5, fix_emitter_path=1), "fixEmitterPath without MMLT"),
                    (dict(technique="mmlt", max_depth=5, timid_after_large=1), "timidAfterLarge")):
        with pytest.raises(ob.OracleError, match=msg):
            ob.Oracle(abi, abi.make_config(work_units=4, **kw), sd, 64)


def test_seed_depths_and_replay(pkg, abi, ob):
    """Bootstrap sample i has depth (i % maxDepth) + 1 (:889); b = mean * maxDepth (:932-934); every chain
    reproduces its seed's luminance on replay (drmlt_proc.cpp:509-512) and its state is [sensor|emitter|direct]."""
    sd = pkg.scenes.cornell_c2(16)
    cfg = abi.make_config(technique="mmlt", type="orbital", max_depth=5, work_units=256, direct_samples=-1,
                          luminance_samples=1000)
    o = ob.Oracle(abi, cfg, sd, 64)
    b = o.seed(77)
    n = max(1000, 256 * 50) * 5
    lums = o.bootstrap_lum(77, 0, n)
    assert b == pytest.approx(lums.astype(np.float64).mean() * 5, rel=1e-5)
    assert (lums[0::5] == 0).all()                      # depth 1: nothing
    assert (lums[1::5] > 0).any() and (lums[4::5] > 0).any()
    cur, u = o.chain_state(64)
    assert (cur["luminance"] > 0).all()                 # replay succeeded (seed() would have failed otherwise)
    assert ((u >= 0) & (u <= 1)).all()


# ---------------------------------------------------------------- technique=bdpt (pathsampler.cpp:321-527)

@pytest.mark.parametrize("name,maxd,rr", [("cornell_c2", 3, 100), ("cornell_c2", 5, 2), ("glass_sphere", 5, 100)])
def test_bdpt_estimate_matches_unidirectional(pkg, abi, ob, name, maxd, rr):
    """All (s, t) connections with Path::miWeight, random walks with russian roulette from rrDepth, light-image splats.
    With the direct component excluded (depth <= 2, :407-408) both estimators cover the same paths."""
    sd = pkg.scenes.SCENES[name](16)
    ref = ob.Oracle(abi, abi.make_config(max_depth=maxd, rr_depth=100, work_units=4, direct_samples=16), sd, 64) \
        .render_pt(6000, seed=7, nthreads=8)
    cfg = abi.make_config(technique="bdpt", max_depth=maxd, rr_depth=rr, work_units=4, direct_samples=16, no_direct_sampling=1)
    img = ob.Oracle(abi, cfg, sd, 64).bdpt_render(16 * 16 * 6000, seed=3, nthreads=8)
    assert lum(img).mean() == pytest.approx(lum(ref).mean(), rel=0.02)
    assert np.abs(blocks(img) - blocks(ref)).mean() / ref.mean() < 0.03


def test_bdpt_splat_lists(pkg, abi, ob):
    """One main splat when the camera ray hits something (:357-361) + one light-image splat per t = 1 connection
    (:514-519); luminance = sum over all splats; without the light image no t < 2 strategy (:372)."""
    sd = pkg.scenes.cornell_c2(16)
    rng = np.random.default_rng(2)
    us, ue = rng.random((3000, 30), dtype=np.float32), rng.random((3000, 30), dtype=np.float32)
    cfg = abi.make_config(technique="bdpt", max_depth=5, rr_depth=3, work_units=4, direct_samples=-1, no_direct_sampling=1)
    rows = ob.Oracle(abi, cfg, sd, 64).bdpt_eval(us, ue)
    n_more = rows[:, 7].astype(int)
    assert rows[:, 1].mean() > 0.9 and n_more.max() <= 5 and n_more.max() >= 3    # edge rays miss the open box
    more = rows[:, 10:].reshape(len(rows), -1, 5)
    total = rows[:, 4:7] @ np.array([0.212671, 0.715160, 0.072169]) + (more[:, :, 2:] @ np.array([0.212671, 0.715160, 0.072169])).sum(1)
    np.testing.assert_allclose(total, rows[:, 0], rtol=1e-5, atol=1e-7)
    assert ((more[:, :, 0] >= 0) & (more[:, :, 0] <= 16) & (more[:, :, 1] >= 0) & (more[:, :, 1] <= 16)).all()
    cfg2 = abi.make_config(technique="bdpt", max_depth=5, rr_depth=3, work_units=4, direct_samples=-1, no_direct_sampling=1,
                           no_light_image=1)
    rows2 = ob.Oracle(abi, cfg2, sd, 64).bdpt_eval(us, ue)
    assert rows2[:, 7].max() == 0
    assert rows2[:, 8].max() <= 2 * 6 + 3 + 2 * 5 + 2            # dims: 2 per step + 1 per roulette test


@pytest.mark.parametrize("kw", [dict(type="orbital"), dict(type="green"), dict(type="mira"), dict(type="orbital", use_mixture=1)],
                         ids=lambda kw: "-".join("%s=%s" % kv for kv in kw.items()))
def test_bdpt_chains_converge(pkg, abi, ob, kw):
    sd = pkg.scenes.glass_sphere(16)
    cfg = abi.make_config(technique="bdpt", max_depth=6, rr_depth=5, work_units=2048, direct_samples=-1, no_direct_sampling=1,
                          luminance_samples=100000, **kw)
    o = ob.Oracle(abi, cfg, sd, 64)
    ref = o.bdpt_render(16 * 16 * 4000, seed=9, nthreads=8)
    b = o.seed(1234)
    o.run(16 * 16 * 3000, nthreads=8)
    img = o.develop()
    assert b == pytest.approx(lum(ref).mean(), rel=0.02)
    assert np.abs(blocks(img) - blocks(ref)).mean() / ref.mean() < 0.03


@pytest.mark.parametrize("name,maxd,rr", [("cornell_c2", 5, 2), ("glass_sphere", 5, 100), ("caustic_c5", 5, 100), ("door_c3", 4, 100)])
def test_bdpt_direct_sampling_is_the_same_estimator(pkg, abi, ob, name, maxd, rr):
    """directSampling = true (the reference's default, pathsampler.cpp:424-452 with miWeight's ratioEmitterDirect,
    path.cpp:936-965): the s = 1 strategies draw their emitter point from the connecting vertex (cone sampling on the sphere
    light of caustic_c5, where the MIS ratio is not 0 / 1), the expectation is unchanged -- full transport, direct light included."""
    sd = pkg.scenes.SCENES[name](16)
    n = 16 * 16 * 8000
    a = ob.Oracle(abi, abi.make_config(technique="bdpt", max_depth=maxd, rr_depth=rr, work_units=4, direct_samples=-1, no_direct_sampling=1), sd, 64) \
        .bdpt_render(n, seed=3, nthreads=8)
    b = ob.Oracle(abi, abi.make_config(technique="bdpt", max_depth=maxd, rr_depth=rr, work_units=4, direct_samples=-1), sd, 64) \
        .bdpt_render(n, seed=5, nthreads=8)
    assert lum(b).mean() == pytest.approx(lum(a).mean(), rel=0.02)
    assert np.abs(blocks(a) - blocks(b)).mean() / a.mean() < 0.04


def test_bdpt_direct_sampling_dimensions_and_chains(pkg, abi, ob):
    """Two components of the direct sampler per s = 1 / t = 1 connection, at most 2 (2 maxDepth - 1) (the reference sizes the
    sampler to maxDepth and overruns it, pssmlt_utils.h:75); t = 1 splats are what they are without direct sampling (a pinhole
    has one point to sample); chains with the three samplers converge to the same image."""
    sd = pkg.scenes.cornell_c2(16)
    rng = np.random.default_rng(2)
    us, ue, ud = (rng.random((2000, 30), dtype=np.float32) for _ in range(3))
    off = ob.Oracle(abi, abi.make_config(technique="bdpt", max_depth=5, rr_depth=3, work_units=4, direct_samples=-1, no_direct_sampling=1), sd, 64)
    on = ob.Oracle(abi, abi.make_config(technique="bdpt", max_depth=5, rr_depth=3, work_units=4, direct_samples=-1), sd, 64)
    r0, r1 = off.bdpt_eval(us, ue), on.bdpt_eval(us, ue, ud)
    extra = r1[:, 8] - r0[:, 8]
    assert (extra % 2 == 0).all() and extra.min() >= 0 and 0 < extra.max() <= 2 * (2 * 5 - 1)
    assert np.array_equal(r0[:, 7], r1[:, 7])                                # the same light-image splats ...
    m0, m1 = r0[:, 10:].reshape(len(r0), -1, 5), r1[:, 10:].reshape(len(r1), -1, 5)
    np.testing.assert_allclose(m1[:, :, :2], m0[:, :, :2], atol=1e-9)        # ... at the same pixels
    # their values differ only through the MIS weights (ratioEmitterDirect is 0 where vertex 2 sees the back of the light)
    nz = (m0[:, :, 2] > 0) & (m1[:, :, 2] > 0)
    assert nz.mean() > 0.2 and np.median(np.abs(m1[:, :, 2][nz] / m0[:, :, 2][nz] - 1)) < 0.05
    with pytest.raises(ob.OracleError, match="direct sampler"):
        on.bdpt_eval(us, ue)
    sd = pkg.scenes.glass_sphere(16)
    cfg = abi.make_config(technique="bdpt", max_depth=6, rr_depth=5, work_units=2048, direct_samples=-1, luminance_samples=100000, type="orbital")
    o = ob.Oracle(abi, cfg, sd, 64)
    ref = o.bdpt_render(16 * 16 * 4000, seed=9, nthreads=8)
    b = o.seed(1234)
    cur, u = o.chain_state(22 + 20 + 22)
    assert u.shape[1] == 64 and ((u >= 0) & (u <= 1)).all() and (u[:, 42:] > 0).any()   # [sensor | emitter | direct] state
    o.run(16 * 16 * 3000, nthreads=8)
    img = o.develop()
    assert b == pytest.approx(lum(ref).mean(), rel=0.02)
    assert np.abs(blocks(img) - blocks(ref)).mean() / ref.mean() < 0.03


def test_sphere_area_light(pkg, abi, ob):
    """Sphere emitters (sphere.cpp:257-385): the unidirectional estimator samples the cone the sphere subtends (NEE) and
    weighs BSDF hits with the same solid-angle density; the bidirectional one samples the sphere's area uniformly.
    Two different sampling routes, one integrand."""
    sd = pkg.scenes.caustic_c5(16)
    ref = ob.Oracle(abi, abi.make_config(max_depth=5, rr_depth=100, work_units=4, direct_samples=16), sd, 64) \
        .render_pt(8000, seed=7, nthreads=8)
    img = ob.Oracle(abi, abi.make_config(technique="bdpt", max_depth=5, rr_depth=100, work_units=4, direct_samples=16,
                                          no_direct_sampling=1), sd, 64).bdpt_render(16 * 16 * 8000, seed=3, nthreads=8)
    assert lum(img).mean() == pytest.approx(lum(ref).mean(), rel=0.02)
    assert np.abs(blocks(img) - blocks(ref)).mean() / ref.mean() < 0.03
    direct = ob.Oracle(abi, abi.make_config(max_depth=2, rr_depth=100, work_units=4, direct_samples=-1), sd, 64) \
        .render_pt(8000, seed=7, nthreads=8)
    m, strat = ob.Oracle(abi, abi.make_config(technique="mmlt", max_depth=2, work_units=4, direct_samples=-1), sd, 64) \
        .mmlt_render(2, 16 * 16 * 8000, seed=5, nthreads=8)
    assert lum(m).mean() == pytest.approx(lum(direct).mean(), rel=0.02) and strat[1] > 0 and strat[2] > 0
