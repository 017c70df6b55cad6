"""GPU parity for technique=mmlt (BASELINE config 5): every call goes through the C-ABI and is
compared with the CPU oracle's restatement of the multiplexed estimator and its three-sampler chains.
Tolerances (fp32 device vs fp64 oracle) are stated per test, as in test_gpu_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LUMW = np.array([0.212671, 0.715160, 0.072169])


def lum(img):
    return img @ LUMW


def make(pkg, ob, sd, precision=64, **kw):
    abi = pkg.abi
    base = dict(technique="mmlt", max_depth=6, direct_samples=-1, luminance_samples=20000)
    base.update(kw)
    cfg = abi.make_config(**base)
    return cfg, pkg.Context(cfg, sd), ob.Oracle(abi, cfg, sd, precision)


@pytest.mark.parametrize("name", ["cornell_c2", "glass_sphere", "door_c3", "caustic_c5"])
def test_eval_matches_oracle(pkg, ob, name, native_lib):
    """f(u) on identical PSS points, every depth and strategy: same (s, t), same ray count for >= 99.5 %,
    luminance within 2e-3 relative at the 99th percentile."""
    sd = pkg.scenes.SCENES[name](res=64)
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", work_units=64)
    rng = np.random.default_rng(7)
    n = 8192
    us, ue = rng.random((n, 14), dtype=np.float32), rng.random((n, 14), dtype=np.float32)  # same row stride for both
    ud = rng.random(n, dtype=np.float32)
    total_g = total_o = 0.0
    for depth in range(1, 7):
        g, stg = ctx.eval_paths_mmlt(depth, us, ue, ud)
        o, sto = orc.mmlt_eval(depth, us, ue, ud)
        assert (stg == sto).all(axis=1).mean() > 0.9999            # int(nStrats * xi) in fp32 vs fp64
        same = (g["n_rays"] == o["n_rays"]) & (stg == sto).all(axis=1) & ((g["luminance"] > 0) == (o["luminance"] > 0))
        assert same.mean() >= 0.995, (depth, same.mean())
        assert (g["n_dims"] == o["n_dims"])[same].mean() > 0.999
        pos = same & (o["luminance"] > 0)
        if pos.any():
            rel = np.abs(g["luminance"] - o["luminance"])[pos] / np.maximum(o["luminance"][pos], 1e-3)
            assert np.quantile(rel, 0.99) < 2e-3, (depth, np.quantile(rel, 0.99))
            # light tracing (t = 1) projects the end of an fp32 light path onto the film: 1e-2 px after 6 bounces
            assert np.allclose(g["x"][pos], o["x"][pos], atol=2e-2) and np.allclose(g["y"][pos], o["y"][pos], atol=2e-2)
            t2 = pos & (sto[:, 1] >= 2)
            assert np.allclose(g["x"][t2], o["x"][t2], atol=1e-3) and np.allclose(g["y"][t2], o["y"][t2], atol=1e-3)
            assert np.allclose(g["rgb"][pos], o["rgb"][pos], rtol=5e-2, atol=1e-3)
        total_g += g["luminance"].sum(); total_o += o["luminance"].sum()
    assert total_g == pytest.approx(total_o, rel=5e-3)


def _glint_scene(pkg, res=64):
    """A big glass sphere in front of the camera under a big ceiling light: camera -> sphere (reflection) -> light is a common path."""
    S = pkg.scenes
    sd = S.SceneData("glint")
    white, red, green, black = sd.diffuse(0.725, 0.71, 0.68), sd.diffuse(0.63, 0.065, 0.05), sd.diffuse(0.14, 0.45, 0.091), sd.diffuse(0.0)
    glass = sd.dielectric(1.5, 1.0)
    S._room(sd, white, red, green)
    sd.sphere((0.0, -0.2, 0.6), 0.6, glass)
    sd.rectangle(S.translate(0, 0.995, 0) @ S.rotate("x", 90) @ S.scale(0.7), black, radiance=10.0)
    sd.set_camera(S.lookat((0, 0, 3.9), (0, 0, 0), (0, 1, 0)), 39.3077, res, res, pkg.abi.FILTER_BOX, 0.5)
    return sd


def test_camera_paths_that_reach_a_light_over_specular_vertices_only(pkg, ob, native_lib):
    """s = 0 with nothing but specular vertices between the camera and the emitter it hits (E S* L: the glint of a light on a
    glass sphere): the reference's "subpaths are connectable" test (pathsampler.cpp:161-173) runs over the sensor vertices 2 .. t
    INCLUDING the last one, which lies on the emitter and is connectable -- such paths count. (Until round 3 the device tested
    positions 2 .. k - 2 only and dropped them: on config 5's scene that is a two-pixel glint at 64 x 64, found by the N = 64
    parity protocol's permutation test.)"""
    sd = _glint_scene(pkg)
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", work_units=64)
    rng = np.random.default_rng(21)
    n = 1 << 17
    us, ue = rng.random((n, 14), dtype=np.float32), rng.random((n, 14), dtype=np.float32)
    for depth in (2, 3, 4):
        ud = (rng.random(n, dtype=np.float32) * np.float32(0.999 / (depth + 1)))      # s = int((depth + 1) * u) = 0: pure camera paths
        g, stg = ctx.eval_paths_mmlt(depth, us, ue, ud)
        o, sto = orc.mmlt_eval(depth, us, ue, ud)
        assert (sto[:, 0] == 0).all() and (stg[:, 0] == 0).all()
        hit = o["luminance"] > 0
        assert hit.sum() > 1000
        both = hit & (g["luminance"] > 0)
        assert both.sum() >= 0.998 * hit.sum(), (depth, both.sum(), hit.sum())          # (the unfixed kernel: 0.98 at depth 2)
        rel = np.abs(g["luminance"] - o["luminance"])[both] / o["luminance"][both]
        assert np.quantile(rel, 0.99) < 5e-3
        assert g["luminance"].sum() == pytest.approx(o["luminance"].sum(), rel=3e-3)


def test_bootstrap_and_seed_replay(pkg, ob, native_lib):
    sd = pkg.scenes.glass_sphere(64)
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", work_units=2048, sample_count=1, luminance_samples=1000)
    bg, bo = ctx.seed(0x5EED), orc.seed(0x5EED)      # 2048 * 50 * 6 bootstrap samples; raises on a replay mismatch
    assert bg == pytest.approx(bo, rel=5e-4)
    dim = 14 + 12 + 1
    (cg, ug), (co, uo) = ctx.chain_state(dim), orc.chain_state(dim)
    assert set(np.unique(cg["n_dims"])) <= set(range(2, 7))              # depth 1 has no contribution
    # same seeds picked for most chains; for those the replayed state matches component by component
    # (components beyond what a path of that depth consumes are not part of the comparison)
    same_seed = (cg["n_dims"] == co["n_dims"]) & (np.abs(cg["luminance"] - co["luminance"]) <= 1e-3 * co["luminance"])
    assert same_seed.mean() > 0.5
    for i in np.flatnonzero(same_seed)[:400]:
        d = int(cg["n_dims"][i])
        np.testing.assert_array_equal(ug[i, :2 * (d + 1)], uo[i, :2 * (d + 1)])
        np.testing.assert_array_equal(ug[i, 14:14 + 2 * d], uo[i, 14:14 + 2 * d])
        assert ug[i, 26] == uo[i, 26]
    assert (cg["n_rays"] == co["n_rays"])[same_seed].all()                # t of the current state
    st = ctx.stats()
    assert st.n_chains == 2048 and st.max_dim == 2 * 24 + 1


VARIANTS = [
    dict(type="orbital"), dict(type="green"), dict(type="mira"),
    dict(type="orbital", fix_emitter_path=1), dict(type="green", fix_emitter_path=1), dict(type="mira", fix_emitter_path=1),
    dict(type="orbital", use_mixture=1), dict(type="orbital", direct_samples=16), dict(type="orbital", no_light_image=1),
]


@pytest.mark.parametrize("kw", VARIANTS, ids=lambda k: "-".join("%s=%s" % i for i in k.items()))
def test_chains_track_the_oracle(pkg, ob, kw, native_lib):
    sd = pkg.scenes.glass_sphere(32)
    n_chains, n_mut = 2048, 40
    cfg, ctx, orc = make(pkg, ob, sd, work_units=n_chains, sample_count=1, luminance_samples=1000, **kw)
    ctx.seed(0xABCD), orc.seed(0xABCD)
    dim = 27
    (c0g, u0g), (c0o, u0o) = ctx.chain_state(dim), orc.chain_state(dim)

    def used(c, u):  # the components a chain's paths can consume
        m = np.zeros_like(u, dtype=bool)
        for i, d in enumerate(c["n_dims"]):
            m[i, :2 * (d + 1)] = True
            m[i, 14:14 + 2 * d] = c["n_rays"][i] <= d   # s >= 1: the emitter state exists
            m[i, 26] = True
        return m

    same0 = (c0g["n_dims"] == c0o["n_dims"]) & np.all((u0g == u0o) | ~used(c0o, u0o), axis=1)
    assert same0.mean() > 0.5
    ctx.run(n_chains * n_mut)
    orc.run(n_chains * n_mut, 8)
    (cg, ug), (co, uo) = ctx.chain_state(dim), orc.chain_state(dim)
    tracked = same0 & (cg["n_rays"] == co["n_rays"]) & np.all((np.abs(ug - uo) < 2e-3) | ~used(co, uo), axis=1)
    assert tracked.sum() / same0.sum() > 0.95, tracked.sum() / same0.sum()
    sg, so = ctx.stats(), orc.stats()
    assert sg.mutations == so.mutations == n_chains * n_mut
    for k in ("first", "large", "bold", "second", "second_bold", "overall"):
        bg, bo = getattr(sg, k + "_base"), getattr(so, k + "_base")
        assert abs(bg - bo) <= 0.02 * max(bo, 1) + 20, (k, bg, bo)
        if bo > 200:
            pg, po = getattr(sg, k + "_acc") / bg, getattr(so, k + "_acc") / bo
            assert abs(pg - po) < 4 * np.sqrt(po * (1 - po) / bo) + 0.015, (k, pg, po)
    assert abs(sg.rays - so.rays) <= 0.03 * so.rays
    fg, fo = ctx.film(), orc.film()
    assert lum(fg).sum() == pytest.approx(lum(fo).sum(), rel=5e-3)
    bgk, bok = (lum(f).reshape(8, 4, 8, 4).sum(axis=(1, 3)) for f in (fg, fo))
    assert np.abs(bgk - bok).sum() / bok.sum() < 0.1


def test_config5_caustic_acceptance_map(pkg, ob, native_lib):
    """BASELINE config 5: glass caustic, mmlt / orbital / fixEmitterPath / acceptanceMap. The map counts accepted
    first-stage (red) and second-stage (green) small steps per pixel (drmlt_proc.cpp:697-709); b is forced to 1."""
    sd = pkg.scenes.caustic_c5(64)      # dielectric sphere + small SPHERE area light (+ a dim quad light)
    n_chains = 8192
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", fix_emitter_path=1, acceptance_map=1, work_units=n_chains,
                         sample_count=1, luminance_samples=1000)
    assert ctx.seed(0x5EED) == 1.0
    orc.seed(0x5EED)
    n_mut = 64
    ctx.run(n_chains * n_mut)
    orc.run(n_chains * n_mut, 8)
    fg, fo = ctx.film().astype(np.float64), orc.film().astype(np.float64)   # counts: exact in fp32, summed in fp64
    sg, so = ctx.stats(), orc.stats()
    assert fg[..., 2].max() == 0
    # one splat of weight ~1 per accepted small step (the discretised box filter is not exactly 1, and a
    # splat on a pixel border touches two pixels: imageblock.h:150-216)
    assert fg[..., 0].sum() == pytest.approx(sg.bold_acc, rel=1e-4) and fg[..., 1].sum() == pytest.approx(sg.second_acc, rel=1e-4)
    assert fg[..., 0].sum() == pytest.approx(fo[..., 0].sum(), rel=0.02)
    assert fg[..., 1].sum() == pytest.approx(fo[..., 1].sum(), rel=0.03)
    heat = lambda f: f[..., 1].sum() / (f[..., 0].sum() + f[..., 1].sum())
    assert heat(fg) == pytest.approx(heat(fo), abs=0.01)
    img = ctx.develop()
    np.testing.assert_allclose(img, fg.astype(np.float32), rtol=1e-6)       # acceptanceMap: develop does not rescale (:834-839)


def test_mmlt_image_matches_path_tracing(pkg, ob, native_lib):
    """Equal-budget protocol of BASELINE.md: device MMLT image vs a converged unidirectional image of the same
    scene (device path tracer), next to the oracle's MMLT image with the same budget."""
    sd = pkg.scenes.glass_sphere(32)
    cfgp = pkg.abi.make_config(technique="path", type="orbital", max_depth=6, rr_depth=100, direct_samples=-1, work_units=64)
    ref = pkg.Context(cfgp, sd).render_pt(8192, seed=5)
    n_chains, spp = 16384, 2048
    cfg, ctx, orc = make(pkg, ob, sd, type="orbital", fix_emitter_path=1, work_units=n_chains, sample_count=spp,
                         luminance_samples=1000)
    bg = ctx.seed(0x5EED)
    ctx.run(32 * 32 * spp)
    img = ctx.develop()
    assert bg == pytest.approx(lum(ref).mean(), rel=0.03)
    blk = lambda a: a.reshape(8, 4, 8, 4, 3).mean((1, 3))
    assert np.abs(blk(img) - blk(ref)).mean() / ref.mean() < 0.04
    orc.seed(0x5EED)
    orc.run(32 * 32 * spp, 8)
    io = orc.develop()
    e_g = np.mean((lum(img) - lum(ref)) ** 2 / (lum(ref) ** 2 + 1e-2 * lum(ref).mean() ** 2))
    e_o = np.mean((lum(io) - lum(ref)) ** 2 / (lum(ref) ** 2 + 1e-2 * lum(ref).mean() ** 2))
    assert e_g < 1.5 * e_o + 1e-3, (e_g, e_o)


def test_refused_configurations(pkg, native_lib):
    sd = pkg.scenes.cornell_c2(16)
    for kw, msg in ((dict(technique="mmlt", max_depth=-1), "no max depth"),
                    (dict(technique="path", max_depth=5, fix_emitter_path=1), "fixEmitterPath without MMLT"),
                    (dict(technique="mmlt", max_depth=5, timid_after_large=1), "timidAfterLarge")):   # (bdpt takes timidAfterLarge since round 4)
        with pytest.raises(pkg.DrmltError, match=msg):
            pkg.Context(pkg.abi.make_config(work_units=64, **kw), sd)


def test_depth_ordered_execution_changes_no_chain(pkg, native_lib):
    """k_mutate_mmlt runs the chains in order of their path depth, deepest first (waves of one depth: a wave costs what its
    deepest chain costs). Only the wave a chain rides in changes: chain ids, states and random streams are untouched."""
    import os
    sd = pkg.scenes.caustic_c5(32)
    n = 4096 + 37                                        # a ragged last wave
    cfg = pkg.abi.make_config(technique="mmlt", type="orbital", max_depth=6, direct_samples=-1, fix_emitter_path=1, work_units=n,
                              sample_count=1, luminance_samples=20000)
    res = []
    for no_sort in (False, True):
        if no_sort:
            os.environ["DRMLT_MMLT_NO_SORT"] = "1"
        try:
            ctx = pkg.Context(cfg, sd)
            ctx.seed(0xBEEF)
        finally:
            os.environ.pop("DRMLT_MMLT_NO_SORT", None)
        ctx.run(n * 50)
        res.append((ctx.chain_state(28), ctx.stats(), ctx.film()))
    (c0, u0), s0, f0 = res[0]
    (c1, u1), s1, f1 = res[1]
    assert np.array_equal(u0, u1) and np.array_equal(c0["luminance"], c1["luminance"]) and np.array_equal(c0["n_dims"], c1["n_dims"])
    assert s0.accepted == s1.accepted and s0.rays == s1.rays and s0.mutations == s1.mutations == n * 50
    assert lum(f0).sum() == pytest.approx(lum(f1).sum(), rel=1e-5)
    depths = c0["n_dims"]                                # chain_state reports the chain's path depth there
    assert depths.min() >= 1 and depths.max() == 6 and len(np.unique(depths)) >= 4      # (depth-1 paths carry nothing here: no seeds)
