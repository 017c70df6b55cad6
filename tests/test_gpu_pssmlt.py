"""GPU parity for algo=pssmlt over technique=path (BASELINE config 1; PSSMLTRenderer::process,
src/integrators/pssmlt/pssmlt_proc.cpp:113-297 with PSSMLTSampler, pssmlt_sampler.cpp:93-168)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LUMW = np.array([0.212671, 0.715160, 0.072169])


def lum(img):
    return img @ LUMW


def make(pkg, ob, sd, **kw):
    abi = pkg.abi
    base = dict(algo=abi.ALGO_PSSMLT, technique="path", type="orbital", max_depth=8, rr_depth=5, direct_samples=-1,
                luminance_samples=20000)
    base.update(kw)
    cfg = abi.make_config(**base)
    return cfg, pkg.Context(cfg, sd), ob.Oracle(abi, cfg, sd, 64)


@pytest.mark.parametrize("kw", [dict(), dict(kelemen_style_mutation=0), dict(kelemen_style_weights=0),
                                dict(kelemen_style_mutation=0, kelemen_style_weights=0, p_large=0.1)],
                         ids=lambda k: "-".join("%s=%s" % i for i in k.items()) or "default")
def test_chains_track_the_oracle(pkg, ob, kw, native_lib):
    sd = pkg.scenes.cornell_c1(32)
    n_chains, n_mut = 2048, 48
    cfg, ctx, orc = make(pkg, ob, sd, work_units=n_chains, sample_count=1, **kw)
    bg, bo = ctx.seed(0xABCD), orc.seed(0xABCD)
    assert bg == pytest.approx(bo, rel=3e-4)
    (c0g, _), (c0o, _) = ctx.chain_state(34), orc.chain_state(34)
    same0 = np.abs(c0g["luminance"] - c0o["luminance"]) <= 1e-3 * c0o["luminance"]
    assert same0.mean() > 0.5
    # two launches: the cumulative weight is flushed at each launch end, the chain continues
    ctx.run(n_chains * 16); ctx.run(n_chains * (n_mut - 16))
    orc.run(n_chains * 16, 8); orc.run(n_chains * (n_mut - 16), 8)
    (cg, ug), (co, uo) = ctx.chain_state(34), orc.chain_state(34)
    tracked = same0 & np.all(np.abs(ug - uo[:, :34]) < 2e-3, axis=1)
    assert tracked.sum() / same0.sum() > 0.97, tracked.sum() / same0.sum()
    sg, so = ctx.stats(), orc.stats()
    assert sg.mutations == so.mutations == n_chains * n_mut
    for k in ("overall", "large", "bold"):
        bg_, bo_ = getattr(sg, k + "_base"), getattr(so, k + "_base")
        assert abs(bg_ - bo_) <= 0.01 * max(bo_, 1) + 20, (k, bg_, bo_)
        pg, po = getattr(sg, k + "_acc") / bg_, getattr(so, k + "_acc") / bo_
        assert abs(pg - po) < 4 * np.sqrt(po * (1 - po) / bo_) + 0.01, (k, pg, po)
    fg, fo = ctx.film(), orc.film()
    assert lum(fg).sum() == pytest.approx(lum(fo).sum(), rel=5e-3)
    bgk, bok = (lum(f).reshape(8, 4, 8, 4).sum(axis=(1, 3)) for f in (fg, fo))
    assert np.abs(bgk - bok).sum() / bok.sum() < 0.08
    ig, io = ctx.develop(), orc.develop()
    assert lum(ig).mean() == pytest.approx(lum(io).mean(), rel=3e-3)


def test_config1_image_matches_path_tracing(pkg, ob, native_lib):
    """BASELINE config 1 (Cornell 256 x 256, pssmlt / path, 2 diffuse quads + area light, sampleCount 64)."""
    sd = pkg.scenes.cornell_c1(256)
    cfg, ctx, _ = make(pkg, ob, sd, work_units=8192, sample_count=64, luminance_samples=100000)
    ref = pkg.Context(pkg.abi.make_config(max_depth=8, rr_depth=5, direct_samples=-1, work_units=64), sd).render_pt(256, seed=2)
    b = ctx.seed(0x5EED)
    ctx.run(256 * 256 * 64)
    img = ctx.develop()
    st = ctx.stats()
    assert st.mutations == 256 * 256 * 64 and st.overall_base == st.mutations and st.large_base + st.bold_base == st.mutations
    assert b == pytest.approx(lum(ref).mean(), rel=0.02)
    assert lum(img).mean() == pytest.approx(b, rel=1e-3)
    blk = lambda a: a.reshape(16, 16, 16, 16, 3).mean((1, 3))
    assert np.abs(blk(img) - blk(ref)).mean() / ref.mean() < 0.03


def test_refusals(pkg, native_lib):
    sd = pkg.scenes.cornell_c1(16)
    with pytest.raises(pkg.DrmltError, match="technique=path"):
        pkg.Context(pkg.abi.make_config(algo=pkg.abi.ALGO_PSSMLT, technique="mmlt", max_depth=5, work_units=64), sd)
