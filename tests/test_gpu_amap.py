"""The acceptance map PER PIXEL (BASELINE config 5's deliverable), through the C-ABI.

Reference semantics (drmlt_proc.cpp:693-709): the accepted proposal is swapped into `current` FIRST, then
splatAcceptanceOnly is called on `proposed.first` / `proposed.second` -- which after the swap own the list that WAS current.
The mark therefore lands on the pixel(s) of the state being LEFT. Rounds 1-2 marked the adopted state in oracle AND
kernels, and tests that compared channel sums could not see it. Two kinds of test here:
  * oracle-free: after ONE mutation from the seeds, the map must be the histogram of the SEED pixels of the chains that
    moved (chain_state before / after) -- the reference's order of operations checked on the device alone;
  * device map vs oracle map pixel by pixel (the oracle's own order is pinned by the literal replay in
    tests/test_oracle_process.py::test_acceptance_map_marks_the_state_being_left)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W_BOX = (1.0 / 1.00002) ** 2     # one mark through the discretised box filter (rfilter.cpp:37-55: radius 0.5 + 1e-5)


def make(pkg, ob, sd, **kw):
    base = dict(max_depth=8, rr_depth=5, direct_samples=-1, luminance_samples=20000, acceptance_map=1, sample_count=1)
    base.update(kw)
    cfg = pkg.abi.make_config(**base)
    return cfg, pkg.Context(cfg, sd), ob.Oracle(pkg.abi, cfg, sd, 64)


def pixel_hist(px, py, w, h):
    ix, iy = np.clip(px.astype(int), 0, w - 1), np.clip(py.astype(int), 0, h - 1)
    out = np.zeros((h, w))
    np.add.at(out, (iy, ix), 1.0)
    return out


def interior(px, py):
    """splats further than 1e-4 from a pixel border: they touch exactly one pixel"""
    fx, fy = px - np.floor(px), py - np.floor(py)
    return (np.minimum(fx, 1 - fx) > 1e-4) & (np.minimum(fy, 1 - fy) > 1e-4)


@pytest.mark.parametrize("tech,kernel", [("path", 4), ("path", 3), ("mmlt", 0)])
def test_one_mutation_marks_the_seed_pixels(pkg, ob, tech, kernel, native_lib, monkeypatch):
    """After one mutation with pLarge = 0 every chain that moved has left exactly one mark -- at its SEED state's pixel."""
    if kernel:
        monkeypatch.setenv("DRMLT_KERNEL", str(kernel))
    res = 96
    n_chains = 4096
    if tech == "path":
        sd, kw, dim = pkg.scenes.cornell_c2(res), dict(type="orbital"), 34
    else:
        sd, kw, dim = pkg.scenes.caustic_c5(res), dict(technique="mmlt", type="orbital", max_depth=6, fix_emitter_path=1), 27
    cfg, ctx, orc = make(pkg, ob, sd, work_units=n_chains, p_large=0.0, **kw)
    orc.close()
    ctx.seed(0xA15)
    c0, u0 = ctx.chain_state(dim)
    ctx.run(n_chains)                                   # one mutation per chain
    c1, u1 = ctx.chain_state(dim)
    moved = np.any(u0 != u1, axis=1)
    st = ctx.stats()
    assert moved.sum() == st.accepted > n_chains // 4
    fm = ctx.film().astype(np.float64)
    got = (fm[..., 0] + fm[..., 1]) / W_BOX
    assert fm[..., 2].max() == 0
    assert fm[..., 0].sum() == pytest.approx(st.bold_acc * W_BOX, rel=1e-4)
    assert fm[..., 1].sum() == pytest.approx(st.second_acc * W_BOX, rel=1e-4)
    ok = interior(c0["x"], c0["y"]) & interior(c1["x"], c1["y"])
    left = pixel_hist(c0["x"][moved & ok], c0["y"][moved & ok], res, res)
    adopted = pixel_hist(c1["x"][moved & ok], c1["y"][moved & ok], res, res)
    border = (moved & ~ok).sum()
    # marks sit where the chains WERE ...
    assert np.abs(got - left).sum() <= 2.0 * border + 1e-6 * moved.sum(), (np.abs(got - left).sum(), border)
    # ... which is not where they went (the misreading of rounds 1-2 would put them there)
    assert np.abs(got - adopted).sum() > 0.2 * moved.sum()
    ctx.close()


@pytest.mark.parametrize("name,kw,tol", [
    ("cornell_c2", dict(type="orbital"), 0.06),
    ("cornell_c2", dict(type="green"), 0.06),
    ("cornell_c2", dict(type="mira", timid_after_large=1), 0.06),
    ("caustic_c5", dict(technique="mmlt", type="orbital", max_depth=6, fix_emitter_path=1, luminance_samples=1000), 0.10),
    ("glass_sphere", dict(technique="bdpt", type="orbital"), 0.12),
], ids=["path-orbital", "path-green", "path-mira-timid", "config5-mmlt", "bdpt"])
def test_map_matches_the_oracle_pixel_by_pixel(pkg, ob, name, kw, tol, native_lib):
    """Device and oracle run the same addressed chains from the same seeds (the oracle is handed the device's seed picks:
    its own differ where fp32 luminances shift the CDF): the two maps agree pixel by pixel except for
    the marks of the few chains that left the oracle's trajectory (an acceptance decided differently in fp32: <= 3-5 % of
    chains after 32 mutations, each displacing its later marks). Tolerance = L1 distance / total mass, stated per case."""
    res = 64
    sd = getattr(pkg.scenes, name)(res)
    n_chains, n_mut = 4096, 32
    cfg, ctx, orc = make(pkg, ob, sd, work_units=n_chains, **kw)
    assert ctx.seed(0x5EED) == 1.0                                       # drmlt.cpp:550-552
    assert orc.seed_with_indices(0x5EED, ctx.seed_indices()) == 1.0
    ctx.run(n_chains * n_mut)
    orc.run(n_chains * n_mut, 8)
    fg, fo = ctx.film().astype(np.float64), orc.film().astype(np.float64)
    assert fg[..., 2].max() == 0 and fo[..., 2].max() == 0
    for ch in (0, 1):
        assert fo[..., ch].sum() > 100
        assert fg[..., ch].sum() == pytest.approx(fo[..., ch].sum(), rel=0.05)
        d = np.abs(fg[..., ch] - fo[..., ch]).sum() / fo[..., ch].sum()
        assert d < tol, (ch, d)
    img = ctx.develop()
    np.testing.assert_allclose(img, fg.astype(np.float32), rtol=1e-6)    # develop does not rescale a map (:834-839)
    ctx.close(); orc.close()


def test_mixture_draws_no_map(pkg, ob, native_lib):
    """processMixture has no acceptance-map branch (drmlt_proc.cpp:161-380): radiance splats whatever the flag says."""
    sd = pkg.scenes.cornell_c2(32)
    cfg, ctx, orc = make(pkg, ob, sd, work_units=1024, type="orbital", use_mixture=1)
    ctx.seed(3); orc.seed(3)
    ctx.run(1024 * 16); orc.run(1024 * 16, 4)
    fg, fo = ctx.film().astype(np.float64), orc.film().astype(np.float64)
    assert fg[..., 2].sum() > 0 and fo[..., 2].sum() > 0
    assert fg.sum() == pytest.approx(fo.sum(), rel=5e-3)
    ctx.close(); orc.close()
