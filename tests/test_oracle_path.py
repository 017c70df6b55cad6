"""The restated estimator (path.cpp:123-315) and the chain loops on real scenes: an analytic form-factor
check, consistency of MLT images with independent sampling of the same integrand (pathsampler.cpp:529-567
is the common f(u)), the bootstrap normalisation, and the seed-replay sanity check (drmlt_proc.cpp:509-512)."""
import math

import numpy as np
import pytest


def lum(img):
    return img @ np.array([0.212671, 0.715160, 0.072169])


def test_one_bounce_irradiance_matches_form_factor(pkg, abi, ob):
    """maxDepth=2: camera -> floor -> light. Radiance = rho/pi * L * integral(cos cos' / r^2 dA)."""
    sc = pkg.scenes
    sd = sc.SceneData("ff")
    rho, L, half, hgt = 0.6, 5.0, 0.3, 1.2
    grey, black = sd.diffuse(rho), sd.diffuse(0.0)
    sd.rectangle(sc.rotate("x", -90) @ sc.scale(50.0), grey)                                   # floor y = 0
    sd.rectangle(sc.translate(0, hgt, 0) @ sc.rotate("x", 90) @ sc.scale(half), black, radiance=L)  # light facing down
    sd.set_camera(sc.lookat((0.0, 3.0, 4.0), (0.6, 0.0, 0.2), (0, 1, 0)), 1.0, 8, 8)              # narrow fov: one floor spot
    cfg = abi.make_config(type="orbital", max_depth=2, rr_depth=5, direct_samples=-1, work_units=4,
                          luminance_samples=1000, sample_count=1)
    o = ob.Oracle(abi, cfg, sd, 64)
    img = o.render_pt(4000, seed=3, nthreads=8)
    got = lum(img).mean()
    # form factor by quadrature at the looked-at point
    x0 = np.array([0.6, 0.0, 0.2])
    g = (np.arange(400) + 0.5) / 400 * 2 * half - half
    X, Z = np.meshgrid(g, g)
    d = np.stack([X - x0[0], np.full_like(X, hgt), Z - x0[2]], -1)
    r2 = (d ** 2).sum(-1)
    cos = hgt / np.sqrt(r2)
    E = (cos * cos / r2).sum() * (2 * half / 400) ** 2 * L
    want = rho / math.pi * E
    assert got == pytest.approx(want, rel=0.02)


@pytest.fixture(scope="module")
def c1_reference(pkg, abi, ob):
    sd = pkg.scenes.cornell_c1(16)
    cfg = abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=64, luminance_samples=20000,
                          sample_count=1)
    o = ob.Oracle(abi, cfg, sd, 64)
    return sd, o.render_pt(6000, seed=11, nthreads=8)


@pytest.mark.parametrize("kw", [
    dict(type="orbital"), dict(type="green"), dict(type="mira"),
    dict(type="orbital", use_mixture=1), dict(type="orbital", timid_after_large=1),
    dict(type="mira", algo=1), dict(type="mira", algo=1, kelemen_style_weights=0),
    dict(type="mira", algo=1, kelemen_style_mutation=0)])
def test_mlt_image_agrees_with_independent_sampling(pkg, abi, ob, c1_reference, kw):
    sd, ref = c1_reference
    spp = 3000
    cfg = abi.make_config(max_depth=8, direct_samples=-1, work_units=64, luminance_samples=20000, sample_count=spp, **kw)
    o = ob.Oracle(abi, cfg, sd, 64)
    b = o.seed(0x5EED)
    assert b == pytest.approx(lum(ref).mean(), rel=0.05)          # b estimates the mean image luminance
    o.run(16 * 16 * spp, 8)
    img = o.develop()
    assert lum(img).mean() == pytest.approx(b, rel=1e-4)          # develop() normalises to b
    lr, li = lum(ref), lum(img)
    mask = lr > 0.05 * lr.mean()
    rel = np.abs(li - lr)[mask] / lr[mask]
    assert np.median(rel) < 0.06 and rel.mean() < 0.10, (np.median(rel), rel.mean())
    # relative MSE with the BASELINE.md epsilon
    rmse = np.mean((li - lr) ** 2 / (lr ** 2 + 1e-2 * lr.mean() ** 2))
    assert rmse < 0.03


def test_float_build_tracks_double_build(pkg, abi, ob):
    sd = pkg.scenes.cornell_c2(32)
    cfg = abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=256, luminance_samples=20000,
                          sample_count=4)
    o64, o32 = ob.Oracle(abi, cfg, sd, 64), ob.Oracle(abi, cfg, sd, 32)
    u = np.random.default_rng(2).random((4000, 50), dtype=np.float32)
    a, b = o64.eval_paths(u), o32.eval_paths(u)
    same = a["n_dims"] == b["n_dims"]
    assert same.mean() > 0.995
    rel = np.abs(a["luminance"] - b["luminance"])[same] / np.maximum(a["luminance"][same], 1e-3)
    assert np.quantile(rel, 0.99) < 1e-3


def test_seed_replay_and_seed_distribution(pkg, abi, ob):
    sd = pkg.scenes.cornell_c2(32)
    cfg = abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=2000, luminance_samples=20000,
                          sample_count=1)
    o = ob.Oracle(abi, cfg, sd, 64)
    o.seed(77)                                   # raises if any replayed luminance disagrees with its seed
    cur, u = o.chain_state(50)
    lum_all = o.bootstrap_lum(77, 0, 20000)
    # seeds are drawn with probability proportional to luminance (pathsampler.cpp:946-954)
    assert cur["luminance"].mean() == pytest.approx((lum_all ** 2).sum() / lum_all.sum(), rel=0.08)
    assert np.all(cur["luminance"] > 0) and np.all((u >= 0) & (u < 1))
    assert np.all(np.diff(np.searchsorted(np.sort(lum_all), cur["luminance"])) > -len(lum_all))  # finite, in range


def test_separate_direct_excludes_first_bounce_direct_light(pkg, abi, ob):
    sd = pkg.scenes.cornell_c1(8)
    u = np.random.default_rng(5).random((3000, 50), dtype=np.float32)
    full = ob.Oracle(abi, abi.make_config(type="orbital", max_depth=2, direct_samples=-1, work_units=1), sd, 64)
    nodirect = ob.Oracle(abi, abi.make_config(type="orbital", max_depth=2, direct_samples=16, work_units=1), sd, 64)
    assert full.eval_paths(u)["luminance"].sum() > 0
    assert nodirect.eval_paths(u)["luminance"].sum() == 0   # depth-2 paths ARE the direct component
