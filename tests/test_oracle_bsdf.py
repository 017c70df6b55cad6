"""Rough conductor restated (roughconductor.cpp:258-409, microfacet.h): the reference's own chi^2 strategy
(test_chisquare.cpp:391-623): sample() must be distributed according to pdf(), and sample()'s weight must equal
eval()/pdf()."""
import numpy as np
import pytest
from scipy import stats

ETA, K = (0.2004, 0.9240, 1.1022), (3.9129, 2.4528, 2.1421)


def hemisphere_grid(nt=64, nphi=128):
    ct = (np.arange(nt) + 0.5) / nt                      # uniform in cos(theta): equal solid angle cells
    ph = (np.arange(nphi) + 0.5) / nphi * 2 * np.pi
    C, P = np.meshgrid(ct, ph, indexing="ij")
    st = np.sqrt(1 - C * C)
    return np.stack([st * np.cos(P), st * np.sin(P), C], -1).reshape(-1, 3), 2 * np.pi / (nt * nphi)


@pytest.mark.parametrize("ggx", [False, True])
@pytest.mark.parametrize("alpha", [0.1, 0.4])
@pytest.mark.parametrize("theta_i", [5.0, 45.0, 75.0])
def test_sample_matches_pdf_and_eval(ob, ggx, alpha, theta_i):
    t = np.radians(theta_i)
    wi = np.array([np.sin(t) * 0.8, np.sin(t) * 0.6, np.cos(t)])
    n = 120000
    rng = np.random.default_rng(int(alpha * 100) + int(theta_i) + ggx)
    s = ob.roughconductor(ggx, alpha, ETA, K, wi, sxy=rng.random((n, 2)))
    ok = s["spdf"] > 0
    # (1) weight * pdf = eval (f cos) for the sampled directions
    e = ob.roughconductor(ggx, alpha, ETA, K, wi, wo=s["wo"][ok])
    assert np.allclose(e["pdf"], s["spdf"][ok], rtol=1e-6)
    assert np.allclose(s["weight"][ok] * s["spdf"][ok, None], e["eval"], rtol=1e-5, atol=1e-12)
    # (2) pdf integrates to the probability that sampling succeeds (samples below the horizon are dropped)
    dirs, dA = hemisphere_grid(256, 512)
    g = ob.roughconductor(ggx, alpha, ETA, K, wi, wo=dirs)
    assert g["pdf"].sum() * dA == pytest.approx(ok.mean(), abs=0.02)
    # (3) chi^2 on a coarse (cos theta, phi) histogram
    nt, nphi = 8, 16
    wo = s["wo"][ok]
    it = np.minimum((wo[:, 2] * nt).astype(int), nt - 1)
    ip = ((np.arctan2(wo[:, 1], wo[:, 0]) % (2 * np.pi)) / (2 * np.pi) * nphi).astype(int) % nphi
    hist = np.bincount(it * nphi + ip, minlength=nt * nphi).astype(float)
    fine = g["pdf"].reshape(256, 512)
    expected = fine.reshape(nt, 256 // nt, nphi, 512 // nphi).sum(axis=(1, 3)).reshape(-1) * dA * n
    keep = expected > 20
    # the 2-D midpoint quadrature of a peaky lobe carries a few % of error: fold it into the variance
    chi2 = (((hist - expected) ** 2) / (expected + (0.03 * expected) ** 2))[keep].sum()
    assert stats.chi2(keep.sum()).sf(chi2) > 1e-4, (chi2, keep.sum())


def test_energy_is_bounded(ob):
    for ggx in (False, True):
        for theta in (10, 60, 85):
            t = np.radians(theta)
            wi = np.array([np.sin(t), 0, np.cos(t)])
            s = ob.roughconductor(ggx, 0.3, (0.0, 0.0, 0.0), (1e3, 1e3, 1e3), wi, sxy=np.random.default_rng(1).random((50000, 2)))
            albedo = s["weight"].mean(axis=0)        # near-perfect mirror coating: albedo <= 1, close to it
            assert np.all(albedo <= 1.0 + 1e-6) and np.all(albedo > 0.6)
