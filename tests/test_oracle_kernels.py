"""Oracle pinning: transition kernels vs the reference's own tools/transition.h (golden vectors produced by
tests/golden/make_transition_golden.py from the reference compiled in place), Philox known answers, and
chi^2 / KS tests of every kernel's sampler against its pdf (the reference's test_chisquare.cpp strategy)."""
import json
import math
import os

import numpy as np
import pytest
from scipy import stats

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "transition_kat.json")


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def test_golden_stream_is_the_oracle_stream(ob, golden):
    u = ob.uniforms(golden["seed"], 0, 3, 0, 0, 2 * golden["n"])
    assert np.array_equal(u, np.asarray(golden["uniforms"], dtype=np.float32))


@pytest.mark.parametrize("prec,tol", [(64, 1e-13), (32, 1e-5)])  # f32: the reference mixes double constants into float math
def test_kernels_match_reference_transition_h(ob, golden, prec, tol):
    for k in golden["kernels"]:
        ref = k["f%d" % prec]
        s = ob.kernel_sample(k["kind"], k["p0"], k["p1"], prec, golden["seed"], golden["n"])
        r = np.asarray(ref["samples"])
        scale = max(1e-30, np.abs(r).max())
        assert np.abs(s - r).max() <= tol * scale, k["name"]
        pdf, logpdf = ob.kernel_pdf(k["kind"], k["p0"], k["p1"], prec, k["du"])
        rp = np.asarray(ref["pdf"])
        assert np.allclose(pdf, rp, rtol=50 * tol, atol=0), k["name"]
        for a, b in zip(logpdf, ref["logpdf"]):
            if b is None:
                assert math.isinf(a)
            else:
                assert abs(a - b) <= 50 * tol * max(1.0, abs(b)), k["name"]


def test_philox_known_answers(ob):
    # Random123 kat_vectors, philox4x32-10
    assert list(ob.philox(0, 0, 0, 0, 0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert list(ob.philox(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)) == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert list(ob.philox(0xa4093822, 0x299f31d0, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_stream_is_uniform(ob):
    u = ob.uniforms(7, 3, 3, 11, 0, 200000)
    assert 0.0 <= u.min() and u.max() < 1.0
    assert stats.kstest(u, "uniform").pvalue > 1e-3
    # addressed: same (tag, major, idx) gives the same number whatever was drawn before
    assert np.array_equal(ob.uniforms(7, 3, 3, 11, 1000, 16), u[1000:1016])


CASES = [
    ("gaussian", 0, 0.1 / 64, 0.0, lambda x, p0, p1: stats.norm(0, p0).cdf(x)),
    ("kelemen", 1, 1 / 1024, 1 / 64, None),
    ("wrapped_cauchy", 3, math.exp(-0.25), 0.0, None),
]


@pytest.mark.parametrize("name,kind,p0,p1,cdf", CASES)
def test_kernel_sampler_matches_pdf(ob, name, kind, p0, p1, cdf):
    n = 200000
    s = ob.kernel_sample(kind, p0, p1, 64, 1234, n)
    if kind == 0:
        assert stats.kstest(s, lambda x: cdf(x, p0, p1)).pvalue > 1e-3
        return
    lo, hi = (-p1, p1) if kind == 1 else (-math.pi, math.pi)
    edges = np.linspace(lo, hi, 65)
    hist, _ = np.histogram(s, bins=edges)
    if kind == 1:  # closed form: |d| is log-uniform on [s1, s2]
        def cdf_k(x):
            a = np.clip(np.abs(x), p0, p1)
            m = np.log(a / p0) / math.log(p1 / p0)
            return np.where(x < 0, 0.5 - 0.5 * m, 0.5 + 0.5 * m)
        expected = cdf_k(edges[1:]) - cdf_k(edges[:-1])
        # ... and the pdf the kernel reports is the derivative of that cdf
        xs = np.array([0.002, -0.005, 0.01, 0.0151])
        pdf, _ = ob.kernel_pdf(kind, p0, p1, 64, xs)
        h = 1e-7
        assert np.allclose(pdf, (cdf_k(xs + h) - cdf_k(xs - h)) / (2 * h), rtol=1e-4)
    else:  # integrate the kernel's own pdf over each bin
        xs = np.linspace(lo, hi, 64 * 200 + 1)
        pdf, _ = ob.kernel_pdf(kind, p0, p1, 64, xs)
        cell = 0.5 * (pdf[1:] + pdf[:-1]) * np.diff(xs)
        expected = cell.reshape(64, 200).sum(axis=1)
    assert abs(expected.sum() - 1.0) < 5e-3, "pdf does not integrate to 1"
    expected = expected / expected.sum() * n
    keep = expected > 5
    chi2 = ((hist[keep] - expected[keep]) ** 2 / expected[keep]).sum()
    assert stats.chi2(keep.sum() - 1).sf(chi2) > 1e-3, name
    assert np.all(hist[~keep] <= 30)
    if kind == 1:  # the Kelemen hole and support
        assert np.abs(s).min() >= p0 * (1 - 1e-12) and np.abs(s).max() <= p1 * (1 + 1e-12)


def test_kelemen_logpdf_is_minus_inf_outside_support(ob):
    pdf, logpdf = ob.kernel_pdf(1, 1 / 1024, 1 / 64, 64, [0.0, 1 / 2048, 1 / 32])
    assert np.all(pdf == 0) and np.all(np.isneginf(logpdf))
