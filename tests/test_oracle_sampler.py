"""DRMLTSampler restatement: structural properties of the stage kernels
(reference drmlt_sampler.cpp:231-414, drmlt_sampler.h:140-144, pssmlt_utils.h:27-77)."""
import math

import numpy as np
import pytest

S1, S2 = 1 / 1024, 1 / 64
GREEN, MIRA, ORBITAL = 0, 1, 2


def _x(dim, seed=0):
    return np.random.default_rng(seed).random(dim) * 0.8 + 0.1  # away from the wrap boundaries


def unwrap_close(a, b):
    return np.abs(a - b)


def test_find_max_dimensions(ob):
    L = ob.lib()
    assert L.oracle_find_max_dim(8, 5) == 50      # (8+2)*(4+1), SURVEY 8
    assert L.oracle_find_max_dim(8, 8) == 40      # no roulette dimension when rrDepth >= maxDepth
    assert L.oracle_find_max_dim(5, 3) == 36      # (5+2)*5 = 35 -> rounded up to a full pair
    assert L.oracle_find_max_dim(3, 10) == 20


@pytest.mark.parametrize("type_", [GREEN, MIRA])
def test_iid_stages(ob, type_):
    dim, used = 20, 20
    x = _x(dim)
    t = ob.sampler_trace(type_, 1 / 64, 0.1, 64, 99, 5, 17, False, x, used)
    dy = np.abs(t["y"] - x)
    assert np.all(dy >= S1 * (1 - 1e-9)) and np.all(dy <= S2 * (1 + 1e-9))       # Kelemen support
    dz = t["z"] - x                                                             # stage 2 is centred on x
    assert np.abs(dz).max() < 8 * 0.1 / 64 and np.std(dz) == pytest.approx(0.1 / 64, rel=0.5)
    if type_ == GREEN:                                                          # y* = z - (y - x)
        assert np.allclose(t["ystar"], t["z"] - (t["y"] - x), atol=1e-12)
    assert np.allclose(t["acc1"][:used], t["y"]) and np.allclose(t["acc2"][:used], t["z"])


def test_mira_transition_ratio(ob):
    dim, used = 20, 12
    x = _x(dim, 3)
    t = ob.sampler_trace(MIRA, 1 / 64, 0.1, 64, 5, 2, 40, False, x, used)
    y, z = t["y"], t["z"]

    def logq(d):
        d = abs(d)
        return -math.inf if (d < S1 or d > S2) else -math.log(2 * d * math.log(S2 / S1))
    # the reference sums i < max(dimStage1, dimStage2) with dimStage = LAST INDEX used (drmlt_sampler.cpp:237,405)
    n = used - 1
    num = sum(logq(z[i] - y[i]) for i in range(n))
    den = sum(logq(x[i] - y[i]) for i in range(n))
    expect = math.exp(num - den) if num > -math.inf else 0.0
    assert t["ratio"] == pytest.approx(expect, rel=1e-9, abs=1e-300)


def test_orbital_pairs(ob):
    dim, used = 24, 24
    x = _x(dim, 1)
    t = ob.sampler_trace(ORBITAL, 1 / 64, 0.1, 64, 7, 1, 3, False, x, used)
    y, z = t["y"], t["z"]
    ry = np.hypot(y[0::2] - x[0::2], y[1::2] - x[1::2])          # pairwise Kelemen radius, scaled 1.9
    assert np.all(ry >= 1.9 * S1 * (1 - 1e-9)) and np.all(ry <= 1.9 * S2 * (1 + 1e-9))
    rz = np.hypot(z[0::2] - y[0::2], z[1::2] - y[1::2])          # orbital: z stays on the circle |x - y| about y
    assert np.allclose(rz, ry, rtol=1e-9)
    # rotation angle is the wrapped-Cauchy draw: same angle reproduces from the kernel stream
    theta = np.arctan2(z[1::2] - y[1::2], z[0::2] - y[0::2]) - np.arctan2(x[1::2] - y[1::2], x[0::2] - y[0::2])
    theta = (theta + math.pi) % (2 * math.pi) - math.pi
    assert np.abs(theta).max() <= math.pi + 1e-9 and np.abs(np.median(theta)) < 1.0


def test_large_step_is_uniform_and_addressed(ob):
    dim = 16
    x = _x(dim)
    t = ob.sampler_trace(ORBITAL, 1 / 64, 0.1, 64, 11, 4, 9, True, x, dim)
    u = ob.uniforms(11, 4, 3, 9, 0, dim)  # TAG_S1 stream of (chain 4, mutation 9): dimension k <-> draw k
    assert np.allclose(t["y"], u)


def test_wrap_reflects(ob):
    x = np.array([0.0005, 0.9995] * 4)
    hits = 0
    for m in range(64):
        t = ob.sampler_trace(GREEN, 1 / 64, 0.1, 64, 3, 0, m, False, x, 8)
        y = t["y"]
        assert np.all((y >= 0) & (y <= 1))
        hits += int(np.any(np.abs(y - x) < S1 * 0.999))  # reflection shortens the apparent step
    assert hits > 0


def test_float_and_double_builds_agree(ob):
    x = _x(20, 5)
    for type_ in (GREEN, MIRA, ORBITAL):
        a = ob.sampler_trace(type_, 1 / 64, 0.1, 64, 21, 6, 2, False, x, 20)
        b = ob.sampler_trace(type_, 1 / 64, 0.1, 32, 21, 6, 2, False, x, 20)
        assert np.abs(a["y"] - b["y"]).max() < 2e-6
        assert np.abs(a["z"] - b["z"]).max() < (2e-4 if type_ == ORBITAL else 2e-6)  # acos round trip in fp32
