"""Two-stage MLT (`twoStage`, drmlt.cpp:278,406-418): luminance image of a reduced first stage
(BidirectionalUtils::mltLuminancePass, util.cpp:96-199) weights the second stage's splat lists
(SplatList::normalize, pathsampler.cpp:1001-1020) and is multiplied back in develop (drmlt_proc.cpp:824-845)."""
import numpy as np
import pytest

LUMW = np.array([0.212671, 0.715160, 0.072169])


def resample_1d(src, n_dst):
    """Independent numpy statement of Resampler (core/rfilter.h:123-198) for the gaussian filter, EClamp."""
    n_src = len(src)
    if n_src == n_dst:
        return src.copy()
    radius, inv_scale = 2.0, 1.0
    if n_dst < n_src:
        scale = n_src / n_dst
        inv_scale, radius = 1 / scale, radius * scale
    taps = int(np.ceil(radius * 2))
    out = np.zeros(n_dst)
    for i in range(n_dst):
        center = (i + 0.5) / n_dst * n_src
        start = int(np.floor(center - radius + 0.5))
        pos = (start + np.arange(taps) + 0.5 - center) * inv_scale
        w = np.maximum(0.0, np.exp(-2.0 * pos ** 2) - np.exp(-2.0 * 4.0))
        idx = np.clip(start + np.arange(taps), 0, n_src - 1)
        out[i] = max(0.0, (src[idx] * (w / w.sum())).sum())
    return out


def test_luminance_map_resampler(ob):
    rng = np.random.default_rng(3)
    small = rng.random((4, 6, 3)).astype(np.float32)
    got = ob.luminance_map(small, 96, 64)
    lum = small.astype(np.float64) @ LUMW
    tmp = np.stack([resample_1d(row, 96) for row in lum])            # X pass, then Y pass (bitmap.cpp:2258-2330)
    want = np.stack([resample_1d(tmp[:, x], 64) for x in range(96)], axis=1)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7)
    flat = ob.luminance_map(np.full((3, 3, 3), 0.25, np.float32), 48, 48)
    np.testing.assert_allclose(flat, 0.25 * LUMW.sum(), rtol=1e-6)    # weights are normalised per target sample
    same = ob.luminance_map(small, 6, 4)
    np.testing.assert_allclose(same, lum, rtol=1e-6)


def test_normalize_with_importance_and_develop(pkg, abi, ob):
    """Chains sample f / importance; develop multiplies the map back: the image stays unbiased, b unchanged."""
    sd = pkg.scenes.cornell_c2(16)
    ref = ob.Oracle(abi, abi.make_config(max_depth=6, rr_depth=100, work_units=4, direct_samples=-1), sd, 64) \
        .render_pt(3000, seed=7, nthreads=8)
    imp = np.maximum(ob.luminance_map(ref.reshape(4, 4, 4, 4, 3).mean((1, 3)), 16, 16), 1e-3)
    # (the reference's seeding rule: the same seeds with and without the map, so the states can be compared one to one)
    cfg = abi.make_config(technique="path", type="orbital", max_depth=6, work_units=2048, direct_samples=-1,
                          luminance_samples=100000, seed_rule="reference")
    plain = ob.Oracle(abi, cfg, sd, 64)
    b0 = plain.seed(99)
    o = ob.Oracle(abi, cfg, sd, 64)
    o.set_importance_map(imp)
    b = o.seed(99)
    assert b == b0                                                     # seeds use the unweighted luminance (:901-903)
    cur, _ = o.chain_state(34)
    cur0, _ = plain.chain_state(34)
    ix = np.clip(cur0["x"].astype(int), 0, 15); iy = np.clip(cur0["y"].astype(int), 0, 15)
    np.testing.assert_allclose(cur["luminance"], cur0["luminance"] / imp[iy, ix], rtol=1e-5)
    o.run(16 * 16 * 4000, 8)
    img = o.develop()
    blk = lambda a: a.reshape(4, 4, 4, 4, 3).mean((1, 3))
    assert np.abs(blk(img) - blk(ref)).mean() / ref.mean() < 0.05
    # the weighted chains spend their mutations more evenly over the image: film mass per pixel is flatter
    plain.run(16 * 16 * 4000, 8)
    f_w, f_p = o.film() @ LUMW, plain.film() @ LUMW
    assert f_w.std() / f_w.mean() < 0.8 * f_p.std() / f_p.mean()
    with pytest.raises(ob.OracleError, match="before seed"):
        o.set_importance_map(imp)


def _bdpt_dims(max_depth, rr_depth=5, direct_sampling=True):
    """[sensor S | emitter E | direct Dd] of a bdpt chain (device_bdpt.h / binding.eval_lists_bdpt)."""
    rr = max_depth + 1 - max(rr_depth, 0)
    S = 2 * (max_depth + 1) + max(rr, 0); S += S & 1
    E = 2 * max_depth + max(rr - 1, 0); E += E & 1
    return S + E + (2 * (2 * max_depth - 1) if direct_sampling else 0)


@pytest.mark.parametrize("tech", ["path", "bdpt"])
def test_seed_rules(pkg, abi, ob, tech):
    """drmlt_config.seed_rule. REFERENCE: seeds in proportion to lum(f) (pathsampler.cpp:901-905, the luminance read before
    SplatList::normalize(importanceMap)). TARGET (the product's default): in proportion to lum(f / importance), the chains' own
    target. b is the mean of lum(f) under both; a constant map makes the two rules pick the same samples."""
    sd = pkg.scenes.cornell_c2(16)
    kw = dict(technique=tech, type="orbital", max_depth=5, work_units=4096, direct_samples=-1, luminance_samples=60000)
    imp = np.tile(np.where((np.arange(16) + 0.5) / 16 < 0.5, 0.02, 1.0), (16, 1)).astype(np.float32)
    dim = 2 if tech == "path" else _bdpt_dims(5)
    picks, b, left = {}, {}, {}
    for rule in ("target", "reference"):
        o = ob.Oracle(abi, abi.make_config(seed_rule=rule, **kw), sd, 64)
        o.set_importance_map(imp)
        b[rule] = o.seed(5)
        picks[rule] = o.seed_indices()
        cur, _ = o.chain_state(dim)
        left[rule] = float((cur["x"] < 8).mean())
        o.close()
    assert b["target"] == b["reference"]
    # where do the bootstrap samples put their luminance? (the bootstrap stream's own samples, by position of their main splat)
    o = ob.Oracle(abi, abi.make_config(seed_rule="reference", **kw), sd, 64)
    o.seed(5)
    cur, _ = o.chain_state(dim)       # seeds ~ f: the share of chains on the left estimates the share of f there
    share_f = float((cur["x"] < 8).mean())
    o.close()
    assert abs(left["reference"] - share_f) < 1e-12                      # the map does not enter the reference's rule
    want = share_f / 0.02 / (share_f / 0.02 + (1 - share_f))             # f / importance: the left half weighs 50 x
    if tech == "path":                                                   # (a bdpt list spreads over pixels: only the direction is checked)
        assert abs(left["target"] - want) < 0.03, (left, want)
    # (bdpt: a list is drawn 50 x more often as soon as ANY of its splats -- light-image splats included -- lies on the left, wherever
    # its main splat is; the main splat's side therefore shifts less)
    assert left["target"] > left["reference"] + (0.2 if tech == "path" else 0.05)
    # a constant map: the same picks under both rules
    flat = np.full((16, 16), 0.37, dtype=np.float32)
    got = []
    for rule in ("target", "reference"):
        o = ob.Oracle(abi, abi.make_config(seed_rule=rule, **kw), sd, 64)
        o.set_importance_map(flat)
        o.seed(5)
        got.append(o.seed_indices())
        o.close()
    assert (got[0] == got[1]).mean() > 0.999                             # (a pick on a CDF step can move by one sample in floating point)
