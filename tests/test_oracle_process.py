"""The chain loops on an analytic 2-D target: every delayed-rejection flavour must leave the target
distribution invariant (detailed balance of Eq. 5/7/11/14), and the statistics counters must obey the
identities implied by drmlt_proc.cpp:715-768."""
import numpy as np
import pytest

GREEN, MIRA, ORBITAL = 0, 1, 2
W = H = 24


def target_hist(ob):
    xs = (np.arange(W * 4) + 0.5) / (W * 4)
    g = np.array([[ob.toy_target(x, y) for x in xs] for y in xs])
    g = g.reshape(H, 4, W, 4).mean(axis=(1, 3))
    return g / g.sum()


@pytest.mark.parametrize("type_,mixture,timid", [
    (ORBITAL, False, False), (GREEN, False, False), (MIRA, False, False),
    (ORBITAL, False, True), (GREEN, True, False), (MIRA, False, True)])
def test_stationary_distribution(ob, abi, type_, mixture, timid):
    n_chains, n_mut = 64, 6000
    hist, st = ob.toy_run(abi, type_, mixture, timid, 0.3, 1 / 64, 0.1, 0xC0FFEE + type_, n_chains, n_mut, W, H)
    total = hist.sum()
    # every mutation deposits total weight 1 of unit-luminance splats (times the box weight^2)
    assert total == pytest.approx(n_chains * n_mut * 0.99998 ** 2, rel=1e-3)
    got = hist / total
    want = target_hist(ob)
    # chains are correlated: compare with a tolerance set by an effective sample size
    err = np.abs(got - want).sum()
    assert err < 0.08, "L1 distance to the target %.3f" % err
    # and no systematic tilt between the two modes
    left = got[:, : W // 2].sum()
    assert left == pytest.approx(want[:, : W // 2].sum(), abs=0.04)


def test_counter_identities(ob, abi):
    hist, st = ob.toy_run(abi, ORBITAL, False, False, 0.3, 1 / 64, 0.1, 5, 32, 2000, W, H)
    M = st.mutations
    assert M == 32 * 2000
    assert st.first_base == M and st.large_base + st.bold_base == M
    assert st.first_acc == st.large_acc + st.bold_acc
    assert st.second_large_base == 0                      # no timid step after a large one by default
    assert st.second_base == st.bold_base - st.bold_acc  # second stage runs exactly when a bold step is rejected
    assert st.overall_base == M + st.second_base and st.overall_acc == st.first_acc + st.second_acc
    assert st.accepted == st.overall_acc
    assert st.path_evals == M + st.second_base
    assert abs(st.large_base / M - 0.3) < 0.01


def test_timid_after_large_adds_second_stages(ob, abi):
    _, a = ob.toy_run(abi, ORBITAL, False, False, 0.3, 1 / 64, 0.1, 5, 16, 1000, W, H)
    _, b = ob.toy_run(abi, ORBITAL, False, True, 0.3, 1 / 64, 0.1, 5, 16, 1000, W, H)
    assert a.second_large_base == 0 and b.second_large_base == b.large_base - b.large_acc > 0


def test_green_counts_reverse_evaluations(ob, abi):
    _, st = ob.toy_run(abi, GREEN, False, False, 0.3, 1 / 64, 0.1, 9, 16, 1000, W, H)
    assert st.mutations + st.second_base <= st.path_evals <= st.mutations + 2 * st.second_base


def test_mixture_counters(ob, abi):
    _, st = ob.toy_run(abi, MIRA, True, False, 0.3, 1 / 64, 0.1, 9, 16, 2000, W, H)
    M = st.mutations
    assert st.overall_base == M and st.first_base + st.second_base == M
    # second-stage proposals are tried for half of the non-large mutations (drmlt_proc.cpp:296-299)
    assert st.second_base / (M - st.large_base) == pytest.approx(0.5, abs=0.02)


# ---- acceptance map: the reference's order of operations, replayed literally -------------------------------------------
TAG_COIN = 2


def _mark(img, pos, colour, w, h):
    """ImageBlock::put with the box filter (imageblock.h:150-216; radius 0.5 + 1e-5, 31-entry table, normalised)."""
    r = 0.5 + 1e-5
    wt = (1.0 / (2 * r)) ** 2
    px, py = pos[0] - 0.5, pos[1] - 0.5
    for y in range(max(int(np.ceil(py - r)), 0), min(int(np.floor(py + r)), h - 1) + 1):
        for x in range(max(int(np.ceil(px - r)), 0), min(int(np.floor(px + r)), w - 1) + 1):
            img[y, x] += wt * np.asarray(colour, dtype=np.float64)


def _replay_reference_order(ob, x0, chain, n_mut, seed, p_large, sigma, scale_second, w, h, mark_adopted=False):
    """drmlt_proc.cpp:518-771 for type=orbital in the test's own words. `current`, `first`, `second` are the reference's
    `current`, `proposed.first`, `proposed.second` (unique_ptr<SplatList>): on acceptance the pointers are SWAPPED and
    the mark is made through the PROPOSAL's pointer -- which by then owns the list that was current (:693-709).
    mark_adopted=True is the misreading rounds 1-2 shipped (mark the adopted state); the test shows it differs."""
    img = np.zeros((h, w, 3))

    def splat_list(u):
        return dict(pos=(u[0] * w, u[1] * h), lum=ob.toy_target(u[0], u[1]))

    x = np.array(x0, dtype=np.float64)
    current = splat_list(x)
    first, second = dict(pos=None, lum=0.0), dict(pos=None, lum=0.0)
    events = []
    for m in range(n_mut):
        coins = ob.uniforms(seed, chain, TAG_COIN, m, 0, 3).astype(np.float64)
        large = coins[0] < p_large
        tr = ob.sampler_trace(ORBITAL, sigma, scale_second, 64, seed, chain, m, large, x, 2)
        first = splat_list(tr["y"])                                       # sampleSplats(*proposed.first)
        a1 = min(1.0, first["lum"] / current["lum"])
        acc1 = a1 >= 1 or coins[1] < a1
        do_second = (not acc1) and (not large)
        acc2 = False
        if do_second:
            second = splat_list(tr["z"])                                  # sampleSplats(*proposed.second)
            if second["lum"] < first["lum"]:
                acc2 = False
            elif second["lum"] >= current["lum"]:
                acc2 = True
            else:
                a2 = (second["lum"] - first["lum"]) / (current["lum"] - first["lum"])
                acc2 = a2 >= 1 or coins[2] < a2
        if acc1:
            first, current = current, first                               # proposed.first.swap(current)
            if not large:
                _mark(img, (current if mark_adopted else first)["pos"], (1, 0, 0), w, h)   # splatAcceptanceOnly(proposed.first.get(), 0)
            x = tr["acc1"]
            events.append("L" if large else "1")
        elif acc2:
            second, current = current, second                             # proposed.second.swap(current)
            _mark(img, (current if mark_adopted else second)["pos"], (0, 1, 0), w, h)      # splatAcceptanceOnly(proposed.second.get(), 1)
            x = tr["acc2"]
            events.append("2")
        else:
            events.append("-")
    return img, "".join(events)


def test_acceptance_map_marks_the_state_being_left(ob):
    """Two chains, 48 mutations each, on the analytic target. The oracle's map must equal the literal replay of the
    reference's swap-then-mark order (the mark lands on the pixel the chain LEAVES), and must differ from the map the
    'mark the adopted state' reading produces."""
    w = h = 128   # (pixels of 1/128: a bold step of up to 1.9/64 usually crosses one)
    seed, n_chains, n_mut, p_large, sigma, ss = 0xACCE97, 2, 48, 0.3, 1 / 64, 0.1
    got, x0 = ob.toy_amap(ORBITAL, p_large, sigma, ss, seed, n_chains, n_mut, w, h)
    want, wrong = np.zeros_like(got), np.zeros_like(got)
    log = []
    for c in range(n_chains):
        img, ev = _replay_reference_order(ob, x0[c], c, n_mut, seed, p_large, sigma, ss, w, h)
        want += img
        log.append(ev)
        wrong += _replay_reference_order(ob, x0[c], c, n_mut, seed, p_large, sigma, ss, w, h, mark_adopted=True)[0]
    # the example exercises every kind of event: bold first-stage acceptances, second-stage acceptances, large steps, rejections
    assert all(k in "".join(log) for k in "12L-"), log
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    assert got[..., 2].sum() == 0 and got[..., 0].sum() > 0 and got[..., 1].sum() > 0
    # same totals, different pixels: what rounds 1-2 could not see with sums alone
    assert got.sum() == pytest.approx(wrong.sum(), rel=1e-12)
    assert np.abs(got - wrong).sum() > 0.25 * got.sum()


def test_acceptance_map_first_mark_is_the_seed_pixel(ob):
    """The first mark of a chain sits at the pixel of its SEED state (the state the first accepted bold move leaves)."""
    w = h = 16
    seed, p_large, sigma, ss = 0xACCE98, 0.0, 1 / 64, 0.1     # no large steps: every acceptance leaves a mark
    for n_mut in range(1, 12):
        got, x0 = ob.toy_amap(ORBITAL, p_large, sigma, ss, seed, 1, n_mut, w, h)
        if got.sum() > 0:
            break
    assert got.sum() == pytest.approx((1.0 / 1.00002) ** 2)   # exactly one mark so far
    y, x = np.unravel_index(np.argmax(got.sum(axis=2)), (h, w))
    assert (x, y) == (int(x0[0, 0] * w), int(x0[0, 1] * h))
