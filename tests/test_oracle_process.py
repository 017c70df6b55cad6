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
