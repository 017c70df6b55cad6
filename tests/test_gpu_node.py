"""Multi-GPU surface behind the C-ABI, exercised on ONE GPU (the box has one): pool seeding, two contexts / two node ranks
whose chains are exactly those of one big context, and the RCCL film exchange called from C++ (world size 1; the N > 1
arithmetic of the same code runs through the loopback transport and through the gloo test). The 1 -> 8 GPU curve itself is
unmeasured on hardware."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LUMW = np.array([0.212671, 0.715160, 0.072169])


def _cfg(pkg, n, **kw):
    base = dict(type="orbital", max_depth=8, rr_depth=5, direct_samples=-1, work_units=n, sample_count=4,
                luminance_samples=1000)
    base.update(kw)
    return pkg.abi.make_config(**base)


@pytest.mark.parametrize("kw", [dict(), dict(type="green"), dict(technique="mmlt", max_depth=6), dict(technique="bdpt", max_depth=6)],
                         ids=["orbital", "green", "mmlt", "bdpt"])
def test_two_contexts_equal_one_context_with_twice_the_chains(pkg, kw, native_lib):
    """Contexts A (chains [0, n)) and B (chains [n, 2n)) seeded from one pool run exactly the chains of one 2n-chain
    context: their films, summed on the device, develop to the same image (up to the order of the float atomics)."""
    sd = pkg.scenes.cornell_c2(64)
    n, total = 2048, 64 * 64 * 4
    big = pkg.Context(_cfg(pkg, 2 * n, **kw), sd)
    b_big = big.seed_pool(0x5EED, 0, 2 * n)
    big.run(total)
    a, b = pkg.Context(_cfg(pkg, n, **kw), sd), pkg.Context(_cfg(pkg, n, **kw), sd)
    ba, bb = a.seed_pool(0x5EED, 0, 2 * n), b.seed_pool(0x5EED, n, 2 * n)
    assert ba == bb == b_big
    a.run(total // 2); b.run(total // 2)
    # same chains: same current states, chain by chain
    dim = big.stats().max_dim if kw.get("technique") == "bdpt" else (27 if kw.get("technique") == "mmlt" else 34)
    (cb_, ub), (ca, ua), (cbb, ubb) = big.chain_state(dim), a.chain_state(dim), b.chain_state(dim)
    assert np.array_equal(ub[:n], ua) and np.array_equal(ub[n:], ubb)
    assert np.array_equal(cb_["luminance"][:n], ca["luminance"]) and np.array_equal(cb_["luminance"][n:], cbb["luminance"])
    sa, sb, sbig = a.stats(), b.stats(), big.stats()
    assert sa.accepted + sb.accepted == sbig.accepted and sa.rays + sb.rays == sbig.rays
    # films summed ON THE DEVICE (torch view of the two film buffers), then developed by context A
    import torch

    class Dev:
        def __init__(self, ptr, k):
            self.__cuda_array_interface__ = {"shape": (k,), "typestr": "<f4", "data": (ptr, False), "version": 2}

    fa = torch.as_tensor(Dev(a.film_device_ptr(), 64 * 64 * 3), device="cuda")
    fb = torch.as_tensor(Dev(b.film_device_ptr(), 64 * 64 * 3), device="cuda")
    torch.cuda.synchronize()
    fa += fb
    torch.cuda.synchronize()
    img2, img1 = a.develop(), big.develop()
    np.testing.assert_allclose(img2, img1, rtol=2e-4, atol=1e-6)
    assert (img2 @ LUMW).mean() == pytest.approx(b_big, rel=1e-5)


@pytest.mark.parametrize("kw", [dict(technique="bdpt", max_depth=6, no_direct_sampling=1), dict(technique="bdpt", max_depth=8, no_direct_sampling=0), dict(technique="mmlt", max_depth=6), dict()],
                         ids=["bdpt-nodirect", "bdpt-direct", "mmlt", "path"])
def test_a_chain_does_not_depend_on_its_wave_neighbours(pkg, kw, native_lib):
    """A RAGGED split of one pool (chains [0, 100) and [100, 356)): every chain of the second context sits in another lane, beside
    other chains, than in the one-context run -- and must end in the same state bit for bit. (bdpt hands the connections of all
    the chains of a wave out to all of its lanes and sums a chain's contributions by a segmented scan: the scan's tree depends on
    the chain's own cell count alone, which is what this test holds it to.)"""
    sd = pkg.scenes.glass_sphere(32)
    n_a, n_b, per_chain = 100, 256, 48
    total = n_a + n_b
    big = pkg.Context(_cfg(pkg, total, **kw), sd)
    big.seed_pool(0xBEEF, 0, total)
    big.run(total * per_chain)
    a, b = pkg.Context(_cfg(pkg, n_a, **kw), sd), pkg.Context(_cfg(pkg, n_b, **kw), sd)
    a.seed_pool(0xBEEF, 0, total); b.seed_pool(0xBEEF, n_a, total)
    a.run(n_a * per_chain); b.run(n_b * per_chain)
    dim = big.stats().max_dim if kw.get("technique") == "bdpt" else (27 if kw.get("technique") == "mmlt" else 34)
    (cb_, ub), (ca, ua), (cbb, ubb) = big.chain_state(dim), a.chain_state(dim), b.chain_state(dim)
    assert np.array_equal(ub[:n_a], ua) and np.array_equal(ub[n_a:], ubb)
    assert np.array_equal(cb_["luminance"][:n_a], ca["luminance"]) and np.array_equal(cb_["luminance"][n_a:], cbb["luminance"])
    sa, sb, sbig = a.stats(), b.stats(), big.stats()
    assert sa.accepted + sb.accepted == sbig.accepted and sa.rays + sb.rays == sbig.rays


@pytest.mark.parametrize("kw", [dict(technique="bdpt", max_depth=6, no_direct_sampling=0), dict(technique="mmlt", max_depth=6)], ids=["bdpt", "mmlt"])
def test_regrouping_the_waves_by_work_changes_no_chain(pkg, kw, native_lib, monkeypatch):
    """Between the launches of a call the host regroups the bidirectional kernels' waves by the evaluations every chain needed in the
    launch just done (chains parked on a glint share waves): many short launches with and without it end in the same states."""
    sd = pkg.scenes.caustic_c5(32)
    n, per_chain = 1000, 96
    res = []
    for off in (False, True):
        monkeypatch.setenv("DRMLT_SLICE", "16")                      # six launches, five regroupings
        if off:
            monkeypatch.setenv("DRMLT_NO_REGROUP", "1")
        else:
            monkeypatch.delenv("DRMLT_NO_REGROUP", raising=False)
        ctx = pkg.Context(_cfg(pkg, n, **kw), sd)
        ctx.seed(0xC0FFEE)
        ctx.run(n * per_chain)
        st = ctx.stats()
        res.append((ctx.chain_state(st.max_dim if kw["technique"] == "bdpt" else 27), st, ctx.film()))
        ctx.close()
    ((ca, ua), sa, fa), ((cb, ub), sb, fb) = res
    assert np.array_equal(ua, ub) and np.array_equal(ca["luminance"], cb["luminance"])
    assert sa.accepted == sb.accepted and sa.rays == sb.rays and sa.path_evals == sb.path_evals and sa.mutations == sb.mutations == n * per_chain
    assert (fa @ LUMW).sum() == pytest.approx((fb @ LUMW).sum(), rel=1e-5)


@pytest.mark.parametrize("kw,n", [(dict(technique="mmlt", max_depth=6), 5000), (dict(technique="bdpt", max_depth=6, no_direct_sampling=0), 3000),
                                  (dict(technique="mmlt", max_depth=11), 1030)], ids=["mmlt", "bdpt", "mmlt-deep"])
def test_the_device_regroups_as_the_host_sort_would(pkg, kw, n, native_lib, monkeypatch):
    """Round 4: the regrouping permutation is made on the device (histogram, scan, stable scatter: kernels_mmlt.hip) behind the launch
    whose counts it reads -- no host round trip between the launches of a call. DRMLT_REGROUP_CHECK compares every one of them with
    the host's stable sort of the same counts (the library fails the call on a difference); the host path (round 3's,
    DRMLT_REGROUP_ON_HOST) ends in the same states. Chain counts that are no multiple of the 1024-chain segments or of a wave."""
    sd = pkg.scenes.caustic_c5(32)
    per_chain = 80
    res = []
    for mode in ("device+check", "host"):
        monkeypatch.setenv("DRMLT_SLICE", "16")                      # five launches, a regrouping after each
        monkeypatch.delenv("DRMLT_REGROUP_CHECK", raising=False); monkeypatch.delenv("DRMLT_REGROUP_ON_HOST", raising=False)
        monkeypatch.setenv("DRMLT_REGROUP_CHECK" if mode == "device+check" else "DRMLT_REGROUP_ON_HOST", "1")
        ctx = pkg.Context(_cfg(pkg, n, **kw), sd)
        ctx.seed(0xFEED)
        ctx.run(n * per_chain)                                       # raises if a device permutation differs from the host's
        st = ctx.stats()
        res.append((ctx.chain_state(st.max_dim if kw["technique"] == "bdpt" else 2 * (kw["max_depth"] + 1) + 2 * kw["max_depth"] + 1), st))
        ctx.close()
    ((ca, ua), sa), ((cb, ub), sb) = res
    assert sa.launches == sb.launches == 5
    assert np.array_equal(ua, ub) and np.array_equal(ca["luminance"], cb["luminance"]) and sa.accepted == sb.accepted and sa.rays == sb.rays


def test_node_with_two_ranks_on_one_gpu_equals_single_context(pkg, native_lib, monkeypatch):
    """drmlt_node_* with two ranks (both on GPU 0: loopback transport for the reduce-scatter arithmetic): seed pool,
    threaded run, tiled develop, summed stats == one context with twice the chains."""
    sd = pkg.scenes.cornell_c2(64)
    n, total = 2048, 64 * 64 * 4
    monkeypatch.setenv("DRMLT_TEST_HOOKS", "1")
    monkeypatch.setenv("DRMLT_NODE_DEVICES", "0,0")
    node = pkg.Node(_cfg(pkg, n), sd, device_mask=1)
    monkeypatch.delenv("DRMLT_NODE_DEVICES")
    assert node.device_count == 2
    big = pkg.Context(_cfg(pkg, 2 * n), sd)
    bn, bb = node.seed(0x5EED), big.seed_pool(0x5EED, 0, 2 * n)
    assert bn == bb
    seen = []
    node.run(total, progress=lambda d, t: seen.append((d, t)))
    big.run(total)
    assert seen and seen[-1] == (total, total)
    img_n, img_b = node.develop(), big.develop()
    np.testing.assert_allclose(img_n, img_b, rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(node.film(0) + node.film(1), big.film(), rtol=2e-4, atol=1e-6)
    sn, sb = node.stats(), big.stats()
    assert sn.mutations == sb.mutations == total and sn.accepted == sb.accepted and sn.rays == sb.rays
    assert sn.n_chains == 2 * n and sn.first_base == sb.first_base and sn.second_acc == sb.second_acc
    direct = np.random.default_rng(1).random((64, 64, 3), dtype=np.float32)
    np.testing.assert_allclose(node.develop(direct), img_n + direct, rtol=1e-5, atol=1e-6)
    with pytest.raises(pkg.DrmltError, match="empty device mask"):
        pkg.Node(_cfg(pkg, n), sd, device_mask=0)
    node.close(); big.close()


def test_single_device_node_is_the_plain_context(pkg, native_lib):
    sd = pkg.scenes.cornell_c2(32)
    node, ctx = pkg.Node(_cfg(pkg, 1024), sd, device_mask=1), pkg.Context(_cfg(pkg, 1024), sd)
    assert node.seed(3) == ctx.seed_pool(3, 0, 1024)
    node.run(32 * 32 * 4); ctx.run(32 * 32 * 4)
    np.testing.assert_allclose(node.develop(), ctx.develop(), rtol=2e-4, atol=1e-6)


def test_rccl_exchange_from_cpp_world_size_one(pkg, native_lib):
    """ncclCommInitRank / ncclReduceScatter / ncclAllReduce issued by libdrmlt_amd.so itself on the context's stream."""
    sd = pkg.scenes.cornell_c2(50)          # 50 rows: not a multiple of anything convenient
    ctx = pkg.Context(_cfg(pkg, 2048), sd)
    b = ctx.seed(11)
    ctx.run(50 * 50 * 4)
    uid = pkg.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    with pytest.raises(pkg.DrmltError, match="drmlt_comm_init"):
        ctx.exchange_tiled(b)
    ctx.comm_init(uid, 0, 1)
    film_before = ctx.film()
    tile, rows, b_mean = ctx.exchange_tiled(b)
    assert rows == (0, 50) and b_mean == b
    np.testing.assert_allclose(tile, ctx.develop(), rtol=2e-5, atol=1e-7)
    np.testing.assert_array_equal(ctx.film(), film_before)          # the local film keeps accumulating untouched
    _, rows2, _ = ctx.exchange_tiled(b, want_tile=False)             # the timed form: no host copy
    assert rows2 == (0, 50)
    with pytest.raises(pkg.DrmltError, match="bad rank"):
        ctx.comm_init(uid, 2, 2)


def test_two_node_ranks_on_a_bvh_scene_run_the_ray_pool_kernel(pkg, native_lib, monkeypatch):
    """The same identity on a scene that is traversed (k_mutate_v5: 64 chains per wave, state in device memory, ragged last
    waves on both ranks): two loopback ranks of 1000 chains each == one context with the 2000 chains of the same pool."""
    sd = pkg.scenes.triangle_soup(2000, 48)
    n, total = 1000, 48 * 48 * 8
    monkeypatch.setenv("DRMLT_TEST_HOOKS", "1")
    monkeypatch.setenv("DRMLT_NODE_DEVICES", "0,0")
    node = pkg.Node(_cfg(pkg, n, sample_count=8), sd, device_mask=1)
    monkeypatch.delenv("DRMLT_NODE_DEVICES")
    big = pkg.Context(_cfg(pkg, 2 * n, sample_count=8), sd)
    assert node.seed(0x5EED) == big.seed_pool(0x5EED, 0, 2 * n)
    node.run(total); big.run(total)
    sn, sb = node.stats(), big.stats()
    assert sn.mutations == sb.mutations == total // (2 * n) * 2 * n and sn.accepted == sb.accepted and sn.rays == sb.rays
    assert sn.bvh_node_visits == sb.bvh_node_visits > 0
    np.testing.assert_allclose(node.film(0) + node.film(1), big.film(), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(node.develop(), big.develop(), rtol=5e-4, atol=1e-6)
    node.close(); big.close()
