"""BASELINE.json configs 3, 4 (one GPU's share) and 5 at their full sizes: size-independent properties only
(the oracle cannot follow at these sizes in seconds): unit luminance per mutation, counter identities, b = mean image
luminance after develop, finite non-negative film, states inside [0,1]."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LUMW = np.array([0.212671, 0.715160, 0.072169])


def lum(img):
    return img @ LUMW


def check_invariants(ctx, M, b, amap=False):
    st = ctx.stats()
    assert st.mutations == M
    assert st.first_base == M and st.large_base + st.bold_base == M
    assert st.second_base == st.bold_base - st.bold_acc
    assert st.overall_base == M + st.second_base and st.overall_acc == st.first_acc + st.second_acc == st.accepted
    assert abs(st.large_base / M - 0.3) < 3e-3
    film = ctx.film()
    assert np.all(np.isfinite(film)) and film.min() >= 0
    if not amap:
        assert lum(film.astype(np.float64)).sum() == pytest.approx(M * 0.99998 ** 2, rel=3e-3)
        img = ctx.develop()
        assert lum(img.astype(np.float64)).mean() == pytest.approx(b, rel=2e-3)
    return st, film


def test_config3_door_green_full_size(pkg, native_lib):
    """Veach-door-style scene, glossy floor, technique=path type=green (Green's reverse evaluations counted)."""
    sd = pkg.scenes.door_c3(512)
    n = 65536
    cfg = pkg.abi.make_config(technique="path", type="green", max_depth=8, rr_depth=5, direct_samples=-1, work_units=n,
                              luminance_samples=655360, sample_count=64)
    ctx = pkg.Context(cfg, sd)
    b = ctx.seed(0x5EED)
    M = 512 * 512 * 32
    ctx.run(M)
    st, _ = check_invariants(ctx, M, b)
    assert M + st.second_base <= st.path_evals <= M + 2 * st.second_base      # + one reverse evaluation per valid second stage
    cur, u = ctx.chain_state(34)
    assert np.all((u >= 0) & (u <= 1)) and np.all(cur["luminance"] > 0)


def test_config4_one_gpu_share_2048(pkg, native_lib):
    """Cornell 2048 x 2048, one GPU's share of the 8-GPU render: 65 536 chains, (2048^2 * 64) / 8 mutations."""
    sd = pkg.scenes.cornell_c2(2048)
    n = 65536
    cfg = pkg.abi.make_config(technique="path", type="orbital", max_depth=8, rr_depth=5, direct_samples=-1, work_units=n,
                              luminance_samples=655360, sample_count=64)
    ctx = pkg.Context(cfg, sd)
    b = ctx.seed(0x5EED, chain_offset=3 * n)                                   # rank 3 of 8: its own chain ids and bootstrap stream
    M = 2048 * 2048 * 64 // 8 // n * n
    ctx.run(M)
    check_invariants(ctx, M, b)
    assert ctx.film().shape == (2048, 2048, 3)


def test_config4_tiled_exchange_at_2048_with_eight_ranks_on_one_gpu(pkg, native_lib, monkeypatch):
    """Config 4's film (2048 x 2048) through the node API with EIGHT ranks -- all on this GPU (loopback transport for the
    reduce-scatter arithmetic): one seed pool split eight ways, threaded run, 256-row tiles developed per rank and
    stitched. The result equals one context running the same 8 x 8192 chains."""
    sd = pkg.scenes.cornell_c2(2048)
    n, ranks = 8192, 8
    mk = lambda w: pkg.abi.make_config(technique="path", type="orbital", max_depth=8, rr_depth=5, direct_samples=-1, work_units=w,
                                       luminance_samples=100000, sample_count=1)
    monkeypatch.setenv("DRMLT_TEST_HOOKS", "1")
    monkeypatch.setenv("DRMLT_NODE_DEVICES", ",".join(["0"] * ranks))
    node = pkg.Node(mk(n), sd, device_mask=1)
    monkeypatch.delenv("DRMLT_NODE_DEVICES")
    assert node.device_count == ranks
    big = pkg.Context(mk(ranks * n), sd)
    assert node.seed(0x5EED) == big.seed_pool(0x5EED, 0, ranks * n)
    total = ranks * n * 64
    node.run(total); big.run(total)
    img_n, img_b = node.develop(), big.develop()
    assert img_n.shape == (2048, 2048, 3)
    np.testing.assert_allclose(img_n, img_b, rtol=5e-4, atol=1e-6)
    sn, sb = node.stats(), big.stats()
    assert sn.mutations == sb.mutations == total and sn.accepted == sb.accepted and sn.rays == sb.rays and sn.n_chains == ranks * n
    node.close(); big.close()


def test_config5_caustic_mmlt_full_size(pkg, native_lib):
    """Glass caustic, mmlt / orbital / fixEmitterPath / acceptanceMap at 512 x 512 with 65 536 chains."""
    sd = pkg.scenes.caustic_c5(512)
    n = 65536
    cfg = pkg.abi.make_config(technique="mmlt", type="orbital", max_depth=6, direct_samples=-1, fix_emitter_path=1,
                              acceptance_map=1, work_units=n, sample_count=64, luminance_samples=100000)
    ctx = pkg.Context(cfg, sd)
    assert ctx.seed(0x5EED) == 1.0
    M = 512 * 512 * 64
    ctx.run(M)
    st, film = check_invariants(ctx, M, 1.0, amap=True)
    f = film.astype(np.float64)
    assert f[..., 2].max() == 0
    assert f[..., 0].sum() == pytest.approx(st.bold_acc, rel=1e-3) and f[..., 1].sum() == pytest.approx(st.second_acc, rel=1e-3)
    heat = pkg.heatmap.stage_ratio(film)
    assert 0.05 < heat[f[..., :2].sum(-1) > 20].mean() < 0.6
    cur, u = ctx.chain_state(27)
    assert set(np.unique(cur["n_dims"])) <= set(range(2, 7)) and np.all((u >= 0) & (u <= 1))
    # same configuration, radiance output: b is the mean image luminance
    cfg2 = pkg.abi.make_config(technique="mmlt", type="orbital", max_depth=6, direct_samples=-1, fix_emitter_path=1,
                               work_units=n, sample_count=16, luminance_samples=100000)
    ctx2 = pkg.Context(cfg2, sd)
    b = ctx2.seed(0x5EED)
    ctx2.run(512 * 512 * 16)
    check_invariants(ctx2, 512 * 512 * 16, b)


# ---- the same properties at the settings bench.py TIMES: its chain counts pick other kernels --
# k_mutate_v5 from 98 304 chains up (config 3: the conductor-only flat build <1>), the regrouped execution order of the
# bidirectional kernels (several launches per call), cuboid records in the ray loop.
def _bench_conf(name):
    import bench
    return bench.CONFIGS[name]


def _run_like_bench(pkg, name, spp, launches, capfd=None):
    conf = _bench_conf(name)
    sd = pkg.scenes.SCENES[conf["scene"][0]](res=conf["res"], **conf["scene"][1])
    n = conf.get("chains", 65536)
    cfg = pkg.abi.make_config(work_units=n, luminance_samples=100000, direct_samples=-1, sample_count=spp, **conf["cfg"])
    ctx = pkg.Context(cfg, sd)
    b = ctx.seed(0x5EED)
    M = conf["res"] * conf["res"] * spp
    assert M % n == 0
    ctx.run(M)
    return ctx, b, M, n


def test_bench_config_2x_full_size(pkg, native_lib):
    """`bench.py --config 2x`: Cornell 512^2, 196 608 chains -> k_mutate_v5's flat build with cuboid records, proposal rows in device
    memory, three waves per SIMD."""
    ctx, b, M, n = _run_like_bench(pkg, "2x", 480, None)
    st, _ = check_invariants(ctx, M, b)
    assert st.n_chains == 196608 and st.launches == 1 and st.bvh_node_visits == 0
    cur, u = ctx.chain_state(34)
    assert np.all((u >= 0) & (u <= 1)) and np.all(cur["luminance"] > 0)


def test_bench_config_3_full_size(pkg, native_lib):
    """`bench.py --config 3`: door scene, type=green, 196 608 chains -> k_mutate_v5<1> with three waves per SIMD (Green's reverse moves
    recomputed from the state in device memory, proposal rows in device memory too), two launches with run-ahead between them."""
    ctx, b, M, n = _run_like_bench(pkg, "3", 480, None)
    st, _ = check_invariants(ctx, M, b)
    assert st.n_chains == 196608 and st.launches == 1
    assert M + st.second_base <= st.path_evals <= M + 2 * st.second_base
    ctx.run(M)                                                                  # a second call continues every chain from its counter
    st2 = ctx.stats()
    assert st2.mutations == 2 * M and st2.launches == 2
    cur, u = ctx.chain_state(34)
    assert np.all((u >= 0) & (u <= 1)) and np.all(cur["luminance"] > 0)


def test_bench_config_5_full_size_regrouped(pkg, native_lib):
    """`bench.py --config 5`: caustic, mmlt / orbital / fixEmitterPath / acceptanceMap, 1 048 576 chains run in depth order and
    regrouped by work between the launches of a call (a short first launch, then the rest)."""
    ctx, _, M, n = _run_like_bench(pkg, "5", 1024, None)
    st, film = check_invariants(ctx, M, 1.0, amap=True)
    assert st.n_chains == 1048576 and st.launches >= 2                          # the first launch of a regrouping call is short
    f = film.astype(np.float64)
    assert f[..., 0].sum() == pytest.approx(st.bold_acc, rel=1e-3) and f[..., 1].sum() == pytest.approx(st.second_acc, rel=1e-3)
    cur, u = ctx.chain_state(27)
    assert set(np.unique(cur["n_dims"])) <= set(range(1, 7)) and np.all((u >= 0) & (u <= 1))


def test_bench_config_bdpt_full_size(pkg, native_lib):
    """`bench.py --config bdpt`: Cornell, bdpt / orbital, maxDepth 8, directSampling, 131 072 chains -> the two-waves-per-SIMD build."""
    ctx, b, M, n = _run_like_bench(pkg, "bdpt", 256, None)
    st = ctx.stats()
    assert st.mutations == M and st.n_chains == 131072 and st.first_base == M and st.large_base + st.bold_base == M
    assert st.second_base == st.bold_base - st.bold_acc and st.overall_acc == st.first_acc + st.second_acc == st.accepted
    film = ctx.film()
    assert np.all(np.isfinite(film)) and film.min() >= 0
    # a list's splats carry unit luminance in total: the film holds one unit per mutation (up to the splats that leave the film)
    assert lum(film.astype(np.float64)).sum() == pytest.approx(M, rel=2e-2)
    img = ctx.develop()
    assert lum(img.astype(np.float64)).mean() == pytest.approx(b, rel=2e-3)


def test_bench_config_soup50k_full_size_three_waves_per_simd(pkg, native_lib, capfd, monkeypatch):
    """`bench.py --config soup50k`: 50 000 triangles, 196 608 chains -> k_mutate_v5 with its proposal rows in device memory, built for
    three waves per SIMD (picked by the chain count, no environment variable)."""
    monkeypatch.setenv("DRMLT_VERBOSE", "1")
    ctx, b, M, n = _run_like_bench(pkg, "soup50k", 60, None)
    assert "proposal rows in device memory" in capfd.readouterr().err
    st, _ = check_invariants(ctx, M, b)
    assert st.n_chains == 196608 and st.bvh_node_visits > 50 * M
    cur, u = ctx.chain_state(34)
    assert np.all((u >= 0) & (u <= 1)) and np.all(cur["luminance"] > 0)
