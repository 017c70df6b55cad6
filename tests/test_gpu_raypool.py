"""k_mutate_v5, the ray-pool chain kernel of BVH scenes (64 chains per wave, rays queued in LDS, any lane traverses any
ray; north_star's "wavefront ballot / prefix-sum ray compaction"): it must run the chains of k_mutate_v4 / k_mutate_v3 bit
for bit -- same addressed draws, same proposal arithmetic, same acceptance code (device_mh.h) -- whatever the order in
which the wave happens to traverse its rays. (The kernel keeps ONE proposal row group in LDS and recomputes, from the state
in device memory and the addressed stream, what Green's reverse move and Mira's transition ratio need beside it.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LUMW = np.array([0.212671, 0.715160, 0.072169])


def lum(img):
    return img @ LUMW


def ctx_with_env(pkg, cfg, sd, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return pkg.Context(cfg, sd)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


CASES = [
    ("triangle_soup", dict(n_tris=2000), dict(type="orbital"), {}),                                  # 16-bit stacks, diffuse-only build
    ("triangle_soup", dict(n_tris=2000), dict(type="orbital", timid_after_large=1), {}),             # second stages after large steps: uniform rows
    ("triangle_soup", dict(n_tris=2000), dict(type="orbital", use_mixture=1), {}),
    ("triangle_soup", dict(n_tris=2000), dict(type="orbital", p_large=0.02, max_depth=5, rr_depth=2), {}),   # another state size (D = 18)
    ("triangle_soup", dict(n_tris=40000), dict(type="orbital"), {}),                                 # 32-bit stacks, overflow area in use
    ("caustic_c5", {}, dict(type="orbital"), dict(DRMLT_BVH_THRESHOLD=0)),                           # spheres + dielectric in the leaves
    ("door_c3", {}, dict(type="orbital"), dict(DRMLT_BVH_THRESHOLD=0)),                              # rough conductor
    # flat scenes: the same kernel with the brute-force loop as its trace phase, scene tables in LDS
    ("cornell_c2", {}, dict(type="orbital"), {}),
    ("cornell_c2", {}, dict(type="orbital", use_mixture=1, timid_after_large=0), {}),
    ("caustic_c5", {}, dict(type="orbital", timid_after_large=1), {}),
    ("door_c3", {}, dict(type="orbital"), {}),
    # Green & Mira's rule: second stage x + g over the rows, reverse move recomputed from the stream, adoption of z recomputed
    ("triangle_soup", dict(n_tris=2000), dict(type="green"), {}),
    ("triangle_soup", dict(n_tris=2000), dict(type="green", timid_after_large=1, p_large=0.5), {}),
    ("triangle_soup", dict(n_tris=2000), dict(type="green", use_mixture=1), {}),
    ("door_c3", {}, dict(type="green"), {}),                                                          # BASELINE config 3's kernel
    ("cornell_c2", {}, dict(type="green", timid_after_large=1, max_depth=5, rr_depth=2), {}),
    # Tierney & Mira's rule: the ratio Q1(y|z) / Q1(y|x) from the rows (z), the state (x) and the first-stage draws again (y)
    ("triangle_soup", dict(n_tris=2000), dict(type="mira"), {}),
    ("triangle_soup", dict(n_tris=2000), dict(type="mira", timid_after_large=1, p_large=0.5), {}),
    ("door_c3", {}, dict(type="mira"), {}),
    ("cornell_c2", {}, dict(type="mira", max_depth=5, rr_depth=2), {}),
]


@pytest.mark.parametrize("scene,skw,kw,env", CASES, ids=["soup-orbital", "soup-timid", "soup-mixture", "soup-short", "soup40k", "caustic-bvh", "door-bvh", "flat-cornell", "flat-cornell-mixture", "flat-caustic-timid", "flat-door",
                              "soup-green", "soup-green-timid", "soup-green-mixture", "flat-door-green", "flat-cornell-green-timid",
                              "soup-mira", "soup-mira-timid", "flat-door-mira", "flat-cornell-mira-short"])
def test_ray_pool_kernel_runs_the_same_chains(pkg, native_lib, scene, skw, kw, env):
    sd = pkg.scenes.SCENES[scene](res=32, **skw)
    n_chains, n_mut = 1000, 60                       # 1000: the last wave of either kernel is ragged
    base = dict(max_depth=8, direct_samples=-1, luminance_samples=20000, work_units=n_chains, sample_count=1)
    base.update(kw)
    cfg = pkg.abi.make_config(**base)
    res = []
    for kern in (4, 5):
        ctx = ctx_with_env(pkg, cfg, sd, DRMLT_KERNEL=kern, **env)
        ctx.seed(0x5005)
        ctx.run(n_chains * n_mut)
        res.append((ctx.chain_state(34), ctx.stats(), ctx.film()))
        ctx.close()
    ((c4, u4), s4, f4), ((c5, u5), s5, f5) = res
    assert s5.mutations == s4.mutations == n_chains * n_mut
    assert np.array_equal(u5, u4)                                                    # states: bit-equal
    assert np.array_equal(c5["luminance"], c4["luminance"])                          # f(u): two compilations of the path step, one rounding (-ffp-contract=on)
    for k in ("first", "large", "bold", "second", "second_large", "second_bold", "overall"):
        assert getattr(s5, k + "_base") == getattr(s4, k + "_base") and getattr(s5, k + "_acc") == getattr(s4, k + "_acc"), k
    assert s5.accepted == s4.accepted and s5.rays == s4.rays and s5.path_evals == s4.path_evals
    assert s5.bvh_node_visits == s4.bvh_node_visits and s5.bvh_prim_tests == s4.bvh_prim_tests   # the same rays, traversed the same way
    assert (s5.bvh_node_visits > 0) == (scene == "triangle_soup" or "DRMLT_BVH_THRESHOLD" in env)
    assert lum(f5).sum() == pytest.approx(lum(f4).sum(), rel=1e-5)
    assert np.abs(lum(f5) - lum(f4)).sum() / lum(f4).sum() < 1e-4


CASE_IDS = ["soup-orbital", "soup-timid", "soup-mixture", "soup-short", "soup40k", "caustic-bvh", "door-bvh", "flat-cornell", "flat-cornell-mixture", "flat-caustic-timid", "flat-door",
            "soup-green", "soup-green-timid", "soup-green-mixture", "flat-door-green", "flat-cornell-green-timid",
            "soup-mira", "soup-mira-timid", "flat-door-mira", "flat-cornell-mira-short"]


@pytest.mark.parametrize("scene,skw,kw,env", CASES, ids=CASE_IDS)
def test_rows_in_device_memory_run_the_same_chains(pkg, native_lib, scene, skw, kw, env):
    """From 163 840 chains per GPU up, k_mutate_v5 keeps its proposal rows in device memory and is built for three waves per SIMD
    (ROWS_MEM; traversed scenes and flat ones alike). Forced here at 1000 chains: the chains of the rows-in-LDS build, bit for bit."""
    sd = pkg.scenes.SCENES[scene](res=32, **skw)
    n_chains, n_mut = 1000, 60
    base = dict(max_depth=8, direct_samples=-1, luminance_samples=20000, work_units=n_chains, sample_count=1)
    base.update(kw)
    cfg = pkg.abi.make_config(**base)
    res = []
    for rows_mem in (0, 1):
        ctx = ctx_with_env(pkg, cfg, sd, DRMLT_KERNEL=5, DRMLT_ROWS_MEM=rows_mem, **env)
        ctx.seed(0x5005)
        ctx.run(n_chains * n_mut)
        res.append((ctx.chain_state(34), ctx.stats(), ctx.film()))
        ctx.close()
    ((c0, u0), s0, f0), ((c1, u1), s1, f1) = res
    assert s1.mutations == s0.mutations == n_chains * n_mut
    assert np.array_equal(u1, u0) and np.array_equal(c1["luminance"], c0["luminance"])
    for k in ("first", "large", "bold", "second", "second_large", "second_bold", "overall"):
        assert getattr(s1, k + "_base") == getattr(s0, k + "_base") and getattr(s1, k + "_acc") == getattr(s0, k + "_acc"), k
    assert s1.accepted == s0.accepted and s1.rays == s0.rays and s1.path_evals == s0.path_evals and s1.bvh_node_visits == s0.bvh_node_visits
    assert lum(f1).sum() == pytest.approx(lum(f0).sum(), rel=1e-5)
    assert np.abs(lum(f1) - lum(f0)).sum() / lum(f0).sum() < 1e-4


@pytest.mark.parametrize("rows_mem", [0, 1], ids=["rows-lds", "rows-mem"])
@pytest.mark.parametrize("scene,skw,env", [("triangle_soup", dict(n_tris=2000), {}), ("caustic_c5", {}, dict(DRMLT_BVH_THRESHOLD=0))], ids=["soup", "caustic-bvh"])
def test_small_tables_in_lds_change_nothing(pkg, native_lib, scene, skw, env, rows_mem):
    """On traversed scenes the BSDF / emitter records and the emitters' shape records are staged in LDS when they fit beside the ray pool
    (HybridTables); scenes whose tables do not fit read them from device memory. Both paths, forced: the same chains bit for bit."""
    sd = pkg.scenes.SCENES[scene](res=32, **skw)
    n_chains, n_mut = 1000, 60
    cfg = pkg.abi.make_config(type="orbital", max_depth=8, direct_samples=-1, luminance_samples=20000, work_units=n_chains, sample_count=1)
    res = []
    for no_small in (None, 1):
        extra = dict(env)
        if no_small:
            extra["DRMLT_NO_SMALL_TABLES"] = 1
        ctx = ctx_with_env(pkg, cfg, sd, DRMLT_KERNEL=5, DRMLT_ROWS_MEM=rows_mem, **extra)
        ctx.seed(0x5005)
        ctx.run(n_chains * n_mut)
        res.append((ctx.chain_state(34), ctx.stats(), ctx.film()))
        ctx.close()
    ((c0, u0), s0, f0), ((c1, u1), s1, f1) = res
    assert np.array_equal(u1, u0) and np.array_equal(c1["luminance"], c0["luminance"])
    assert s1.accepted == s0.accepted and s1.rays == s0.rays and s1.bvh_node_visits == s0.bvh_node_visits > 0
    assert np.abs(lum(f1) - lum(f0)).sum() / lum(f0).sum() < 1e-4


def test_ray_pool_kernel_with_run_ahead_and_acceptance_map(pkg, native_lib):
    sd = pkg.scenes.triangle_soup(2000, 32)
    n_chains, per_chain = 1536, 300
    cfg = pkg.abi.make_config(type="orbital", max_depth=8, direct_samples=-1, luminance_samples=20000, work_units=n_chains, sample_count=1,
                              acceptance_map=1)
    res = []
    for env in (dict(DRMLT_KERNEL=4, DRMLT_SLICE=64), dict(DRMLT_KERNEL=5, DRMLT_SLICE=64), dict(DRMLT_KERNEL=5, DRMLT_SLICE=64, DRMLT_NO_RUN_AHEAD=1)):
        ctx = ctx_with_env(pkg, cfg, sd, **env)
        ctx.seed(9)
        os.environ.update({k: str(v) for k, v in env.items()})
        try:
            ctx.run(n_chains * per_chain)
        finally:
            for k in env:
                del os.environ[k]
        res.append((ctx.chain_state(34)[1], ctx.stats(), ctx.film()))
        ctx.close()
    for u, s, f in res[1:]:
        assert np.array_equal(u, res[0][0]) and s.accepted == res[0][1].accepted and s.mutations == n_chains * per_chain
        np.testing.assert_array_equal(f, res[0][2])                                   # marks are whole numbers of box weights: exact


@pytest.mark.parametrize("scene,skw,filt", [("triangle_soup", dict(n_tris=2000), "box"), ("cornell_c2", {}, "gauss")], ids=["soup-importance", "cornell-gaussian-importance"])
def test_ray_pool_kernel_with_importance_map_and_gaussian_film(pkg, native_lib, scene, skw, filt):
    """Two-stage MLT's luminance image (SplatList::normalize divides every splat by it, pathsampler.cpp:1001-1020) and the gaussian
    film filter go through the same shared code in both kernels: same chains, same film."""
    abi = pkg.abi
    kw = dict(skw)
    if filt == "gauss":
        kw["filt"] = abi.FILTER_GAUSSIAN
    sd = pkg.scenes.SCENES[scene](res=32, **kw)
    rng = np.random.default_rng(3)
    imp = (0.2 + rng.random((32, 32))).astype(np.float32)
    imp[:4, :4] = 0.0                                              # unlit region: proposals into it are rejected (luminance inf)
    n_chains, n_mut = 1536, 50
    cfg = abi.make_config(type="orbital", max_depth=8, direct_samples=-1, luminance_samples=20000, work_units=n_chains, sample_count=1)
    res = []
    for kern in (4, 5):
        ctx = ctx_with_env(pkg, cfg, sd, DRMLT_KERNEL=kern)
        ctx.set_importance_map(imp)
        ctx.seed(77)
        ctx.run(n_chains * n_mut)
        res.append((ctx.chain_state(34)[1], ctx.stats(), ctx.film(), ctx.develop()))
        ctx.close()
    (u4, s4, f4, i4), (u5, s5, f5, i5) = res
    assert np.array_equal(u5, u4) and s5.accepted == s4.accepted and s5.rays == s4.rays
    assert np.abs(lum(f5) - lum(f4)).sum() / lum(f4).sum() < 1e-4
    np.testing.assert_allclose(i5, i4, rtol=2e-3, atol=1e-6)


def test_default_kernel_choice(pkg, native_lib, capfd):
    """Which chain kernel runs where: BVH scenes -> the ray pool (any chain count); flat scenes -> the ray pool from 98 304
    chains up (workUnits = -1 derives 196 608), the lane-pair kernel below (BASELINE config 2's 65 536 chains)."""
    def kernel_of(sd, **kw):
        cfg = pkg.abi.make_config(max_depth=8, direct_samples=-1, luminance_samples=1000, **kw)
        os.environ["DRMLT_VERBOSE"] = "1"
        try:
            ctx = pkg.Context(cfg, sd)
            ctx.seed(1)
            capfd.readouterr()
            ctx.run(ctx.stats().n_chains * 2)
            err = capfd.readouterr().err
            n = ctx.stats().n_chains
            ctx.close()
        finally:
            del os.environ["DRMLT_VERBOSE"]
        return ("v5" if "k_mutate_v5" in err else "v4" if "k_mutate_v4" in err else "?"), n
    soup, cornell = pkg.scenes.triangle_soup(2000, 64), pkg.scenes.cornell_c2(512)
    assert kernel_of(soup, type="orbital", work_units=1024, sample_count=1) == ("v5", 1024)
    assert kernel_of(soup, type="mira", work_units=1024, sample_count=1) == ("v5", 1024)
    assert kernel_of(cornell, type="orbital", work_units=65536, sample_count=256) == ("v4", 65536)      # BASELINE configs[1]
    assert kernel_of(cornell, type="orbital", work_units=-1, sample_count=256) == ("v5", 196608)
    assert kernel_of(cornell, type="green", work_units=131072, sample_count=256) == ("v5", 131072)
    # traversed scenes: three waves per SIMD as well (workUnits = -1 derives the count that fills them)
    assert kernel_of(pkg.scenes.triangle_soup(2000, 512), type="orbital", work_units=-1, sample_count=256) == ("v5", 196608)
    assert kernel_of(pkg.scenes.triangle_soup(40000, 512), type="orbital", work_units=-1, sample_count=256) == ("v5", 196608)
