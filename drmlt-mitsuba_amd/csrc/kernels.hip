// HIP kernels of the DRMLT hot path for gfx950 (MI355X). One Markov chain per lane.
//
//   k_bootstrap     luminance samples of PathSampler::generateSeeds (pathsampler.cpp:879-920)
//   k_init_chains   seed replay + fillReplay + luminance sanity check (drmlt_proc.cpp:467-514)
//   k_mutate_v4     DRMLTRenderer::process / processMixture chain loop (drmlt_proc.cpp:161-380,518-770); k_mutate_v3 = its
//                   predecessor, kept as the bit-equality cross-check; k_mutate_pssmlt = PSSMLTRenderer::process
//   k_eval_paths    PathSampler::sampleSplats on caller-supplied PSS points (pathsampler.cpp:529-567)
//   k_render_pt     independent samples of the same integrand (validation image)
//   k_lum_sum / k_develop   DRMLTProcess::develop (drmlt_proc.cpp:824-849)
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <type_traits>
#include "device_path.h"

#include "kernel_common.h"

// Bootstrap sample `index` of the replayable stream. ONE compiled body serves k_bootstrap and k_init_chains: the replay is
// checked for EQUALITY with the bootstrap luminance (drmlt_proc.cpp:509-512), and two inlined copies of the same source
// may be contracted / scheduled differently by the compiler -- a rounding difference that flips one discrete decision in
// one of 1e5 paths fails the seeding.
__device__ __attribute__((noinline)) DSplat eval_boot_sample(const DParams &P, uint32_t index, uint32_t &nd) {
    Sampler smp;
    smp.key0 = P.key0; smp.key1 = P.key1; smp.chain = P.boot_stream; smp.major = index;
    smp.mode = SM_BOOT; smp.type = P.type; smp.large = false; smp.sigma2 = P.sigma2; smp.lane = 0u; smp.arr = nullptr;
    uint32_t nr;
    return eval_path(P, smp, nr, nd);
}

__global__ void __launch_bounds__(64) k_bootstrap(DParams P, uint32_t n, float *lum_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t nd;
    DSplat s = eval_boot_sample(P, i, nd);
    lum_out[i] = s.lum;
    if (P.boot_weighted) { normalize_splat(s, P); lum_out[(size_t) n + i] = s.lum; } // luminance of f / importance
}

__global__ void __launch_bounds__(64) k_init_chains(DParams P, const uint32_t *seed_index, const float *seed_lum) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.n_chains) return;
    Sampler smp;
    smp.key0 = P.key0; smp.key1 = P.key1; smp.chain = P.boot_stream; smp.major = seed_index[c];
    smp.mode = SM_BOOT; smp.type = P.type; smp.large = false; smp.sigma2 = P.sigma2; smp.lane = 0u; smp.arr = nullptr;
    uint32_t nd;
    DSplat s = eval_boot_sample(P, smp.major, nd);
    // sanity check of drmlt_proc.cpp:509-512: same function, same inputs -> bit-equal on the device
    if (!(s.lum == seed_lum[c])) atomicExch(P.error_flag, 1);
    normalize_splat(s, P);
    P.cur_lum[c] = s.lum; P.cur_px[c] = s.px; P.cur_py[c] = s.py;
    P.cur_r[c] = s.r; P.cur_g[c] = s.g; P.cur_b[c] = s.b;
    P.chain_depth[c] = (int32_t) nd; // pssmlt: components that exist after the replay (no fillReplay there)
    // replayed components + fillReplay top-up: dimension k of bootstrap sample i is U(BOOT, i, k)
    smp.reset_caches();
    for (uint32_t k = 0; k < (uint32_t) P.eff_dim; ++k) P.x[(size_t) k * P.n_chains + c] = smp.u_boot(k, TAG_BOOT);
}

// ------------------------------------------------------------------------------------------
// k_mutate_pssmlt: PSSMLTRenderer::process (src/integrators/pssmlt/pssmlt_proc.cpp:113-297) over sampleSplats(path).
// One proposal per mutation, Kelemen-style weights (with b and pLarge) or Veach's expectations, and the deferred splat
// of the current state with its cumulative weight (:215-226,262-266). The cumulative weight is flushed at the end of
// every launch (the reference does it once per work unit; splatting is linear, so the film is the same).
__global__ void __launch_bounds__(CHAIN_BLOCK) k_mutate_pssmlt(DParams P, uint32_t n_mut, uint32_t mut_base) {
    const uint32_t lane = threadIdx.x;
    const uint32_t c = blockIdx.x * CHAIN_BLOCK + lane;
    const bool live = c < P.n_chains;
    const uint32_t cc = live ? c : P.n_chains - 1;
    const int D = P.eff_dim;
    for (int k = 0; k < D; ++k) lds_x[k * 64 + lane] = P.x[(size_t) k * P.n_chains + cc];
    DSplat cur;
    cur.lum = P.cur_lum[cc]; cur.px = P.cur_px[cc]; cur.py = P.cur_py[cc];
    cur.r = P.cur_r[cc]; cur.g = P.cur_g[cc]; cur.b = P.cur_b[cc];
    PssmltSampler smp;
    smp.key0 = P.key0; smp.key1 = P.key1; smp.chain = P.chain_offset + cc;
    smp.kelemen = P.kelemen_mutation != 0; smp.sigma = P.pss_sigma; smp.lane = lane;
    smp.n_exist = mut_base == 0u ? (uint32_t) min(P.chain_depth[cc], D) : (uint32_t) D;
    Counters ct = {0u, 0u, 0u, 0u, 0u};
    float cumulative = 0.f;
    const float b = P.luminance_b, pLarge = P.p_large;

    if (live) for (uint32_t it = 0; it < n_mut; ++it) {
        const uint32_t m = mut_base + it;
        const u4 coins = philox4x32_10(P.key0, P.key1, 0u, m, smp.chain, TAG_COIN);
        const bool large = u32_to_unit(coins.x) < pLarge;
        smp.major = m; smp.large = large;
        smp.reset_caches();
        uint32_t nr, nd;
        DSplat y = eval_path(P, smp, nr, nd);
        ct.rays += nr;
        normalize_splat(y, P);
        float a = fminf(1.f, y.lum / cur.lum);
        if (isnan(y.lum) || y.lum < 0.f) a = 0.f; // :188-191
        bool accept = false;
        float wc, wp = 0.f;
        if (a > 0.f) {
            if (P.kelemen_weights && !P.importance) { // :197-203: "Kelemen-style weights don't work for 2-stage MLT" (the a <= 0 branch keeps them)
                wc = (1.f - a) * cur.lum / (cur.lum / b + pLarge);
                wp = (a + (large ? 1.f : 0.f)) * y.lum / (y.lum / b + pLarge);
            } else {
                wc = 1.f - a;
                wp = a;
            }
            accept = a == 1.f || u32_to_unit(coins.y) < a;
        } else {
            wc = P.kelemen_weights ? cur.lum / (cur.lum / b + pLarge) : 1.f;
        }
        cumulative += wc;
        mh_count(ct, large, accept, false, false);
        // the whole vector was rewritten by the proposal; components that did not exist yet stay even on rejection
        const uint32_t n_exist = smp.n_exist;
        if (accept) {
            film_put(P, cur.px, cur.py, mk3(cur.r * cumulative, cur.g * cumulative, cur.b * cumulative));
            cumulative = wp;
            for (int k = 0; k < D; ++k) lds_x[k * 64 + lane] = smp.next((uint32_t) k);
            cur = y;
        } else {
            film_put(P, y.px, y.py, mk3(y.r * wp, y.g * wp, y.b * wp));
            for (uint32_t k = n_exist; k < (uint32_t) D; ++k) lds_x[k * 64u + lane] = smp.next(k);
        }
        smp.n_exist = (uint32_t) D;
    }
    if (live) {
        film_put(P, cur.px, cur.py, mk3(cur.r * cumulative, cur.g * cumulative, cur.b * cumulative)); // "Perform the last splat"
        for (int k = 0; k < D; ++k) P.x[(size_t) k * P.n_chains + c] = lds_x[k * 64 + lane];
        P.cur_lum[c] = cur.lum; P.cur_px[c] = cur.px; P.cur_py[c] = cur.py;
        P.cur_r[c] = cur.r; P.cur_g[c] = cur.g; P.cur_b[c] = cur.b;
    }
    flush_counters(P, ct, lane);
}

// ------------------------------------------------------------------------------------------
struct ChainState {
    DSplat cur, y, z;
    float a1, coin_acc1, coin_acc2, coin_mix;
    uint32_t it, nd1, nd2;
    int stage;       // -1: no mutation in flight, 0/1/2: evaluating first / second / reverse
    bool large, do_second;
};

// The bookkeeping branch in four pieces so that k_mutate_v3 can share the two heavy ones
// (committing D_eff dimensions, drawing the next mutation's uniforms) between a chain lane and
// its helper lane:
//   mh_decide  digest the finished evaluation; if the mutation is decided: weights + counters,
//              returns the commit mode (0 none, SM_STAGE1 = adopt y, SM_STAGE2 = adopt z)
//   commit     x[k] = wrap(proposal[k]) for a range of dimensions           (shareable)
//   mh_start   advance to the next evaluation (next stage or next mutation); returns which
//              uniforms must be drawn (0 none, 1 first stage, 2 second stage)
//   fill       Philox draws into LDS for a range of blocks                    (shareable)
// Outcome of one digested evaluation. `decided`: the mutation is over and w = the expectation weights of the current
// state, the first-stage and the second-stage proposal; `commit`: 0 none, SM_STAGE1 = adopt y, SM_STAGE2 = adopt z;
// `amap`: acceptance-map mark (AMAP_*) for the pixel of the state that is being REPLACED (device_mh.h: mh_amap_mark).
// The caller splats: k_mutate_v3 at once, k_mutate_v4 through its LDS queue.
struct MhOutcome {
    bool decided;
    int commit, amap;
    MhWeights w;
};

// Tierney & Mira's transition ratio Q1(y|z) / Q1(y|x) over the dimensions either stage used (drmlt_sampler.cpp:400-414)
template <class SamplerT> DEV float mira_ratio(SamplerT &smp, uint32_t nd1, uint32_t nd2) {
    const uint32_t dimStage = max(nd1, nd2) - 1u;
    float num = 0.f, den = 0.f;
    for (uint32_t i = 0; i < dimStage; ++i) {
        const float yi = smp.y_raw(i);
        num += kelemen_logpdf(smp.z_raw(i) - yi);
        den += kelemen_logpdf(smp.x(i) - yi);
    }
    return __expf(num - den);
}

template <class SamplerT> DEV MhOutcome mh_decide(const DParams &P, ChainState &cs, SamplerT &smp, PathState &ps, Counters &ct) {
    const bool mix = P.use_mixture != 0;
    MhOutcome out{false, 0, AMAP_NONE, {0.f, 0.f, 0.f}};
    if (cs.stage < 0) return out; // nothing evaluated yet (first call of a launch)
    float a2 = 0.f;
    bool acc1 = false, acc2 = false;
    DSplat res;
    res.px = ps.px; res.py = ps.py; res.r = ps.Li.x; res.g = ps.Li.y; res.b = ps.Li.z;
    res.lum = luminance3(ps.Li);
    normalize_splat(res, P);
    ct.rays += ps.nrays;
    if (cs.stage == 0) {
        cs.y = res; cs.nd1 = ps.k;
        mh_first(mix, P.timid_after_large != 0, cs.large, res.lum, cs.cur.lum, cs.coin_acc1, cs.coin_mix, cs.a1, acc1, cs.do_second);
        if (cs.do_second) { cs.stage = 1; return out; }
    } else if (cs.stage == 1) {
        cs.z = res; cs.nd2 = ps.k;
        if (mix) {
            cs.a1 = 0.f; // the second proposal replaces the first
            mh_second_mixture(res.lum, cs.cur.lum, cs.coin_acc2, a2, acc2);
        } else if (!lum_invalid(res.lum)) {
            if (P.type == 0) { cs.stage = 2; return out; } // Green: evaluate the reverse move first
            if (P.type == 1) {
                float ratio = 1.f;
                if (!cs.large && !(fminf(1.f, cs.y.lum / res.lum) >= 1.f)) ratio = mira_ratio(smp, cs.nd1, cs.nd2);
                mh_second_mira(cs.y.lum, res.lum, cs.cur.lum, cs.a1, ratio, cs.coin_acc2, a2, acc2);
            } else {
                mh_second_orbital(cs.y.lum, res.lum, cs.cur.lum, cs.coin_acc2, a2, acc2);
            }
        }
    } else {
        ct.acc2b_rev += 1u << 16;
        mh_second_green(res.lum, cs.z.lum, cs.cur.lum, cs.a1, cs.coin_acc2, a2, acc2);
    }
    out.decided = true;
    out.w = mh_weights(mix, P.acceptance_map != 0, cs.do_second, cs.a1, a2);
    mh_count(ct, cs.large, acc1, acc2, cs.do_second);
    if (acc1 || acc2) out.commit = acc1 ? SM_STAGE1 : SM_STAGE2;
    out.amap = mh_amap_mark(mix, P.acceptance_map != 0, cs.large, acc1, acc2);
    cs.it++;
    cs.stage = -1;
    return out;
}

// k_mutate_v3: splat the decided mutation at once and adopt the accepted proposal. The three splats go through ONE
// film_put site (the call expands to ~150 instructions; six inlined copies of it were a sixth of the kernel's code):
// slot 0 current state, 1 first-stage, 2 second-stage proposal.
DEV int mh_decide_splat(const DParams &P, ChainState &cs, LdsSampler &smp, PathState &ps, Counters &ct) {
    const MhOutcome o = mh_decide(P, cs, smp, ps, ct);
    if (!o.decided) return 0;
#pragma nounroll
    for (int i = 0; i < 3; ++i) {
        const float w = i == 0 ? o.w.w0 : (i == 1 ? o.w.w1 : o.w.w2);
        const DSplat sp = select_splat(i == 0, cs.cur, select_splat(i == 1, cs.y, cs.z));
        if (w > 0.f) film_put(P, sp.px, sp.py, mk3(sp.r * w, sp.g * w, sp.b * w));
    }
    if (o.commit) {
        if (o.amap) film_put(P, cs.cur.px, cs.cur.py, mh_amap_colour(o.amap)); // the state that is LEFT (device_mh.h)
        cs.cur = select_splat(o.commit == SM_STAGE1, cs.y, cs.z);
    }
    return o.commit;
}

// DRMLTSampler::accept for dimensions [k0, k1): uCurrent = wrap(chosen proposal)
DEV void commit_range(LdsSampler &smp, int commit_mode, uint32_t k0, uint32_t k1) {
    smp.mode = commit_mode;
    if (smp.type == 2 && !(k0 & 1u) && !(k1 & 1u)) { // orbital: pair by pair (k0, k1 are pair-aligned)
        const bool second = commit_mode == SM_STAGE2;
        for (uint32_t k = k0; k < k1; k += 2u) {
            float v0, v1;
            smp.orbital_pair(k, second, v0, v1);
            // a large step's proposal is the uniforms themselves (second stage after a large step: the s2 rows): selects,
            // so that lanes committing a large step do not drag the wave through the per-component loop
            const float l0 = second ? smp.s2(k) : smp.u1(k), l1 = second ? smp.s2(k + 1u) : smp.u1(k + 1u);
            lds_x[k * smp.stride + smp.lane] = wrap01(smp.large ? l0 : v0);
            lds_x[(k + 1u) * smp.stride + smp.lane] = wrap01(smp.large ? l1 : v1);
        }
        return;
    }
    for (uint32_t k = k0; k < k1; ++k) lds_x[k * smp.stride + smp.lane] = smp.next(k);
}

DEV int mh_start(const DParams &P, ChainState &cs, LdsSampler &smp, PathState &ps, uint32_t n_mut, uint32_t mut_base) {
    int fill = 0;
    if (cs.stage < 0) { // next mutation
        if (cs.it >= n_mut) { ps.phase = PH_IDLE; return 0; }
        const uint32_t m = mut_base + cs.it;
        const u4 coins = philox4x32_10(P.key0, P.key1, 0u, m, smp.chain, TAG_COIN);
        cs.large = u32_to_unit(coins.x) < P.p_large;
        cs.coin_acc1 = u32_to_unit(coins.y); cs.coin_acc2 = u32_to_unit(coins.z); cs.coin_mix = u32_to_unit(coins.w);
        smp.major = m;
        smp.large = cs.large;
        cs.stage = 0;
        cs.do_second = false;
        cs.nd1 = cs.nd2 = 0u;
        fill = 1;
    } else if (cs.stage == 1) {
        fill = 2;
    }
    smp.mode = cs.stage == 0 ? SM_STAGE1 : (cs.stage == 1 ? SM_STAGE2 : SM_REVERSE);
    path_init(P, ps);
    return fill;
}

// ------------------------------------------------------------------------------------------
// k_mutate_v3: two lanes per chain. 64 k chains are only 1024 waves = ONE wave per SIMD: every
// LDS / scalar-cache / transcendental latency is exposed and a lone wave can issue a VALU op only
// every 4 cycles (MI355X_MICROARCH.md). Lanes 0..31 of a wave run the chain state machines of
// k_mutate_v2; lane 32+i is the helper of lane i and traces the shadow (NEE) ray of a vertex
// while lane i traces the BSDF-sampled ray of the same vertex. A bounce then costs one loop
// iteration instead of two, a wave carries 32 chains, and the same 64 k chains occupy 2048
// waves = two per SIMD, which hide each other's latencies. Per chain the arithmetic and the
// order of all additions are those of k_mutate_v2.
DEV float from_lower(float v) { // value of lane (l & 31) for every lane l (v_permlane32_swap)
    unsigned u = __float_as_uint(v);
    return __uint_as_float(__builtin_amdgcn_permlane32_swap(u, u, false, false)[0]);
}
DEV unsigned from_lower_u(unsigned u) { return __builtin_amdgcn_permlane32_swap(u, u, false, false)[0]; }
DEV unsigned from_upper_u(unsigned u) { // value of lane 32 + (l & 31) for every lane l
    return __builtin_amdgcn_permlane32_swap(u, u, false, false)[1];
}

// LDS_TABLES: scene tables staged in LDS (small scenes) or read from HBM/L2 -- a launch-time property, compiled in so
// that the kernel carries ONE copy of the path step (the two-way runtime branch doubled the hot loop's code and pushed
// it past the 64 KB instruction cache).
template <int FEAT, bool LDS_TABLES>
__global__ void __launch_bounds__(CHAIN_BLOCK) k_mutate_v3(DParams P, uint32_t n_mut, uint32_t mut_base) {
    const uint32_t lane = threadIdx.x;
    const uint32_t sub = lane & 31u;
    const bool helper = lane >= 32u;
    const uint32_t c = blockIdx.x * 32u + sub;
    const bool live = !helper && c < P.n_chains;
    const uint32_t cc = c < P.n_chains ? c : P.n_chains - 1;
    const int D = P.eff_dim;
    const uint32_t D4 = ((uint32_t) D + 3u) & ~3u;
    if (!helper)
        for (int k = 0; k < D; ++k) lds_x[(uint32_t) k * 32u + sub] = P.x[(size_t) k * P.n_chains + cc];

    ChainState cs;
    cs.cur.lum = P.cur_lum[cc]; cs.cur.px = P.cur_px[cc]; cs.cur.py = P.cur_py[cc];
    cs.cur.r = P.cur_r[cc]; cs.cur.g = P.cur_g[cc]; cs.cur.b = P.cur_b[cc];
    cs.y = cs.cur; cs.z = cs.cur;
    cs.a1 = 0.f; cs.coin_acc1 = cs.coin_acc2 = cs.coin_mix = 0.f;
    cs.it = 0u; cs.nd1 = cs.nd2 = 0u; cs.stage = -1; cs.large = false; cs.do_second = false;

    LdsSampler smp;
    smp.key0 = P.key0; smp.key1 = P.key1; smp.chain = P.chain_offset + cc; smp.major = 0u;
    smp.mode = SM_STAGE1; smp.type = P.type; smp.large = false; smp.sigma2 = P.sigma2; smp.lane = sub;
    smp.stride = 32u;
    smp.u1_off = (uint32_t) D * 32u;
    smp.s2_off = smp.u1_off + D4 * 32u;
    smp.timing_probe = false;
    Counters ct = {0u, 0u, 0u, 0u, 0u};
    PathState ps;
    path_init(P, ps);
    ps.phase = (live && n_mut > 0u) ? PH_DONE : PH_IDLE; // helpers stay PH_IDLE for good
    ps.o = mk3(0.f, 0.f, 0.f); ps.d = mk3(0.f, 0.f, 1.f); ps.tmin = 0.f; ps.tmax = 0.f;
    bool helper_has_ray = false;
    Hit h{-1, 0.f, 0.f, 0.f};
    const int batch = P.mh_batch > 32 ? 32 : P.mh_batch;
    LdsTables LT;
    LT.shade_off = smp.s2_off + D4 * 32u;
    LT.bsdf_off = LT.shade_off + (uint32_t) P.n_shade * 16u;
    LT.emit_off = LT.bsdf_off + (uint32_t) P.n_bsdfs * 12u;
    const GlobalTables GT{P.shade, P.bsdfs, P.emitters};
    if (LDS_TABLES) stage_tables(P, LT, lane);

    // Wave priority by loop section: the two waves of a SIMD are then rarely in the same section with the same claim on the
    // issue port -- the ray loop (dense VALU, its scalar loads pipelined) yields to a partner that is in the latency-bound
    // path step or bookkeeping branch. Measured +5 % on config 2 (any assignment of distinct levels gives most of it;
    // DRMLT_DEBUG bit 1024 switches it off for A/B runs).
    const bool prio = (P.debug & 1024) == 0;
    const bool stamps = (P.debug & 128) != 0;
    unsigned long long t_mh = 0, t_trace = 0, t_step = 0, n_iter = 0, n_mh = 0, n_busy = 0;
    unsigned long long t_decide = 0, t_commit = 0, t_start = 0, t_fill = 0;
    unsigned long long hist[6] = {0, 0, 0, 0, 0, 0}; // iterations by number of chains tracing a closest-hit ray: 0, 1-4, 5-8, 9-16, 17-24, 25-32
#define STAMP() (stamps ? __builtin_amdgcn_s_memtime() : 0ull)
    for (;;) {
        const bool parked = ps.phase == PH_DONE;
        const unsigned long long pmask = __ballot(parked);
        const unsigned long long rmask = __ballot(ps.phase != PH_DONE && ps.phase != PH_IDLE);
        if (!pmask && !rmask) break;
        const unsigned long long s0 = STAMP();
        if (pmask && (__popcll(pmask) >= batch || !rmask)) {
            n_mh++;
            if (prio) __builtin_amdgcn_s_setprio(2);
            // decide (chain lanes) -> commit (both lanes of a pair) -> start (chain lanes) -> draw (both lanes)
            int commit = 0;
            if (parked) commit = mh_decide_splat(P, cs, smp, ps, ct);
            const unsigned long long m1 = STAMP();
            const int commit_pair = (int) from_lower_u((unsigned) commit);
            const uint32_t maj_c = from_lower_u(smp.major);
            const bool large_c = from_lower_u(smp.large ? 1u : 0u) != 0u;
            if (helper) { smp.major = maj_c; smp.large = large_c; }
            if (commit_pair) { // pair-aligned halves: orbital pairs never straddle the split
                const uint32_t split = (((uint32_t) D / 2u) + 1u) & ~1u;
                commit_range(smp, commit_pair, helper ? split : 0u, helper ? (uint32_t) D : split);
            }
            const unsigned long long m2 = STAMP();
            int fill = 0;
            if (parked) fill = mh_start(P, cs, smp, ps, n_mut, mut_base);
            const unsigned long long m3 = STAMP();
            const int fill_pair = (int) from_lower_u((unsigned) fill);
            const uint32_t maj_f = from_lower_u(smp.major);
            const bool large_f = from_lower_u(smp.large ? 1u : 0u) != 0u;
            if (helper) { smp.major = maj_f; smp.large = large_f; }
            if (fill_pair == 1) {
                const uint32_t nb = D4 / 4u, hb = (nb + 1u) / 2u;
                smp.fill_stage1(helper ? hb : 0u, helper ? nb : hb);
            } else if (fill_pair == 2) {
                smp.fill_stage2(D4, helper ? 1u : 0u, 2u);
            }
            const unsigned long long m4 = STAMP();
            t_decide += m1 - s0; t_commit += m2 - m1; t_start += m3 - m2; t_fill += m4 - m3;
        }
        // one ray per lane: chain lanes their camera / bounce ray, helpers the shadow ray they were handed
        const unsigned long long s1 = STAMP();
        const bool tracing = helper ? helper_has_ray : ps.phase == PH_CLOSEST;
        if (stamps) n_busy += __popcll(__ballot(tracing));
        if (stamps) { const int nl = __popcll(__ballot(tracing && !helper)); hist[nl == 0 ? 0 : (nl <= 4 ? 1 : (nl <= 8 ? 2 : (nl <= 16 ? 3 : (nl <= 24 ? 4 : 5))))]++; }
        if (prio) __builtin_amdgcn_s_setprio(0);
        if (tracing) h = trace<FEAT>(P, ps.o, ps.d, ps.tmin, ps.tmax, helper);
        if (prio) __builtin_amdgcn_s_setprio(3);
        const unsigned long long s2 = STAMP();
        const unsigned occluded = from_upper_u((helper_has_ray && h.prim >= 0) ? 1u : 0u);
        helper_has_ray = false;
        ShadowRay sr;
        sr.o = ps.o; sr.d = ps.d; sr.tmin = 0.f; sr.tmax = 0.f; sr.valid = false;
        if (!helper && ps.phase != PH_DONE && ps.phase != PH_IDLE) {
            if (LDS_TABLES) path_step<true, FEAT>(P, LT, ps, smp, h, occluded == 0u, sr);
            else path_step<true, FEAT>(P, GT, ps, smp, h, occluded == 0u, sr);
        }
        // hand the shadow ray of this vertex to the helper lane
        const float ox = from_lower(sr.o.x), oy = from_lower(sr.o.y), oz = from_lower(sr.o.z);
        const float dx = from_lower(sr.d.x), dy = from_lower(sr.d.y), dz = from_lower(sr.d.z);
        const float t0 = from_lower(sr.tmin), t1 = from_lower(sr.tmax);
        const float vf = from_lower(sr.valid ? 1.f : 0.f);
        if (helper) {
            ps.o = mk3(ox, oy, oz); ps.d = mk3(dx, dy, dz); ps.tmin = t0; ps.tmax = t1;
            helper_has_ray = vf != 0.f;
        }
        const unsigned long long s3 = STAMP();
        t_mh += s1 - s0; t_trace += s2 - s1; t_step += s3 - s2; n_iter++;
    }
#undef STAMP
    if (stamps && lane == 0) {
        atomicAdd(P.stats + 16, t_mh); atomicAdd(P.stats + 17, t_trace); atomicAdd(P.stats + 18, t_step);
        atomicAdd(P.stats + 19, n_iter); atomicAdd(P.stats + 20, n_mh); atomicAdd(P.stats + 21, n_busy);
        for (int q = 0; q < 6; ++q) atomicAdd(P.stats + 26 + q, hist[q]);
        atomicAdd(P.stats + 22, t_decide); atomicAdd(P.stats + 23, t_commit); atomicAdd(P.stats + 24, t_start); atomicAdd(P.stats + 25, t_fill);
    }

    if (live) {
        for (int k = 0; k < D; ++k) P.x[(size_t) k * P.n_chains + c] = lds_x[(uint32_t) k * 32u + sub];
        P.cur_lum[c] = cs.cur.lum; P.cur_px[c] = cs.cur.px; P.cur_py[c] = cs.cur.py;
        P.cur_r[c] = cs.cur.r; P.cur_g[c] = cs.cur.g; P.cur_b[c] = cs.cur.b;
    }
    flush_counters(P, ct, lane);
}

// ------------------------------------------------------------------------------------------
// k_mutate_v4: k_mutate_v3's two lanes per chain, without its lock-step rounds.
//
// In v3 the bookkeeping branch fires when all 32 chains of a wave are parked: a round lasts as long as the longest of 32
// paths (9.6 loop iterations per mutation where the mean path needs 3.9), because the branch costs the wave the same
// whether 1 or 32 chains take it -- per lane it walks D_eff dimensions (commit) and ~10 Philox blocks (next draws).
// Here that per-chain work is FLATTENED over the wave: the parked chains are compacted into a list (ballot + prefix
// count -> LDS) and all 64 lanes share the (chain, dimension pair) commit items and the (chain, Philox block) draw
// items, so the branch costs in proportion to the number of chains that take it and chains can run free: a chain starts
// its next evaluation as soon as its last one is digested. What is left per lane is the decision itself.
//
// Film: splats are queued in LDS and flushed by whole waves with the three colour channels of a splat in three adjacent
// lanes (one 12-byte segment per splat at the memory side instead of three scattered dwords), and the current state is
// splatted with its CUMULATIVE weight when it is replaced (or the launch ends) instead of once per mutation -- the
// reference's own PSSMLT loop does the same (pssmlt_proc.cpp:215-226,262-266); splatting is linear, the film is the
// same sum. An accepted proposal's weight starts the cumulative weight of the new current state.
//
// Chains are the same as k_mutate_v2/v3's (same addressed draws, same arithmetic per chain).
#define V4_STRIDE 33u // row stride of the sampler rows: (row + chain) mod 32 banks serve per-chain AND per-dimension access patterns
#define V4_QCAP 160u  // splat queue entries: flushed when a bookkeeping branch (at most 3 x 32 new entries) might not fit
#define V4_STACK32_CAP 11 // LDS entries of a 32-bit traversal stack (the rest spills): 11 + 3 spare rows of 256 B keep eight waves on a CU
#define V4_QCAP_BVH 100u // BVH scenes: their kernel also keeps the traversal stack in LDS (6 KB); flushes are a negligible part of it

struct V4Lds {
    uint32_t coin_off, list_off, q_off; // float offsets into lds_x
    uint32_t qcap;                      // queue entries (row length of the five queue rows)
};

// one colour channel of ImageBlock::put (see film_put): lanes 3s, 3s+1, 3s+2 carry the channels of splat s
DEV void film_put_channel(const DParams &P, float px, float py, float v, int ch) {
    if (P.debug & 1) return;
    float posx = px - 0.5f, posy = py - 0.5f;
    int minx = max((int) ceilf(posx - P.filter_radius), 0), miny = max((int) ceilf(posy - P.filter_radius), 0);
    int maxx = min((int) floorf(posx + P.filter_radius), P.width - 1), maxy = min((int) floorf(posy + P.filter_radius), P.height - 1);
    const bool box = P.box_weight > 0.f;
    for (int y = miny; y <= maxy; ++y) {
        const int iy = min((int) fabsf(((float) y - posy) * P.filter_scale), 31);
        float wy = box ? (iy < 31 ? P.box_weight : 0.f) : P.filter_lut[iy];
        for (int x = minx; x <= maxx; ++x) {
            const int ix = min((int) fabsf(((float) x - posx) * P.filter_scale), 31);
            float w = (box ? (ix < 31 ? P.box_weight : 0.f) : P.filter_lut[ix]) * wy;
            atomic_add_global_f32(P.film + ((size_t) y * P.width + x) * 3 + ch, w * v);
        }
    }
}

// queue a splat (all lanes call; `want` selects). ImageBlock::put's validity test (imageblock.h:155-165) is applied here.
DEV void v4_enqueue(const V4Lds &L, uint32_t &qn, bool want, float px, float py, float r, float g, float b) {
    want = want && isfinite(r) && isfinite(g) && isfinite(b) && r >= 0.f && g >= 0.f && b >= 0.f;
    const unsigned long long m = __ballot(want);
    if (want) {
        const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
        float *q = &lds_x[L.q_off + slot];
        q[0] = px; q[L.qcap] = py; q[2u * L.qcap] = r; q[3u * L.qcap] = g; q[4u * L.qcap] = b;
    }
    qn += (uint32_t) __popcll(m);
}
DEV void v4_flush(const DParams &P, const V4Lds &L, uint32_t &qn, uint32_t lane) {
    const uint32_t s = lane / 3u, ch = lane - 3u * s;
    for (uint32_t base = 0u; base < qn; base += 21u) {
        const uint32_t e = base + s;
        if (e < qn && s < 21u) {
            const float *q = &lds_x[L.q_off + e];
            film_put_channel(P, q[0], q[L.qcap], q[(2u + ch) * L.qcap], (int) ch);
        }
    }
    qn = 0u;
}

// STAMPS: the diagnostic build (DRMLT_DEBUG bit 128) with s_memtime section stamps; a compile-time switch because its
// sixteen 64-bit wave-uniform accumulators would otherwise sit in (and spill from) the scalar registers of the real kernel.
// Everything wave-uniform that k_mutate_v4 derives from the parameter block: recomputed (a handful of scalar ops) by each
// loop section from its own copy of the block, so that none of it occupies scalar registers across sections.
struct V4Layout {
    RowSampler smp;  // uniform fields only; `mode` and `lane` are per lane
    V4Lds L;
    LdsTables LT;
    uint32_t D, D4, nb1;
};
DEV V4Layout v4_layout(const DParams &P, uint32_t qcap) {
    V4Layout Y;
    Y.L.qcap = qcap;
    Y.D = (uint32_t) P.eff_dim;
    Y.D4 = (Y.D + 3u) & ~3u;
    Y.nb1 = Y.D4 / 4u; // first-stage Philox blocks of a mutation; item nb1 of a chain = the coins of its NEXT mutation
    Y.smp.key0 = P.key0; Y.smp.key1 = P.key1;
    Y.smp.mode = SM_STAGE1; Y.smp.type = P.type; Y.smp.sigma2 = P.sigma2; Y.smp.lane = 0u;
    Y.smp.stride = V4_STRIDE;
    Y.smp.y_off = Y.D * V4_STRIDE;
    Y.smp.z_off = Y.smp.y_off + Y.D4 * V4_STRIDE;
    Y.L.coin_off = Y.smp.z_off + Y.D4 * V4_STRIDE;
    Y.L.list_off = Y.L.coin_off + 4u * V4_STRIDE;
    Y.L.q_off = Y.L.list_off + 32u;
    Y.LT.shade_off = (Y.L.q_off + 5u * qcap + 3u) & ~3u;
    Y.LT.bsdf_off = Y.LT.shade_off + (uint32_t) P.n_shade * 16u;
    Y.LT.emit_off = Y.LT.bsdf_off + (uint32_t) P.n_bsdfs * 12u;
    return Y;
}

template <int FEAT, bool LDS_TABLES, bool STAMPS, bool STACK16 = false, bool OVF = false>
__global__ void __launch_bounds__(CHAIN_BLOCK) k_mutate_v4(DParams P, uint32_t n_mut, uint32_t mut_base) {
    // The parameter block is ~80 dwords, most of it used by one loop section only. Left to itself the compiler loads every
    // field it will ever need before the loop and then spills scalar registers into vector lanes all through the loop
    // (a sixth of the kernel's VALU instructions were v_readlane / v_writelane). Each loop section therefore works on its
    // own copy of the block, read through a kernarg pointer the compiler cannot see through: the fields a section uses
    // are scalar loads at its head (scalar cache hits) and dead at its end.
    typedef const DParams __attribute__((address_space(4))) *KArg;
    const KArg kp = (KArg) __builtin_amdgcn_kernarg_segment_ptr();
#define SECTION_PARAMS(name)                                               \
    KArg name##_q = kp;                                                    \
    asm volatile("" : "+s"(name##_q));                                     \
    DParams name;                                                          \
    load_params(name, name##_q)
    const uint32_t lane = threadIdx.x;
    const uint32_t sub = lane & 31u;
    const bool helper = lane >= 32u;
    const uint32_t c = blockIdx.x * 32u + sub;
    const bool live = !helper && c < P.n_chains;
    const uint32_t cc = c < P.n_chains ? c : P.n_chains - 1;
    const uint32_t S = V4_STRIDE;
    constexpr uint32_t QCAP = (FEAT & 8) ? V4_QCAP_BVH : V4_QCAP;
    int smp_mode = SM_STAGE1; // per-lane part of the sampler (which proposal the path in flight reads)
    uint32_t qn = 0u;
    ChainState cs;
    float cum = 0.f; // cumulative weight of the current state since it was adopted
    Counters ct = {0u, 0u, 0u, 0u, 0u};
    PathState ps;
    bool helper_has_ray = false;
    Hit h{-1, 0.f, 0.f, 0.f};
    int batch;
    uint32_t base = 0u, target = 0u, limit = 0u; // mutation index of this chain's first mutation of the launch; see run-ahead below
    bool reported = false;                       // this wave has told the grid that all its chains are at the target
    constexpr bool RESUMABLE = (FEAT & 8) != 0; // BVH scenes: traversals survive loop iterations (device_path.h: Trav)
    Trav T;
    T.active = false; T.cur = 0; T.sp = 0; T.ovf = 0; T.rx = T.ry = T.rz = 0u; T.any_hit = false; T.h = h; T.tmin = 0.f;
    T.o = T.d = T.inv = T.oi = mk3(0.f, 0.f, 0.f);
    trav_reset_counters(T);
    int rstate = 0; // ray of this lane: 0 none, 1 issued, 2 being traversed, 3 result waiting to be consumed
    {
        const V4Layout Y = v4_layout(P, QCAP);
        if (!helper)
            for (uint32_t k = 0; k < Y.D; ++k) lds_x[k * S + sub] = P.x[(size_t) k * P.n_chains + cc];
        cs.cur.lum = P.cur_lum[cc]; cs.cur.px = P.cur_px[cc]; cs.cur.py = P.cur_py[cc];
        cs.cur.r = P.cur_r[cc]; cs.cur.g = P.cur_g[cc]; cs.cur.b = P.cur_b[cc];
        cs.y = cs.cur; cs.z = cs.cur;
        cs.a1 = 0.f; cs.coin_acc1 = cs.coin_acc2 = cs.coin_mix = 0.f;
        cs.it = 0u; cs.nd1 = cs.nd2 = 0u; cs.stage = -1; cs.large = false; cs.do_second = false;
        path_init(P, ps);
        // Run-ahead (P.chain_done): `n_mut` is then the TARGET every chain must have reached when the launch ends, counted from
        // the chain's seeding; a chain that is there keeps mutating -- up to P.run_limit, the render's total -- for as long as
        // some chain of the grid is still short of the target. Every mutation executed is one of the chain's fixed total (its
        // index, hence its random numbers, is the chain's own count), so lanes that would have idled until the slowest chain of
        // the slowest wave is done do useful work instead; the render ends with every chain at exactly its total.
        base = (P.chain_done && live) ? P.chain_done[cc] : mut_base;
        target = P.chain_done ? n_mut : mut_base + n_mut;
        limit = P.chain_done ? P.run_limit : target;
        ps.phase = (live && base < limit) ? PH_DONE : PH_IDLE; // helpers stay PH_IDLE for good
        ps.o = mk3(0.f, 0.f, 0.f); ps.d = mk3(0.f, 0.f, 1.f); ps.tmin = 0.f; ps.tmax = 0.f;
        batch = P.mh_batch > 32 ? 32 : P.mh_batch;
        if (LDS_TABLES) stage_tables(P, Y.LT, lane);
        // coins of every chain's first mutation of this launch (afterwards they are drawn one mutation ahead, beside the proposal)
        if (!helper) {
            const u4 coins = philox4x32_10(P.key0, P.key1, 0u, base, P.chain_offset + blockIdx.x * 32u + sub, TAG_COIN);
            float *dst = &lds_x[Y.L.coin_off + sub];
            dst[0] = u32_to_unit(coins.x); dst[S] = u32_to_unit(coins.y); dst[2u * S] = u32_to_unit(coins.z); dst[3u * S] = u32_to_unit(coins.w);
        }
    }

    constexpr bool prio = true; // wave priority by loop section (see k_mutate_v3)
    constexpr bool stamps = STAMPS;
    unsigned long long t_mh = 0, t_trace = 0, t_step = 0, n_iter = 0, n_mh = 0, n_busy = 0;
    unsigned long long t_decide = 0, t_commit = 0, t_start = 0, t_fill = 0;
    unsigned long long hist[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long dg[6] = {0, 0, 0, 0, 0, 0};
#define STAMP() (stamps ? __builtin_amdgcn_s_memtime() : 0ull)
    for (;;) {
        const bool parked = ps.phase == PH_DONE;
        const unsigned long long pmask = __ballot(parked);
        const unsigned long long rmask = __ballot(ps.phase != PH_DONE && ps.phase != PH_IDLE);
        if (!pmask && !rmask) break;
        const unsigned long long s0 = STAMP();
        if (pmask && (__popcll(pmask) >= batch || !rmask)) {
            n_mh++;
            SECTION_PARAMS(Pm);
            const V4Layout Y = v4_layout(Pm, QCAP);
            const V4Lds &L = Y.L;
            RowSampler smp = Y.smp;
            smp.lane = sub; smp.mode = smp_mode;
            const uint32_t nb1 = Y.nb1, D4 = Y.D4, D = Y.D;
            int *const lds_list = reinterpret_cast<int *>(&lds_x[L.list_off]);
            if (prio) __builtin_amdgcn_s_setprio(2);
            if (qn + 96u > QCAP) { SECTION_PARAMS(Pf); v4_flush(Pf, L, qn, lane); } // room for this branch's splats (at most 3 per chain)
            // ---- decide (parked chain lanes): weights, commit mode, what the chain does next
            int commit = 0, kind = 0; // kind: 0 nothing / finished, 1 next mutation, 2 second stage, 3 Green's reverse
            bool want0 = false, want1 = false, want2 = false;
            float e0x = 0.f, e0y = 0.f, e0r = 0.f, e0g = 0.f, e0b = 0.f;
            float e1x = 0.f, e1y = 0.f, e1r = 0.f, e1g = 0.f, e1b = 0.f;
            float e2x = 0.f, e2y = 0.f, e2r = 0.f, e2g = 0.f, e2b = 0.f;
            if (parked) {
                const MhOutcome o = mh_decide(Pm, cs, smp, ps, ct);
                if (o.decided) {
                    cum += o.w.w0;
                    const bool a1st = o.commit == SM_STAGE1, a2nd = o.commit == SM_STAGE2;
                    // rejected proposals are splatted now, an adopted one carries its weight into `cum`
                    want1 = !a1st && o.w.w1 > 0.f;
                    e1x = cs.y.px; e1y = cs.y.py; e1r = cs.y.r * o.w.w1; e1g = cs.y.g * o.w.w1; e1b = cs.y.b * o.w.w1;
                    want2 = !a2nd && o.w.w2 > 0.f;
                    e2x = cs.z.px; e2y = cs.z.py; e2r = cs.z.r * o.w.w2; e2g = cs.z.g * o.w.w2; e2b = cs.z.b * o.w.w2;
                    if (o.commit) {
                        want0 = cum > 0.f;
                        e0x = cs.cur.px; e0y = cs.cur.py; e0r = cs.cur.r * cum; e0g = cs.cur.g * cum; e0b = cs.cur.b * cum;
                        cum = a1st ? o.w.w1 : o.w.w2;
                        cs.cur = select_splat(a1st, cs.y, cs.z);
                        if (o.amap) { // acceptance map: a mark at the pixel of the state that was LEFT (the weights are all zero)
                            const f3 mc = mh_amap_colour(o.amap);
                            want1 = true; e1x = e0x; e1y = e0y; e1r = mc.x; e1g = mc.y; e1b = mc.z;
                        }
                    }
                    commit = o.commit;
                }
                kind = cs.stage < 0 ? 4 : (cs.stage == 1 ? 2 : 3); // 4: between mutations -- resolved below
            }
            {
                // between mutations: go on while short of the target; beyond it (run-ahead) while anybody in the grid is short
                const uint32_t done_now = base + cs.it;
                const bool under = __ballot(!helper && live && done_now < target) != 0ull;
                bool more = under;
                if (Pm.chain_done) {
                    if (!under && !reported) { reported = true; if (lane == 0) atomicSub(Pm.waves_left, 1u); }
                    if (!under) more = __builtin_amdgcn_readfirstlane((int) __hip_atomic_load(Pm.waves_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0;
                }
                if (kind == 4) kind = (done_now < target || (done_now < limit && more)) ? 1 : 0;
            }
            v4_enqueue(L, qn, want0, e0x, e0y, e0r, e0g, e0b);
            v4_enqueue(L, qn, want1, e1x, e1y, e1r, e1g, e1b);
            v4_enqueue(L, qn, want2, e2x, e2y, e2r, e2g, e2b);
            const unsigned long long m1 = STAMP();

            // ---- commit (DRMLTSampler::accept: uCurrent = wrap(chosen proposal)), flattened: items (accepted chain j, row
            // quad q), chain-minor so that a pass touches as many different chains (banks) as possible
            const uint32_t cmask = (uint32_t) __ballot(commit != 0);
            if (cmask) {
                if (commit) lds_list[__builtin_amdgcn_mbcnt_lo(cmask, 0u)] = (int) sub;
                const uint32_t n = (uint32_t) __popc(cmask), total = n * nb1;
                const float rcp_n = 1.f / (float) n;
                for (uint32_t base = 0u; base < total; base += 64u) {
                    const uint32_t i = base + lane;
                    const bool valid = i < total;
                    const uint32_t ii = valid ? i : 0u;
                    const uint32_t q = (uint32_t) (((float) ii + 0.5f) * rcp_n), j = ii - q * n;
                    const uint32_t cj = (uint32_t) lds_list[j];
                    const int mode_j = __shfl(commit, (int) cj, 64);
                    if (valid) {
                        const float *src = &lds_x[(mode_j == SM_STAGE1 ? smp.y_off : smp.z_off) + 4u * q * S + cj];
                        float *dst = &lds_x[4u * q * S + cj];
                        float v[4]; // (reads first: the writes may alias them for the compiler)
#pragma unroll
                        for (uint32_t r = 0; r < 4u; ++r) v[r] = src[(4u * q + r < D ? r : 0u) * S];
#pragma unroll
                        for (uint32_t r = 0; r < 4u; ++r)
                            if (4u * q + r < D) dst[r * S] = wrap01(v[r]);
                    }
                }
            }
            const unsigned long long m2 = STAMP();

            // ---- start (parked chain lanes): the coins of the mutation that begins were drawn with the previous one
            if (parked && kind == 1) {
                const float *cn = &lds_x[L.coin_off + sub];
                cs.large = cn[0] < Pm.p_large;
                cs.coin_acc1 = cn[S]; cs.coin_acc2 = cn[2u * S]; cs.coin_mix = cn[3u * S];
                cs.stage = 0;
                cs.do_second = false;
                cs.nd1 = cs.nd2 = 0u;
            }
            const unsigned long long m3 = STAMP();

            // ---- proposals of the chains that start a mutation, flattened: items (chain j, Philox block b) -> dimensions
            // 4b..4b+3 of y; block nb1 = the four coins (large step, first / second acceptance, mixture) of the NEXT mutation
            SECTION_PARAMS(Pg);
            const V4Layout Yg = v4_layout(Pg, QCAP);
            RowSampler smg = Yg.smp;
            const uint32_t chain_base_g = Pg.chain_offset + blockIdx.x * 32u;
            const uint32_t maj_mine = base + cs.it; // the mutation in flight (cs.it counts the mutations decided in this launch)
            const unsigned info = cs.large ? 1u : 0u;
            const uint32_t f1mask = (uint32_t) __ballot(kind == 1);
            if (f1mask) {
                if (kind == 1) lds_list[__builtin_amdgcn_mbcnt_lo(f1mask, 0u)] = (int) sub;
                const uint32_t n = (uint32_t) __popc(f1mask), total = n * (nb1 + 1u);
                const float rcp_n = 1.f / (float) n;
                for (uint32_t base = 0u; base < total; base += 64u) {
                    const uint32_t i = base + lane;
                    const bool valid = i < total;
                    const uint32_t ii = valid ? i : 0u;
                    const uint32_t b = (uint32_t) (((float) ii + 0.5f) * rcp_n), j = ii - b * n;
                    const uint32_t cj = (uint32_t) lds_list[j];
                    const uint32_t mj = (uint32_t) __shfl((int) maj_mine, (int) cj, 64);
                    const unsigned inf = (unsigned) __shfl((int) info, (int) cj, 64);
                    if (valid) {
                        if (b < nb1) smg.fill_first(cj, b, mj, chain_base_g + cj, inf != 0u);
                        else {
                            const u4 coins = philox4x32_10(Pg.key0, Pg.key1, 0u, mj + 1u, chain_base_g + cj, TAG_COIN);
                            float *dst = &lds_x[L.coin_off + cj];
                            dst[0] = u32_to_unit(coins.x); dst[S] = u32_to_unit(coins.y); dst[2u * S] = u32_to_unit(coins.z); dst[3u * S] = u32_to_unit(coins.w);
                        }
                    }
                }
            }
            const uint32_t f2mask = (uint32_t) __ballot(kind == 2);
            if (f2mask) { // second-stage proposals (rare: rejected bold steps)
                if (kind == 2) lds_list[__builtin_amdgcn_mbcnt_lo(f2mask, 0u)] = (int) sub;
                // blocks per chain: uniforms for a large step (one per dim), the orbital angles (one per pair), Gaussian pairs otherwise
                const uint32_t nb2 = Pg.type == 2 ? (Pg.timid_after_large ? D4 / 4u : (D4 / 2u + 3u) / 4u) : D4 / 2u;
                const uint32_t n = (uint32_t) __popc(f2mask), total = n * nb2;
                const float rcp_n = 1.f / (float) n;
                for (uint32_t base = 0u; base < total; base += 64u) {
                    const uint32_t i = base + lane;
                    const bool valid = i < total;
                    const uint32_t ii = valid ? i : 0u;
                    const uint32_t b = (uint32_t) (((float) ii + 0.5f) * rcp_n), j = ii - b * n;
                    const uint32_t cj = (uint32_t) lds_list[j];
                    const uint32_t mj = (uint32_t) __shfl((int) maj_mine, (int) cj, 64);
                    const unsigned inf = (unsigned) __shfl((int) info, (int) cj, 64);
                    if (valid) smg.fill_second(cj, b, D4, mj, chain_base_g + cj, inf != 0u);
                }
            }
            const unsigned long long m4 = STAMP();

            // ---- begin the evaluation (parked chain lanes): film position and camera ray from the first two components
            if (parked) {
                if (kind == 0) ps.phase = PH_IDLE;
                else {
                    SECTION_PARAMS(Pb);
                    smp_mode = smp.mode = cs.stage == 0 ? SM_STAGE1 : (cs.stage == 1 ? SM_STAGE2 : SM_REVERSE);
                    path_init(Pb, ps);
                    const float v0 = smp.next(0u), v1 = smp.next(1u);
                    path_begin(Pb, ps, v0, v1);
                    if (RESUMABLE) rstate = 1;
                }
            }
            const unsigned long long m5 = STAMP();
            t_decide += m1 - s0; t_commit += m2 - m1; t_fill += m4 - m3; t_start += (m3 - m2) + (m5 - m4);
        }
        const unsigned long long s1 = STAMP();
        unsigned long long s2 = s1;
        if constexpr (!RESUMABLE) {
            // one ray per lane: chain lanes their camera / bounce ray, helpers the shadow ray they were handed
            const bool tracing = helper ? helper_has_ray : ps.phase == PH_CLOSEST;
            if (stamps) n_busy += __popcll(__ballot(tracing));
            if (stamps) { const int nl = __popcll(__ballot(tracing && !helper)); hist[nl == 0 ? 0 : (nl <= 4 ? 1 : (nl <= 8 ? 2 : (nl <= 16 ? 3 : (nl <= 24 ? 4 : 5))))]++; }
            if (prio) __builtin_amdgcn_s_setprio(0);
            {
                SECTION_PARAMS(Pt);
                if (tracing) h = trace<FEAT>(Pt, ps.o, ps.d, ps.tmin, ps.tmax, helper);
            }
            if (prio) __builtin_amdgcn_s_setprio(3);
            s2 = STAMP();
            const unsigned occluded = from_upper_u((helper_has_ray && h.prim >= 0) ? 1u : 0u);
            helper_has_ray = false;
            ShadowRay sr;
            sr.o = ps.o; sr.d = ps.d; sr.tmin = 0.f; sr.tmax = 0.f; sr.valid = false;
            {
                SECTION_PARAMS(Ps);
                if (!helper && ps.phase != PH_DONE && ps.phase != PH_IDLE) {
                    const V4Layout Y = v4_layout(Ps, QCAP);
                    RowSampler smp = Y.smp;
                    smp.lane = sub; smp.mode = smp_mode;
                    if (LDS_TABLES) path_step<true, FEAT, RowSampler, LdsTables, false>(Ps, Y.LT, ps, smp, h, occluded == 0u, sr);
                    else path_step<true, FEAT, RowSampler, GlobalTables, false>(Ps, GlobalTables{Ps.shade, Ps.bsdfs, Ps.emitters}, ps, smp, h, occluded == 0u, sr);
                }
            }
            // hand the shadow ray of this vertex to the helper lane
            const float ox = from_lower(sr.o.x), oy = from_lower(sr.o.y), oz = from_lower(sr.o.z);
            const float dx = from_lower(sr.d.x), dy = from_lower(sr.d.y), dz = from_lower(sr.d.z);
            const float t0 = from_lower(sr.tmin), t1 = from_lower(sr.tmax);
            const float vf = from_lower(sr.valid ? 1.f : 0.f);
            if (helper) {
                ps.o = mk3(ox, oy, oz); ps.d = mk3(dx, dy, dz); ps.tmin = t0; ps.tmax = t1;
                helper_has_ray = vf != 0.f;
            }
        } else {
            // ---- BVH scenes. A lane's traversal continues across iterations; a slice ends when every traversal is done
            // or `trace_yield` lanes have finished theirs. A chain lane steps when its own ray AND its partner's shadow
            // ray (of the previous vertex) are both done; everybody else just keeps traversing next time round.
            if (prio) __builtin_amdgcn_s_setprio(0);
            if (P.debug & 1024) { // diagnostic: where the 64 lanes are when a traversal slice starts
                dg[0]++; dg[1] += __popcll(__ballot(rstate == 1 || rstate == 2)); dg[2] += __popcll(__ballot(!helper && rstate == 3));
                dg[3] += __popcll(__ballot(!helper && ps.phase == PH_DONE)); dg[4] += __popcll(__ballot(helper && rstate == 0));
                dg[5] += __popcll(__ballot(!helper && ps.phase == PH_FLUSH));
            }
            {
                SECTION_PARAMS(Pt);
                if (rstate == 1) { trav_begin(T, ps.o, ps.d, ps.tmin, ps.tmax, helper); rstate = 2; }
                if (STACK16) trav_run<short, DParams, OVF, BVH_STACK, FEAT>(Pt, T, rstate == 2, Pt.trace_yield);
                else trav_run<int, DParams, true, V4_STACK32_CAP, FEAT>(Pt, T, rstate == 2, Pt.trace_yield); // always with the overflow paths
                if (rstate == 2 && !T.active) rstate = 3;
            }
            if (prio) __builtin_amdgcn_s_setprio(3);
            s2 = STAMP();
            const unsigned partner_busy = from_upper_u((rstate == 1 || rstate == 2) ? 1u : 0u);
            const unsigned occluded = from_upper_u((rstate == 3 && T.h.prim >= 0) ? 1u : 0u);
            const bool ready = !helper && partner_busy == 0u && ((ps.phase == PH_CLOSEST && rstate == 3) || ps.phase == PH_FLUSH);
            ShadowRay sr;
            sr.o = ps.o; sr.d = ps.d; sr.tmin = 0.f; sr.tmax = 0.f; sr.valid = false;
            {
                SECTION_PARAMS(Ps);
                if (ready) {
                    h = T.h;
                    rstate = 0;
                    const V4Layout Y = v4_layout(Ps, QCAP);
                    RowSampler smp = Y.smp;
                    smp.lane = sub; smp.mode = smp_mode;
                    if (LDS_TABLES) path_step<true, FEAT, RowSampler, LdsTables, false>(Ps, Y.LT, ps, smp, h, occluded == 0u, sr);
                    else path_step<true, FEAT, RowSampler, GlobalTables, false>(Ps, GlobalTables{Ps.shade, Ps.bsdfs, Ps.emitters}, ps, smp, h, occluded == 0u, sr);
                    if (ps.phase == PH_CLOSEST) rstate = 1;
                }
            }
            // the partner's result has been consumed; hand it the shadow ray of this vertex, if any
            const unsigned stepped = from_lower_u(ready ? 1u : 0u);
            const float ox = from_lower(sr.o.x), oy = from_lower(sr.o.y), oz = from_lower(sr.o.z);
            const float dx = from_lower(sr.d.x), dy = from_lower(sr.d.y), dz = from_lower(sr.d.z);
            const float t0 = from_lower(sr.tmin), t1 = from_lower(sr.tmax);
            const float vf = from_lower(sr.valid ? 1.f : 0.f);
            if (helper && stepped != 0u) {
                rstate = 0;
                if (vf != 0.f) { ps.o = mk3(ox, oy, oz); ps.d = mk3(dx, dy, dz); ps.tmin = t0; ps.tmax = t1; rstate = 1; }
            }
        }
        const unsigned long long s3 = STAMP();
        t_mh += s1 - s0; t_trace += s2 - s1; t_step += s3 - s2; n_iter++;
    }
#undef STAMP
#undef SECTION_PARAMS
    // "Perform the last splat": the current states with what they have accumulated since they were adopted
    const V4Layout Y = v4_layout(P, QCAP);
    v4_enqueue(Y.L, qn, live && cum > 0.f, cs.cur.px, cs.cur.py, cs.cur.r * cum, cs.cur.g * cum, cs.cur.b * cum);
    v4_flush(P, Y.L, qn, lane);
    if (stamps && lane == 0) {
        atomicAdd(P.stats + 16, t_mh); atomicAdd(P.stats + 17, t_trace); atomicAdd(P.stats + 18, t_step);
        atomicAdd(P.stats + 19, n_iter); atomicAdd(P.stats + 20, n_mh); atomicAdd(P.stats + 21, n_busy);
        for (int q = 0; q < 6; ++q) atomicAdd(P.stats + 26 + q, hist[q]);
        atomicAdd(P.stats + 22, t_decide); atomicAdd(P.stats + 23, t_commit); atomicAdd(P.stats + 24, t_start); atomicAdd(P.stats + 25, t_fill);
    }

    if (live) {
        for (uint32_t k = 0; k < Y.D; ++k) P.x[(size_t) k * P.n_chains + c] = lds_x[k * S + sub];
        P.cur_lum[c] = cs.cur.lum; P.cur_px[c] = cs.cur.px; P.cur_py[c] = cs.cur.py;
        P.cur_r[c] = cs.cur.r; P.cur_g[c] = cs.cur.g; P.cur_b[c] = cs.cur.b;
        if (P.chain_done) P.chain_done[c] = base + cs.it;
    }
    if (P.chain_done && !reported && lane == 0) atomicSub(P.waves_left, 1u); // (a wave none of whose chains had anything to do)
    flush_counters(P, ct, lane);
    const unsigned long long decided = wave_sum((live && !helper) ? cs.it : 0u); // mutations decided in this launch (with run-ahead: not n_mut per chain)
    if (lane == 0) atomicAdd(P.stats + 9, decided);
    if (RESUMABLE) {
        const unsigned long long nn = wave_sum(T.n_nodes), np = wave_sum(T.n_prims);
        if (lane == 0) { atomicAdd(P.stats + 10, nn); atomicAdd(P.stats + 11, np); atomicAdd(P.stats + 12, (unsigned long long) T.it_inner); atomicAdd(P.stats + 13, (unsigned long long) T.it_leaf); }
        if ((P.debug & 1024) && lane == 0)
            for (int q = 0; q < 6; ++q) atomicAdd(P.stats + 20 + q, dg[q]);
    }
}

// ------------------------------------------------------------------------------------------
// k_mutate_v5: the chain loop with MORE RAYS THAN LANES (all three types).
//
// In k_mutate_v4 a ray belongs to a lane: chain lane i traverses its camera / bounce ray, helper lane 32 + i the shadow ray
// of the same vertex. On a scene that is traversed (not looped over) the wave then advances ~22 of its 64 lanes per node
// iteration: helpers mostly have no ray (no NEE at a vertex, the path parked or in its bookkeeping), chains wait for their
// partner, and a slice drains towards its longest ray. Here rays and lanes are decoupled:
//   * 64 chains per wave, chain c = lane c for everything that is per chain (path state, acceptance logic);
//   * every ray a path step issues -- the bounce ray and, beside it, the shadow ray of the same vertex -- goes into a RAY POOL
//     in LDS (slot c: chain c's closest-hit ray, slot 64 + c: its shadow ray; origin, direction, interval; the result is
//     written over the record) and its slot number into a FIFO of pending rays, appended with ballot + prefix count
//     (wave-level compaction of the lanes that have something to add);
//   * the traversal loop runs over whatever rays the pool holds: a lane whose traversal finishes writes the result to the
//     slot, and the idle lanes take the next pending slots off the queue (ballot + prefix count again) INSIDE the loop; a
//     phase ends when the pool is dry or `trace_yield` closest-hit rays have finished -- their chains then step (any lane
//     whose chain has its results) and refill the pool.
// With 64 chains a wave holds ~75-80 rays, so the queue keeps the lanes fed until a phase is nearly over.
//
// LDS (20 KB per wave = eight waves per CU = two per SIMD with 131 072 chains): v4's three row groups (x, y, z: 3 x 9 KB for
// 64 chains) cannot sit beside the pool. The CURRENT STATE x therefore stays at home in device memory ([dim][chain], the
// layout it has between launches anyway): it is read when a mutation's proposal is made (flattened over the wave, chain-minor:
// partly coalesced 256 B rows) and written when a proposal is adopted -- SURVEY 8(d)'s B_state, now real traffic (~250 B per
// mutation, L2 / MALL resident). LDS holds ONE row group: the proposal under evaluation -- y, overwritten in place by z when a
// chain enters its second stage. The orbital rule needs only the luminances of y afterwards; Green's reverse move and Mira's
// ratio need x, y and z together and RECOMPUTE what the rows no longer hold from the state and the addressed stream (Green:
// v5_iid_second_again, flattened; Mira: PoolRowSampler::y_raw, per deciding lane -- a twentieth of the mutations get that far).
// Bookkeeping is v4's: decide per lane, commit / proposals / coins flattened over the 64 lanes. Same addressed draws, same arithmetic per component: the same chains.
#define V5_QCAP 96u        // splat queue entries (a round of the bookkeeping branch adds at most 64: flushed in between)
#define V5_QCAP_STACK32 0u  // the builds with 32-bit traversal stacks splat straight from the bookkeeping branch: their LDS goes to the stack column
#define V5_STACK32_CAP 25 // (no splat queue, coins drawn per lane instead of kept in four rows: 28 rows of 256 B for the column; measured on 50 000 /
                          // 1 000 000 triangles: 11 entries 2.23e8 / 5.65e7, 16 2.48e8 / 7.03e7, 20 2.69e8 / 7.62e7)
#define V5_SLOTS 128u
#ifndef V5_ROWS_MEM_WAVES
#define V5_ROWS_MEM_WAVES 3 // waves per SIMD the ROWS_MEM builds are compiled for (registers) and sized for (LDS)
#endif
#ifndef V5_ROWS_MEM_STACK32_CAP
#define V5_ROWS_MEM_STACK32_CAP 25 // as the rows-in-LDS build (larger columns cost the twelfth wave: LDS is granted in steps)
#endif
enum { RS_IDLE = 0, RS_BUSY = 1, RS_DONE = 2 };

struct V5Lds {
    uint32_t coin_off, list_off; // 4 rows of coins (one mutation ahead), the compacted chain list of the bookkeeping branch
    uint32_t q_off;              // splat queue: 5 rows of V5_QCAP floats
    uint32_t pool_off;           // ray pool: 8 rows of V5_SLOTS floats (ox oy oz dx dy dz tmin tmax; result over rows 0..3)
    uint32_t ring_off;           // pending slots, FIFO: V5_SLOTS bytes
    uint32_t status_off;         // RS_* per slot: V5_SLOTS bytes
};
DEV V5Lds v5_layout(uint32_t D, uint32_t qcap, bool coin_rows) {
    V5Lds L;
    L.coin_off = D * 64u;
    L.list_off = L.coin_off + (coin_rows ? 4u * 64u : 0u);
    L.q_off = L.list_off + 64u;
    L.pool_off = L.q_off + 5u * qcap;
    L.ring_off = L.pool_off + 8u * V5_SLOTS;
    L.status_off = L.ring_off + V5_SLOTS / 4u;
    return L;
}
static size_t v5_lds_bytes(uint32_t D, uint32_t qcap, bool coin_rows) { return ((size_t) D * 64u + (coin_rows ? 4u * 64u : 0u) + 64u + 5u * qcap + 8u * V5_SLOTS + 2u * (V5_SLOTS / 4u)) * sizeof(float); } // (+ the scene tables, when they are staged)

// Where a wave's proposal rows live. RowsLds: 64 columns of lds_x (one per chain of the wave). RowsMem: device memory, [dim][chain]
// beside the state -- the builds that give the rows' 8.7 KB of LDS (and a few registers) for a THIRD wave per SIMD on scenes that
// are traversed: there the wave is parked on node fetches more than half of its time, and what covers a fetch is another wave
// (1 -> 2 waves per SIMD: x 1.69 on 1 000 000 triangles, x 1.58 on 50 000). A path step reads its handful of components
// through the L2s instead of LDS: one more fetch beside the ~60 node fetches of the ray it follows.
struct RowsLds {
    DEV float get(uint32_t k, uint32_t col) const { return lds_x[k * 64u + col]; }
    DEV void put(uint32_t k, uint32_t col, float v) const { lds_x[k * 64u + col] = v; }
};
struct RowsMem {
    typedef float __attribute__((address_space(1))) *GPtr;
    GPtr base;       // column 0 of this wave: P.rows + wave_base
    uint32_t stride; // P.n_chains
    // (plain loads and stores: with the streaming hint -- nt, to keep the L2 lines for the BVH -- the row reads themselves miss:
    // 50 000 triangles 3.13e8 -> 2.79e8, 1 000 000 9.16e7 -> 8.52e7. The same hint on the STATE instead -- read once per mutation,
    // a whole mutation of every wave of the XCD apart: it does not survive in the L2 anyway -- loses as well: Cornell at 196 608
    // chains 2.54e9 -> 2.18e9, door 1.09e9 -> 1.05e9, 50 000 triangles 3.45e8 -> 3.26e8. nt bypasses more than the L2.)
    DEV float get(uint32_t k, uint32_t col) const { return base[(size_t) k * stride + col]; }
    DEV void put(uint32_t k, uint32_t col, float v) const { base[(size_t) k * stride + col] = v; }
};

// the proposal rows as the path step sees them: whatever stage is being evaluated sits in the one row group
template <class Rows> struct PoolRowSampler {
    static constexpr bool batch_draws = true; // (device_path.h: path_step reads a step's components together)
    uint32_t lane;
    Rows rows;
    // Mira's ratio alone looks behind the rows (they hold z by then): the state in device memory and the first-stage draws,
    // set by the kernel before a decision (mira_*), block cache of the TAG_S1 stream
    const float *mira_x = nullptr;
    size_t mira_stride = 0;
    uint32_t mira_k0 = 0u, mira_k1 = 0u, mira_major = 0u, mira_chain = 0u, cached = 0xffffffffu;
    u4 blk = {0u, 0u, 0u, 0u};
    DEV void reset_caches() { cached = 0xffffffffu; }
    DEV float row(uint32_t k) const { return rows.get(k, lane); }
    DEV float next(uint32_t k) const { return wrap01(row(k)); }
    DEV float x(uint32_t k) const { return load_global_f32(mira_x + (size_t) k * mira_stride); }
    DEV float y_raw(uint32_t k) { // iid Kelemen step, small (the only caller: mira_ratio)
        FP_STRICT;
        if (cached != (k >> 2)) { cached = k >> 2; blk = philox4x32_10(mira_k0, mira_k1, cached, mira_major, mira_chain, TAG_S1); }
        // (values, not loads: a select between two single-use loads of this struct's fields becomes a load through a selected
        // ADDRESS and the whole sampler -- 60 bytes per lane -- then lives in scratch memory: VERDICT r03 #8)
        uint32_t bx = blk.x, by = blk.y, bz = blk.z, bw = blk.w;
        asm volatile("" : "+v"(bx), "+v"(by), "+v"(bz), "+v"(bw));
        const uint32_t c = k & 3u, w = (c & 2u) ? ((c & 1u) ? bw : bz) : ((c & 1u) ? by : bx);
        return x(k) + kelemen_sample(u32_to_unit(w), KELEMEN_S2);
    }
    DEV float z_raw(uint32_t k) const { return row(k); }
};

// first-stage proposal of the chain in column `col` (state column `xcol` of P.x), dimensions 4b .. 4b+3, from Philox block b
// (the four state components of the block are read by the CALLER, one pass of the flattened loop ahead: v5_state4)
struct State4 { float x0, x1, x2, x3; };
DEV State4 v5_state4(const DParams &P, uint32_t D, size_t xcol, uint32_t b) {
    const bool hi = 4u * b + 2u < D; // D is even: a block holds two pairs or, at the end of the vector, one
    const float *xs = P.x + (size_t) (4u * b) * P.n_chains + xcol;
    State4 X;
    X.x0 = load_global_f32(xs); X.x1 = load_global_f32(xs + P.n_chains);
    X.x2 = load_global_f32(xs + (hi ? 2u : 0u) * (size_t) P.n_chains); X.x3 = load_global_f32(xs + (hi ? 3u : 1u) * (size_t) P.n_chains);
    return X;
}
template <class Rows> DEV void v5_fill_first(const DParams &P, const Rows &rows, uint32_t D, uint32_t col, const State4 &X, uint32_t b, uint32_t major, uint32_t chain, bool large) {
    FP_STRICT;
    const u4 r = philox4x32_10(P.key0, P.key1, b, major, chain, TAG_S1);
    const float u0 = u32_to_unit(r.x), u1 = u32_to_unit(r.y), u2 = u32_to_unit(r.z), u3 = u32_to_unit(r.w);
    const bool hi = 4u * b + 2u < D;
    const float x0 = X.x0, x1 = X.x1, x2 = X.x2, x3 = X.x3;
    float y0, y1, y2, y3;
    if (P.type == 2) { // pairwise orbital: radius from the Kelemen kernel (x 1.9), uniform angle (drmlt_sampler.cpp:354-361)
        const float d0 = kelemen_sample(u0, KELEMEN_S2 * ORBITAL_SCALE), d1 = kelemen_sample(u2, KELEMEN_S2 * ORBITAL_SCALE);
        y0 = fmaf(d0, cos_rev(u1), x0); y1 = fmaf(d0, cos_rev(u1 - 0.25f), x1);
        y2 = fmaf(d1, cos_rev(u3), x2); y3 = fmaf(d1, cos_rev(u3 - 0.25f), x3);
    } else { // iid Kelemen kernel (Green, Mira)
        y0 = x0 + kelemen_sample(u0, KELEMEN_S2); y1 = x1 + kelemen_sample(u1, KELEMEN_S2);
        y2 = x2 + kelemen_sample(u2, KELEMEN_S2); y3 = x3 + kelemen_sample(u3, KELEMEN_S2);
    }
    rows.put(4u * b, col, large ? u0 : y0); rows.put(4u * b + 1u, col, large ? u1 : y1);
    if (hi) { rows.put(4u * b + 2u, col, large ? u2 : y2); rows.put(4u * b + 3u, col, large ? u3 : y3); }
}
// second-stage proposal from Philox block b of the TAG_S2 stream, written OVER the first-stage rows: a large step
// (timidAfterLarge) -> dims 4b .. 4b+3 (uniforms); orbital -> the angles of pairs 4b .. 4b+3 = dims 8b .. 8b+7 (reads the y rows
// it replaces); iid kernels -> the Gaussian perturbations of dims 2b, 2b+1 (draws 2k, 2k+1 belong to dim k)
template <class Rows> DEV void v5_fill_second(const DParams &P, const Rows &rows, uint32_t D, uint32_t col, size_t xcol, uint32_t b, uint32_t major, uint32_t chain, bool large) {
    FP_STRICT;
    const u4 r = philox4x32_10(P.key0, P.key1, b, major, chain, TAG_S2);
    const float u[4] = {u32_to_unit(r.x), u32_to_unit(r.y), u32_to_unit(r.z), u32_to_unit(r.w)};
    if (large) {
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i)
            if (4u * b + i < D) rows.put(4u * b + i, col, u[i]);
        return;
    }
    if (P.type != 2) {
        const uint32_t k = 2u * b; // (k + 1 < D: the caller's block count)
        const float x0 = load_global_f32(P.x + (size_t) k * P.n_chains + xcol), x1 = load_global_f32(P.x + (size_t) (k + 1u) * P.n_chains + xcol);
        rows.put(k, col, x0 + gaussian_sample(u[0], u[1], P.sigma2));
        rows.put(k + 1u, col, x1 + gaussian_sample(u[2], u[3], P.sigma2));
        return;
    }
    // (all loads first: the stores below may alias them for the compiler, and with the rows in device memory every pair would
    // otherwise wait for its own round trip behind the previous pair's stores)
    float xa[4], xb[4], ya[4], yb[4];
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) {
        const uint32_t k0 = 2u * (4u * b + i);
        const uint32_t kk = k0 + 1u < D ? k0 : 0u;
        xa[i] = load_global_f32(P.x + (size_t) kk * P.n_chains + xcol); xb[i] = load_global_f32(P.x + (size_t) (kk + 1u) * P.n_chains + xcol);
        ya[i] = rows.get(kk, col); yb[i] = rows.get(kk + 1u, col);
    }
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) {
        const uint32_t k0 = 2u * (4u * b + i);
        if (k0 + 1u < D) {
            const float x0 = xa[i], x1 = xb[i], y0 = ya[i], y1 = yb[i];
            // theta ~ wrapped Cauchy by inverse CDF (transition.h:157-173); z = y + R(theta)(x - y) (drmlt_sampler.cpp:374-391)
            float xi = u[i], sign = 1.f;
            if (xi < 0.5f) { xi *= 2.f; } else { sign = -1.f; xi = 2.f * (xi - 0.5f); }
            const float V = cos_rev(xi);
            const float A = fminf(1.f, fmaxf(-1.f, (V + WC_DISPERSION) / (1.f + WC_DISPERSION * V)));
            const float ct = A, st = sign * sqrtf(fmaxf(0.f, 1.f - A * A));
            const float dx0 = x0 - y0, dx1 = x1 - y1;
            rows.put(k0, col, y0 + (ct * dx0 - st * dx1));
            rows.put(k0 + 1u, col, y1 + (st * dx0 + ct * dx1));
        }
    }
}
// Green & Mira's reverse move and the adoption of a second-stage proposal under Green, for the iid kernels: dims 2b, 2b+1
// from Philox block b of TAG_S2 (z = x + g), block b / 2 of TAG_S1 (y = x + kelemen) and the state in device memory.
//   reverse:  rows := y* = z - (y - x)   (drmlt_proc.cpp:588-598; the rows held z, which is recomputed, not read)
//   !reverse: state := wrap(z)            (the rows hold y* by then: DRMLTSampler::accept(second), drmlt_sampler.cpp:189-199)
// A large step (timidAfterLarge): y and z are the uniforms themselves, dims 4b .. 4b+3 of block b of either stream.
template <class Rows> DEV void v5_iid_second_again(const DParams &P, const Rows &rows, uint32_t D, uint32_t col, size_t xcol, uint32_t b, uint32_t major, uint32_t chain, bool large, bool reverse) {
    FP_STRICT;
    const u4 r2 = philox4x32_10(P.key0, P.key1, b, major, chain, TAG_S2);
    const float u2[4] = {u32_to_unit(r2.x), u32_to_unit(r2.y), u32_to_unit(r2.z), u32_to_unit(r2.w)};
    if (large) {
        u4 r1 = r2;
        if (reverse) r1 = philox4x32_10(P.key0, P.key1, b, major, chain, TAG_S1);
        const float u1[4] = {u32_to_unit(r1.x), u32_to_unit(r1.y), u32_to_unit(r1.z), u32_to_unit(r1.w)};
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) {
            const uint32_t k = 4u * b + i;
            if (k < D) {
                float *xg = P.x + (size_t) k * P.n_chains + xcol;
                if (reverse) rows.put(k, col, u2[i] - (u1[i] - load_global_f32(xg)));
                else *xg = wrap01(u2[i]);
            }
        }
        return;
    }
    const uint32_t k = 2u * b;
    float *xg0 = P.x + (size_t) k * P.n_chains + xcol, *xg1 = xg0 + P.n_chains;
    const float x0 = load_global_f32(xg0), x1 = load_global_f32(xg1);
    const float z0 = x0 + gaussian_sample(u2[0], u2[1], P.sigma2), z1 = x1 + gaussian_sample(u2[2], u2[3], P.sigma2);
    if (!reverse) { *xg0 = wrap01(z0); *xg1 = wrap01(z1); return; }
    const u4 r1 = philox4x32_10(P.key0, P.key1, k >> 2, major, chain, TAG_S1);
    const float ua = (k & 2u) ? u32_to_unit(r1.z) : u32_to_unit(r1.x), ub = (k & 2u) ? u32_to_unit(r1.w) : u32_to_unit(r1.y);
    const float y0 = x0 + kelemen_sample(ua, KELEMEN_S2), y1 = x1 + kelemen_sample(ub, KELEMEN_S2);
    rows.put(k, col, z0 - (y0 - x0));
    rows.put(k + 1u, col, z1 - (y1 - x1));
}

// FEAT & 8 (BVH scenes): the traversal loop described above. Flat scenes (FEAT without bit 3; LDS_TABLES: their shading /
// BSDF / emitter tables staged in LDS) run the same kernel with the wave-uniform brute-force loop as their "trace phase":
// up to 64 rays off the queue per pass, every one of them done when the pass ends -- so nearly all 64 chains step together
// (k_mutate_v4 steps at most its 32 chain lanes; the helper lanes idle through the path step).
// ROWS_MEM (chains for more than two waves per SIMD; traversed scenes and flat ones with their tables in LDS): proposal rows in device
// memory (RowsMem), registers for three waves (the flat builds need 169 - 179 as they are: 0 - 4 spilled).
template <int FEAT, bool STACK16, bool OVF, bool STAMPS = false, bool LDS_TABLES = false, bool ROWS_MEM = false>
__global__ void __launch_bounds__(CHAIN_BLOCK, ROWS_MEM ? V5_ROWS_MEM_WAVES : 2) k_mutate_v5(DParams P, uint32_t n_mut, uint32_t mut_base) {
    constexpr bool FLAT = (FEAT & 8) == 0;
    typedef typename std::conditional<ROWS_MEM, RowsMem, RowsLds>::type RowsT;
    typedef PoolRowSampler<RowsT> SamplerT;
    constexpr uint32_t QCAP = (FLAT || STACK16) ? V5_QCAP : V5_QCAP_STACK32;
    constexpr bool COIN_ROWS = FLAT || STACK16; // coins drawn one mutation ahead by the flattened proposal pass (else: per lane, when a mutation starts)
    // per-section copies of the parameter block, read through a kernarg pointer the compiler cannot see through (see k_mutate_v4):
    // the fields a section uses are scalar loads at its head and dead at its end, instead of ~200 spilled scalar registers
    typedef const DParams __attribute__((address_space(4))) *KArg;
    const KArg kp = (KArg) __builtin_amdgcn_kernarg_segment_ptr();
#define SECTION_PARAMS(name)                                               \
    KArg name##_q = kp;                                                    \
    asm volatile("" : "+s"(name##_q));                                     \
    DParams name;                                                          \
    load_params(name, name##_q)
    SECTION_PARAMS(P0);
    const uint32_t lane = threadIdx.x;
    const uint32_t wave_base = blockIdx.x * 64u;
    const uint32_t c = wave_base + lane;
    const bool live = c < P0.n_chains;
    const uint32_t cc = live ? c : P0.n_chains - 1;
    const uint32_t D = (uint32_t) P0.eff_dim, nb1 = (D + 3u) / 4u;
    const V5Lds L = v5_layout(ROWS_MEM ? 0u : D, QCAP, COIN_ROWS);
    const V4Lds Lq{0u, 0u, L.q_off, QCAP};
    unsigned char *const ring = reinterpret_cast<unsigned char *>(&lds_x[L.ring_off]);
    unsigned char *const status = reinterpret_cast<unsigned char *>(&lds_x[L.status_off]);
    int *const lds_list = reinterpret_cast<int *>(&lds_x[L.list_off]);
    float *const pool = &lds_x[L.pool_off];
    status[lane] = RS_IDLE; status[64u + lane] = RS_IDLE;
    LdsTables LT;
    LT.shade_off = L.status_off + V5_SLOTS / 4u;
    LT.bsdf_off = LT.shade_off + (uint32_t) P0.n_shade * 16u;
    LT.emit_off = LT.bsdf_off + (uint32_t) P0.n_bsdfs * 12u;
    if (LDS_TABLES) stage_tables(P0, LT, lane);
    // traversed scenes: BSDF / emitter records and the emitters' shape records behind the pool when the launcher found room for them
    HybridTables HT;
    HT.L.shade_off = L.status_off + V5_SLOTS / 4u;
    HT.L.bsdf_off = HT.L.shade_off + (uint32_t) P0.n_emitters * 16u;
    HT.L.emit_off = HT.L.bsdf_off + (uint32_t) P0.n_bsdfs * 12u;
    HT.lds = !FLAT && P0.small_tables_lds != 0;
    if (HT.lds) stage_bsdfs_emitters(P0, HT.L, lane);

    ChainState cs;
    cs.cur.lum = P0.cur_lum[cc]; cs.cur.px = P0.cur_px[cc]; cs.cur.py = P0.cur_py[cc];
    cs.cur.r = P0.cur_r[cc]; cs.cur.g = P0.cur_g[cc]; cs.cur.b = P0.cur_b[cc];
    cs.y = cs.cur; cs.z = cs.cur;
    cs.a1 = 0.f; cs.coin_acc1 = cs.coin_acc2 = cs.coin_mix = 0.f;
    cs.it = 0u; cs.nd1 = cs.nd2 = 0u; cs.stage = -1; cs.large = false; cs.do_second = false;
    float cum = 0.f; // weight of the current state since it was adopted (one splat per residence, as k_mutate_v4)
    uint32_t qn = 0u;
    Counters ct = {0u, 0u, 0u, 0u, 0u};
    RowsT rows;
    if constexpr (ROWS_MEM) { rows.base = (RowsMem::GPtr) (uintptr_t) (P0.rows + wave_base); rows.stride = P0.n_chains; }
    SamplerT smp{lane, rows};

    PathState ps;
    path_init(P0, ps);
    ps.o = mk3(0.f, 0.f, 0.f); ps.d = mk3(0.f, 0.f, 1.f); ps.tmin = 0.f; ps.tmax = 0.f;
    // run-ahead between the launches of a call: as k_mutate_v4 (chain_done / waves_left / run_limit)
    const uint32_t base = (P0.chain_done && live) ? P0.chain_done[cc] : mut_base;
    const uint32_t target = P0.chain_done ? n_mut : mut_base + n_mut;
    const uint32_t limit = P0.chain_done ? P0.run_limit : target;
    bool reported = false;
    ps.phase = (live && base < limit) ? PH_DONE : PH_IDLE;
    const int batch = P0.mh_batch > 64 ? 64 : P0.mh_batch;
    if (COIN_ROWS) { // coins of every chain's first mutation of this launch (afterwards they are drawn one mutation ahead, beside the proposal)
        const u4 coins = philox4x32_10(P0.key0, P0.key1, 0u, base, P0.chain_offset + cc, TAG_COIN);
        float *dst = &lds_x[L.coin_off + lane];
        dst[0] = u32_to_unit(coins.x); dst[64] = u32_to_unit(coins.y); dst[128] = u32_to_unit(coins.z); dst[192] = u32_to_unit(coins.w);
    }

    // traversal: this lane's column of the stack, the ray it is working on (`slot`), the FIFO of pending slots
    typedef typename std::conditional<STACK16, short, int>::type StackT;
    constexpr int CAP = STACK16 ? BVH_STACK : (ROWS_MEM ? V5_ROWS_MEM_STACK32_CAP : V5_STACK32_CAP);
    __shared__ StackT v5_stack[FLAT ? 1 : (CAP + 3) * 64];
    StackT *const my_stack = v5_stack + (FLAT ? 0u : lane);
    Trav T;
    T.active = false; T.cur = 0; T.sp = 0; T.ovf = 0; T.rx = T.ry = T.rz = 0u; T.any_hit = false; T.h = Hit{-1, 0.f, 0.f, 0.f}; T.tmin = 0.f;
    T.o = T.d = T.inv = T.oi = mk3(0.f, 0.f, 0.f);
    trav_reset_counters(T);
    uint32_t slot = 0u;
    uint32_t q_head = 0u, q_count = 0u; // wave-uniform
    unsigned long long n_phase = 0ull, n_refill = 0ull, n_lanes_at_start = 0ull;
    unsigned long long t_mh = 0ull, t_step = 0ull, t_trace = 0ull, n_outer = 0ull, n_mh = 0ull, n_stepping = 0ull, n_parked = 0ull;
#define STAMP5() (STAMPS ? __builtin_amdgcn_s_memtime() : 0ull)
    auto prefix = [](unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u)); };

    // a ray goes into the pool slot of its chain (slot c: closest hit, 64 + c: shadow) and the slot number into the FIFO: ray
    // compaction by ballot + prefix count over the lanes that issued one
    auto push_rays = [&](bool push_c, bool push_s, const ShadowRay &sr) {
        if (push_c) {
            float *r = pool + lane;
            r[0] = ps.o.x; r[V5_SLOTS] = ps.o.y; r[2u * V5_SLOTS] = ps.o.z; r[3u * V5_SLOTS] = ps.d.x; r[4u * V5_SLOTS] = ps.d.y; r[5u * V5_SLOTS] = ps.d.z;
            r[6u * V5_SLOTS] = ps.tmin; r[7u * V5_SLOTS] = ps.tmax;
            status[lane] = RS_BUSY;
        }
        if (push_s) {
            float *r = pool + 64u + lane;
            r[0] = sr.o.x; r[V5_SLOTS] = sr.o.y; r[2u * V5_SLOTS] = sr.o.z; r[3u * V5_SLOTS] = sr.d.x; r[4u * V5_SLOTS] = sr.d.y; r[5u * V5_SLOTS] = sr.d.z;
            r[6u * V5_SLOTS] = sr.tmin; r[7u * V5_SLOTS] = sr.tmax;
            status[64u + lane] = RS_BUSY;
        }
        // ray compaction: the lanes that issued a ray append its slot to the FIFO (ballot + prefix count)
        const unsigned long long mc = __ballot(push_c);
        if (push_c) ring[(q_head + q_count + prefix(mc)) & (V5_SLOTS - 1u)] = (unsigned char) lane;
        q_count += (uint32_t) __popcll(mc);
        const unsigned long long ms = __ballot(push_s);
        if (push_s) ring[(q_head + q_count + prefix(ms)) & (V5_SLOTS - 1u)] = (unsigned char) (64u + lane);
        q_count += (uint32_t) __popcll(ms);
    };

    for (;;) {
        {
            const unsigned long long pm = __ballot(ps.phase == PH_DONE), rm = __ballot(ps.phase != PH_DONE && ps.phase != PH_IDLE);
            if (!pm && !rm) break;
        }
        if (STAMPS) n_outer++;
        auto do_step = [&]() {
            const unsigned long long t0 = STAMP5();
            // ---------------------------------------------------------------- step: chains whose ray results are in
            {
                SECTION_PARAMS(Ps);
                const int st_c = status[lane], st_s = status[64u + lane];
                const bool ready = (ps.phase == PH_CLOSEST && st_c == RS_DONE && st_s != RS_BUSY) || (ps.phase == PH_FLUSH && st_s != RS_BUSY);
                bool push_c = false, push_s = false;
                if (STAMPS) n_stepping += (unsigned long long) __popcll(__ballot(ready));
                ShadowRay sr;
                sr.o = ps.o; sr.d = ps.d; sr.tmin = 0.f; sr.tmax = 0.f; sr.valid = false;
                if (ready) {
                    Hit h{-1, 0.f, 0.f, 0.f};
                    if (ps.phase == PH_CLOSEST) {
                        h.prim = __float_as_int(pool[lane]); h.t = pool[V5_SLOTS + lane]; h.u = pool[2u * V5_SLOTS + lane]; h.v = pool[3u * V5_SLOTS + lane];
                    }
                    const bool shadow_clear = st_s == RS_DONE ? pool[64u + lane] == 0.f : true;
                    status[lane] = RS_IDLE; status[64u + lane] = RS_IDLE;
                    if (LDS_TABLES) path_step<true, FEAT, SamplerT, LdsTables, false>(Ps, LT, ps, smp, h, shadow_clear, sr);
                    else { HT.sh = Ps.shade; HT.bs = Ps.bsdfs; HT.em = Ps.emitters; path_step<true, FEAT, SamplerT, HybridTables, false>(Ps, HT, ps, smp, h, shadow_clear, sr); }
                    push_c = ps.phase == PH_CLOSEST;
                    push_s = sr.valid;
                }
                push_rays(push_c, push_s, sr);
            }
            t_step += STAMP5() - t0;
        };
        auto do_bookkeeping = [&]() {
            const unsigned long long t0 = STAMP5();
            // ---------------------------------------------------------------- bookkeeping: decide, commit, proposals, start
            const bool parked = ps.phase == PH_DONE;
            const unsigned long long pmask = __ballot(parked);
            // (the step above has refilled the queue: "nothing in flight" means every chain that is not idle is parked)
            const bool rays_in_flight = q_count != 0u || __ballot(T.active) != 0ull;
            if (pmask && (__popcll(pmask) >= batch || !rays_in_flight)) {
                SECTION_PARAMS(Pm);
                if (STAMPS) { n_mh++; n_parked += (unsigned long long) __popcll(pmask); }
                if (QCAP != 0u && qn + 64u > QCAP) v4_flush(Pm, Lq, qn, lane);
                int commit = 0, kind = 0; // kind: 0 nothing / finished, 1 next mutation, 2 second stage (4: between mutations, resolved below)
                bool want0 = false, want1 = false, want2 = false;
                float e0x = 0.f, e0y = 0.f, e0r = 0.f, e0g = 0.f, e0b = 0.f;
                float e1x = 0.f, e1y = 0.f, e1r = 0.f, e1g = 0.f, e1b = 0.f;
                float e2x = 0.f, e2y = 0.f, e2r = 0.f, e2g = 0.f, e2b = 0.f;
                if (parked) {
                    if (Pm.type == 1) { // Mira: what its ratio reads (PoolRowSampler)
                        smp.mira_x = Pm.x + cc; smp.mira_stride = Pm.n_chains; smp.mira_k0 = Pm.key0; smp.mira_k1 = Pm.key1;
                        smp.mira_major = base + cs.it; smp.mira_chain = Pm.chain_offset + cc; smp.reset_caches();
                    }
                    const MhOutcome o = mh_decide(Pm, cs, smp, ps, ct);
                    if (o.decided) {
                        cum += o.w.w0;
                        const bool a1st = o.commit == SM_STAGE1, a2nd = o.commit == SM_STAGE2;
                        want1 = !a1st && o.w.w1 > 0.f;
                        e1x = cs.y.px; e1y = cs.y.py; e1r = cs.y.r * o.w.w1; e1g = cs.y.g * o.w.w1; e1b = cs.y.b * o.w.w1;
                        want2 = !a2nd && o.w.w2 > 0.f;
                        e2x = cs.z.px; e2y = cs.z.py; e2r = cs.z.r * o.w.w2; e2g = cs.z.g * o.w.w2; e2b = cs.z.b * o.w.w2;
                        if (o.commit) {
                            want0 = cum > 0.f;
                            e0x = cs.cur.px; e0y = cs.cur.py; e0r = cs.cur.r * cum; e0g = cs.cur.g * cum; e0b = cs.cur.b * cum;
                            cum = a1st ? o.w.w1 : o.w.w2;
                            cs.cur = select_splat(a1st, cs.y, cs.z);
                            if (o.amap) { // acceptance map: the mark goes to the pixel of the state that was LEFT (device_mh.h)
                                const f3 mc = mh_amap_colour(o.amap);
                                want1 = true; e1x = e0x; e1y = e0y; e1r = mc.x; e1g = mc.y; e1b = mc.z;
                            }
                        }
                        commit = o.commit;
                    }
                    kind = cs.stage < 0 ? 4 : (cs.stage == 1 ? 2 : 3); // 3: Green's reverse move
                }
                {
                    // between mutations: go on while short of the target; beyond it (run-ahead) while anybody in the grid is short
                    const uint32_t done_now = base + cs.it;
                    const bool under = __ballot(live && done_now < target) != 0ull;
                    bool more = under;
                    if (Pm.chain_done) {
                        if (!under && !reported) { reported = true; if (lane == 0) atomicSub(Pm.waves_left, 1u); }
                        if (!under) more = __builtin_amdgcn_readfirstlane((int) __hip_atomic_load(Pm.waves_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0;
                    }
                    if (kind == 4) kind = (done_now < target || (done_now < limit && more)) ? 1 : 0;
                }
                if constexpr (QCAP != 0u) { // the queue holds QCAP entries, a round adds at most 64
                    v4_enqueue(Lq, qn, want0, e0x, e0y, e0r, e0g, e0b);
                    if (qn + 64u > QCAP) v4_flush(Pm, Lq, qn, lane);
                    v4_enqueue(Lq, qn, want1, e1x, e1y, e1r, e1g, e1b);
                    if (qn + 64u > QCAP) v4_flush(Pm, Lq, qn, lane);
                    v4_enqueue(Lq, qn, want2, e2x, e2y, e2r, e2g, e2b);
                } else { // no queue in this build: ImageBlock::put straight away (film_put applies the validity test)
                    if (want0) film_put(Pm, e0x, e0y, mk3(e0r, e0g, e0b));
                    if (want1) film_put(Pm, e1x, e1y, mk3(e1r, e1g, e1b));
                    if (want2) film_put(Pm, e2x, e2y, mk3(e2r, e2g, e2b));
                }

                // ---- commit (DRMLTSampler::accept: uCurrent = wrap(adopted proposal)) to the state's home in device memory,
                // flattened: items (accepted chain j, row quad q), chain-minor
                // Under Green an adopted SECOND stage finds the rows holding the reverse move y*, not z: those chains' commits
                // recompute z from the state and the stream (v5_iid_second_again); every other adoption copies the rows.
                const bool green = Pm.type == 0 && !Pm.use_mixture;
                const unsigned long long cmask = __ballot(commit != 0 && !(green && commit == SM_STAGE2));
                if (cmask) {
                    if (commit != 0 && !(green && commit == SM_STAGE2)) lds_list[prefix(cmask)] = (int) lane;
                    const uint32_t n = (uint32_t) __popcll(cmask), total = n * nb1;
                    const float rcp_n = 1.f / (float) n;
                    for (uint32_t ib = 0u; ib < total; ib += 64u) {
                        const uint32_t i = ib + lane;
                        const bool valid = i < total;
                        const uint32_t ii = valid ? i : 0u;
                        const uint32_t q = (uint32_t) (((float) ii + 0.5f) * rcp_n), j = ii - q * n;
                        const uint32_t cj = (uint32_t) lds_list[j];
                        if (valid) {
                            float *dst = Pm.x + (size_t) (4u * q) * Pm.n_chains + wave_base + cj;
                            float v[4]; // (loads first: see v5_fill_second)
#pragma unroll
                            for (uint32_t r = 0; r < 4u; ++r) v[r] = rows.get(4u * q + r < D ? 4u * q + r : 4u * q, cj);
#pragma unroll
                            for (uint32_t r = 0; r < 4u; ++r)
                                if (4u * q + r < D) dst[(size_t) r * Pm.n_chains] = wrap01(v[r]);
                        }
                    }
                }
                const uint32_t chain_base = Pm.chain_offset + wave_base;
                const uint32_t maj_done = base + cs.it - 1u; // the mutation just decided (cs.it was advanced by the decision)
                const unsigned large_done = cs.large ? 1u : 0u;
                const unsigned long long gmask = __ballot(green && commit == SM_STAGE2);
                if (gmask) {
                    if (green && commit == SM_STAGE2) lds_list[prefix(gmask)] = (int) lane;
                    const uint32_t nbz = Pm.timid_after_large ? max(nb1, D / 2u) : D / 2u;
                    const uint32_t n = (uint32_t) __popcll(gmask), total = n * nbz;
                    const float rcp_n = 1.f / (float) n;
                    for (uint32_t ib = 0u; ib < total; ib += 64u) {
                        const uint32_t i = ib + lane;
                        const bool valid = i < total;
                        const uint32_t ii = valid ? i : 0u;
                        const uint32_t bq = (uint32_t) (((float) ii + 0.5f) * rcp_n), j = ii - bq * n;
                        const uint32_t cj = (uint32_t) lds_list[j];
                        const uint32_t mj = (uint32_t) __shfl((int) maj_done, (int) cj, 64);
                        const unsigned lg = (unsigned) __shfl((int) large_done, (int) cj, 64);
                        if (valid && bq < (lg ? nb1 : D / 2u)) v5_iid_second_again(Pm, rows, D, cj, (size_t) wave_base + cj, bq, mj, chain_base + cj, lg != 0u, false);
                    }
                }
                // this wave's own stores to the state rows must have landed before the proposals below read them back (same CU: the
                // wait is all a workgroup-scope fence amounts to)
                if (cmask | gmask) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                // ---- start (parked chain lanes): the coins of the mutation that begins were drawn with the previous one
                if (parked && kind == 1) {
                    if (COIN_ROWS) {
                        const float *cn = &lds_x[L.coin_off + lane];
                        cs.large = cn[0] < Pm.p_large;
                        cs.coin_acc1 = cn[64]; cs.coin_acc2 = cn[128]; cs.coin_mix = cn[192];
                    } else {
                        const u4 coins = philox4x32_10(Pm.key0, Pm.key1, 0u, base + cs.it, Pm.chain_offset + cc, TAG_COIN);
                        cs.large = u32_to_unit(coins.x) < Pm.p_large;
                        cs.coin_acc1 = u32_to_unit(coins.y); cs.coin_acc2 = u32_to_unit(coins.z); cs.coin_mix = u32_to_unit(coins.w);
                    }
                    cs.stage = 0;
                    cs.do_second = false;
                    cs.nd1 = cs.nd2 = 0u;
                }
                // ---- proposals, flattened: items (chain j, Philox block b) -> dimensions 4b .. 4b+3 of y from the state in device
                // memory (the commits above are this wave's own stores: visible to its later loads); block nb1 = the four coins of
                // the NEXT mutation
                const uint32_t maj_mine = base + cs.it; // the mutation in flight
                const unsigned info = cs.large ? 1u : 0u;
                const unsigned long long f1mask = __ballot(kind == 1);
                if (f1mask) {
                    if (kind == 1) lds_list[prefix(f1mask)] = (int) lane;
                    const uint32_t n = (uint32_t) __popcll(f1mask), total = n * (nb1 + (COIN_ROWS ? 1u : 0u));
                    const float rcp_n = 1.f / (float) n;
                    // The state components an item perturbs come from device memory (past the L2s at three waves per SIMD): the reads of pass
                    // i + 1 are issued before pass i's arithmetic (Philox, the transition kernel) instead of behind its stores, which they could not
                    // pass for the compiler.
                    auto locate = [&](uint32_t ib, uint32_t &b_, uint32_t &cj_, bool &valid_) {
                        const uint32_t i = ib + lane;
                        valid_ = i < total;
                        const uint32_t ii = valid_ ? i : 0u;
                        b_ = (uint32_t) (((float) ii + 0.5f) * rcp_n);
                        cj_ = (uint32_t) lds_list[ii - b_ * n];
                    };
                    uint32_t b_n, cj_n; bool valid_n;
                    locate(0u, b_n, cj_n, valid_n);
                    State4 X_n = v5_state4(Pm, D, (size_t) wave_base + cj_n, valid_n && b_n < nb1 ? b_n : 0u);
                    for (uint32_t ib = 0u; ib < total; ib += 64u) {
                        const uint32_t b = b_n, cj = cj_n;
                        const bool valid = valid_n;
                        const State4 X = X_n;
                        if (ib + 64u < total) {
                            locate(ib + 64u, b_n, cj_n, valid_n);
                            X_n = v5_state4(Pm, D, (size_t) wave_base + cj_n, valid_n && b_n < nb1 ? b_n : 0u);
                        }
                        const uint32_t mj = (uint32_t) __shfl((int) maj_mine, (int) cj, 64);
                        const unsigned inf = (unsigned) __shfl((int) info, (int) cj, 64);
                        if (valid) {
                            if (b < nb1) v5_fill_first(Pm, rows, D, cj, X, b, mj, chain_base + cj, inf != 0u);
                            else {
                                const u4 coins = philox4x32_10(Pm.key0, Pm.key1, 0u, mj + 1u, chain_base + cj, TAG_COIN);
                                float *dst = &lds_x[L.coin_off + cj];
                                dst[0] = u32_to_unit(coins.x); dst[64] = u32_to_unit(coins.y); dst[128] = u32_to_unit(coins.z); dst[192] = u32_to_unit(coins.w);
                            }
                        }
                    }
                }
                const unsigned long long f2mask = __ballot(kind == 2);
                if (f2mask) { // second-stage proposals (rejected bold steps), in place over the first-stage rows
                    if (kind == 2) lds_list[prefix(f2mask)] = (int) lane;
                    // blocks per chain: uniforms for a large step (one per dim), the orbital angles (one per pair), Gaussian pairs (one block per two dims)
                    const uint32_t nbs = Pm.type == 2 ? (D / 2u + 3u) / 4u : D / 2u;
                    const uint32_t nb2 = Pm.timid_after_large ? max(nb1, nbs) : nbs;
                    const uint32_t n = (uint32_t) __popcll(f2mask), total = n * nb2;
                    const float rcp_n = 1.f / (float) n;
                    for (uint32_t ib = 0u; ib < total; ib += 64u) {
                        const uint32_t i = ib + lane;
                        const bool valid = i < total;
                        const uint32_t ii = valid ? i : 0u;
                        const uint32_t b = (uint32_t) (((float) ii + 0.5f) * rcp_n), j = ii - b * n;
                        const uint32_t cj = (uint32_t) lds_list[j];
                        const uint32_t mj = (uint32_t) __shfl((int) maj_mine, (int) cj, 64);
                        const unsigned inf = (unsigned) __shfl((int) info, (int) cj, 64);
                        if (valid && b < (inf ? nb1 : nbs)) v5_fill_second(Pm, rows, D, cj, (size_t) wave_base + cj, b, mj, chain_base + cj, inf != 0u);
                    }
                }
                const unsigned long long f3mask = __ballot(kind == 3);
                if (f3mask) { // Green's reverse move y* = z - (y - x), over the rows that held z
                    if (kind == 3) lds_list[prefix(f3mask)] = (int) lane;
                    const uint32_t nb3 = Pm.timid_after_large ? max(nb1, D / 2u) : D / 2u;
                    const uint32_t n = (uint32_t) __popcll(f3mask), total = n * nb3;
                    const float rcp_n = 1.f / (float) n;
                    for (uint32_t ib = 0u; ib < total; ib += 64u) {
                        const uint32_t i = ib + lane;
                        const bool valid = i < total;
                        const uint32_t ii = valid ? i : 0u;
                        const uint32_t b = (uint32_t) (((float) ii + 0.5f) * rcp_n), j = ii - b * n;
                        const uint32_t cj = (uint32_t) lds_list[j];
                        const uint32_t mj = (uint32_t) __shfl((int) maj_mine, (int) cj, 64);
                        const unsigned inf = (unsigned) __shfl((int) info, (int) cj, 64);
                        if (valid && b < (inf ? nb1 : D / 2u)) v5_iid_second_again(Pm, rows, D, cj, (size_t) wave_base + cj, b, mj, chain_base + cj, inf != 0u, true);
                    }
                }
                // (rows in device memory: the flattened stores above land before their chains' lanes read them -- same CU, as for the state)
                if (ROWS_MEM && (f1mask | f2mask | f3mask)) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                // ---- begin the evaluation: film position and camera ray from the first two components; the ray goes into the pool
                if (parked) {
                    if (kind == 0) ps.phase = PH_IDLE;
                    else {
                        path_init(Pm, ps);
                        const float v0 = smp.next(0u), v1 = smp.next(1u);
                        path_begin(Pm, ps, v0, v1);
                    }
                }
                {
                    ShadowRay none;
                    none.o = ps.o; none.d = ps.d; none.tmin = 0.f; none.tmax = 0.f; none.valid = false;
                    push_rays(parked && ps.phase == PH_CLOSEST, false, none); // the camera rays of the evaluations that begin
                }
            }
            t_mh += STAMP5() - t0;
        };
        auto do_trace = [&]() {
            const unsigned long long t0 = STAMP5();
            // ---------------------------------------------------------------- trace: the pool's rays, any lane any ray
            if constexpr (FLAT) {
                // flat scenes: ONE pass of the brute-force loop over up to 64 pending rays; all of them are done when it returns
                if (q_count != 0u) {
                    SECTION_PARAMS(Pt);
                    if (STAMPS) { n_phase++; n_lanes_at_start += (unsigned long long) min(q_count, 64u); }
                    const bool take = lane < q_count;
                    slot = ring[(q_head + lane) & (V5_SLOTS - 1u)];
                    if (take) {
                        const float *r = pool + slot;
                        const Hit h = trace<FEAT>(Pt, mk3(r[0], r[V5_SLOTS], r[2u * V5_SLOTS]), mk3(r[3u * V5_SLOTS], r[4u * V5_SLOTS], r[5u * V5_SLOTS]),
                                                  r[6u * V5_SLOTS], r[7u * V5_SLOTS], slot >= 64u);
                        if (slot < 64u) {
                            float *w = pool + slot;
                            w[0] = __int_as_float(h.prim); w[V5_SLOTS] = h.t; w[2u * V5_SLOTS] = h.u; w[3u * V5_SLOTS] = h.v;
                        } else {
                            pool[slot] = h.prim >= 0 ? 1.f : 0.f;
                        }
                        status[slot] = RS_DONE;
                    }
                    const uint32_t taken = min(q_count, 64u);
                    q_head = (q_head + taken) & (V5_SLOTS - 1u);
                    q_count -= taken;
                }
            } else {
                SECTION_PARAMS(Pt);
                const TravLoop<StackT, DParams, OVF || !STACK16, CAP, FEAT> TL(Pt, my_stack);
                const int yield_lanes = Pt.trace_yield, refill_at = Pt.pool_refill;
                int finished_closest = 0;
                bool first = true;
                if (STAMPS) n_phase++;
                for (;;) {
                    // idle lanes take pending slots off the queue (at the start of a phase, and whenever a few lanes have run dry)
                    const unsigned long long idle = __ballot(!T.active);
                    if (q_count != 0u && idle != 0ull && (first || __popcll(idle) >= refill_at || idle == ~0ull)) {
                        const uint32_t rank = prefix(idle);
                        const bool take = !T.active && rank < q_count;
                        if (take) {
                            slot = ring[(q_head + rank) & (V5_SLOTS - 1u)];
                            const float *r = pool + slot;
                            trav_begin(T, mk3(r[0], r[V5_SLOTS], r[2u * V5_SLOTS]), mk3(r[3u * V5_SLOTS], r[4u * V5_SLOTS], r[5u * V5_SLOTS]),
                                       r[6u * V5_SLOTS], r[7u * V5_SLOTS], slot >= 64u);
                        }
                        const uint32_t taken = min((uint32_t) __popcll(idle), q_count);
                        q_head = (q_head + taken) & (V5_SLOTS - 1u);
                        q_count -= taken;
                        if (STAMPS) n_refill++;
                    }
                    if (STAMPS && first) n_lanes_at_start += (unsigned long long) __popcll(__ballot(T.active));
                    first = false;
                    if (finished_closest >= yield_lanes) break;
                    bool any;
                    const bool done_now = TL.step(T, true, any);
                    if (!any) break; // (the queue is empty too: an idle wave with pending slots refills above)
                    if (done_now) { // result over the ray's record; the owner chain picks it up in the step phase
                        if (slot < 64u) {
                            float *r = pool + slot;
                            r[0] = __int_as_float(T.h.prim); r[V5_SLOTS] = T.h.t; r[2u * V5_SLOTS] = T.h.u; r[3u * V5_SLOTS] = T.h.v;
                        } else {
                            pool[slot] = T.h.prim >= 0 ? 1.f : 0.f;
                        }
                        status[slot] = RS_DONE;
                    }
                    finished_closest += __popcll(__ballot(done_now && slot < 64u));
                }
        
            }
            t_trace += STAMP5() - t0;
        };
        // Flat scenes: every ray of a pass is done when the pass returns, so the chains step first and the bookkeeping branch
        // sees who parked and what is still queued (it then fires for a full batch, or when nothing else is left to do). BVH
        // scenes: a phase leaves traversals running; the branch comes first and its camera rays join the phase that follows.
        if constexpr (FLAT) { do_step(); do_bookkeeping(); do_trace(); }
        else { do_bookkeeping(); do_step(); do_trace(); }
    }
#undef STAMP5
    SECTION_PARAMS(Pe);
#undef SECTION_PARAMS
    if (STAMPS && lane == 0) {
        atomicAdd(Pe.stats + 16, t_mh); atomicAdd(Pe.stats + 17, t_trace); atomicAdd(Pe.stats + 18, t_step); atomicAdd(Pe.stats + 19, n_outer);
        atomicAdd(Pe.stats + 23, n_mh); atomicAdd(Pe.stats + 24, n_parked); atomicAdd(Pe.stats + 25, n_stepping);
    }
    // "Perform the last splat": the current states with what they have accumulated since they were adopted
    if constexpr (QCAP != 0u) {
        if (qn + 64u > QCAP) v4_flush(Pe, Lq, qn, lane);
        v4_enqueue(Lq, qn, live && cum > 0.f, cs.cur.px, cs.cur.py, cs.cur.r * cum, cs.cur.g * cum, cs.cur.b * cum);
        v4_flush(Pe, Lq, qn, lane);
    } else if (live && cum > 0.f) film_put(Pe, cs.cur.px, cs.cur.py, mk3(cs.cur.r * cum, cs.cur.g * cum, cs.cur.b * cum));
    if (live) { // (the PSS state is at home in device memory already)
        Pe.cur_lum[c] = cs.cur.lum; Pe.cur_px[c] = cs.cur.px; Pe.cur_py[c] = cs.cur.py;
        Pe.cur_r[c] = cs.cur.r; Pe.cur_g[c] = cs.cur.g; Pe.cur_b[c] = cs.cur.b;
        if (Pe.chain_done) Pe.chain_done[c] = base + cs.it;
    }
    if (Pe.chain_done && !reported && lane == 0) atomicSub(Pe.waves_left, 1u); // (a wave none of whose chains had anything to do)
    flush_counters(Pe, ct, lane);
    const unsigned long long decided = wave_sum(live ? cs.it : 0u);
    const unsigned long long nn = FLAT ? 0ull : wave_sum(T.n_nodes), np = FLAT ? 0ull : wave_sum(T.n_prims);
    if (lane == 0) {
        atomicAdd(Pe.stats + 9, decided);
        atomicAdd(Pe.stats + 10, nn); atomicAdd(Pe.stats + 11, np); atomicAdd(Pe.stats + 12, (unsigned long long) T.it_inner); atomicAdd(Pe.stats + 13, (unsigned long long) T.it_leaf);
        if (STAMPS) { atomicAdd(Pe.stats + 20, n_phase); atomicAdd(Pe.stats + 21, n_lanes_at_start); atomicAdd(Pe.stats + 22, n_refill); }
    }
}

__global__ void __launch_bounds__(64) k_eval_paths(DParams P, const float *u, uint32_t n, uint32_t dim, float *out8) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Sampler smp;
    smp.key0 = smp.key1 = smp.chain = smp.major = 0u;
    smp.mode = SM_ARRAY; smp.type = P.type; smp.large = false; smp.sigma2 = P.sigma2; smp.lane = 0u;
    smp.arr = u + (size_t) i * dim;
    uint32_t nr, nd;
    DSplat s = eval_path(P, smp, nr, nd);
    float *o = out8 + (size_t) i * 8;
    o[0] = s.lum; o[1] = s.px; o[2] = s.py; o[3] = s.r; o[4] = s.g; o[5] = s.b;
    o[6] = __int_as_float((int) nd); o[7] = __int_as_float((int) nr);
}

__global__ void __launch_bounds__(64) k_render_pt(DParams P, uint64_t n_samples, uint32_t stream, float scale) {
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (; i < n_samples; i += stride) {
        Sampler smp;
        smp.key0 = P.key0; smp.key1 = P.key1; smp.chain = stream + (uint32_t) (i >> 32); smp.major = (uint32_t) i;
        smp.mode = SM_PT; smp.type = P.type; smp.large = false; smp.sigma2 = P.sigma2; smp.lane = 0u; smp.arr = nullptr;
        uint32_t nr, nd;
        DSplat s = eval_path(P, smp, nr, nd);
        if (s.lum > 0.f) film_put(P, s.px, s.py, mk3(s.r * scale, s.g * scale, s.b * scale));
    }
}

// sum of pixel luminances in double (one atomic per block)
__global__ void __launch_bounds__(256) k_lum_sum(const float *film, const float *importance, uint32_t n_pixels, double *sum) {
    __shared__ double part[256];
    double acc = 0.0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += gridDim.x * blockDim.x)
        acc += ((double) film[3 * i] * 0.212671 + (double) film[3 * i + 1] * 0.715160 + (double) film[3 * i + 2] * 0.072169) *
               (importance ? (double) importance[i] : 1.0); // drmlt_proc.cpp:826-832
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int) threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(sum, part[0]);
}

__global__ void __launch_bounds__(256) k_develop(const float *film, const float *direct, const float *importance, float factor, uint32_t n,
                                                 float *out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = film[i] * (importance ? factor * importance[i / 3u] : factor) + (direct ? direct[i] : 0.f); // :841-847
}

__global__ void k_set_u32(uint32_t *p, uint32_t v) { *p = v; }
void launch_set_u32(uint32_t *p, uint32_t v, hipStream_t st) { hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, p, v); }
__global__ void k_set2(double *p, double a, double b) { p[0] = a; p[1] = b; }
void launch_set2(double *p, double a, double b, hipStream_t st) { hipLaunchKernelGGL(k_set2, dim3(1), dim3(1), 0, st, p, a, b); }

// Grids of the film kernels are capped (grid-stride loops): a launch of MORE than 2048 workgroups in front of a chain kernel
// costs that kernel 14 % (k_mutate_v4 41.0 -> 47.0 ms after a 3072-block develop, whatever the develop reads or writes;
// 2048 blocks and fewer: no effect) -- the chain kernel's 2048 workgroups fill the device exactly, eight to a CU, two to a
// SIMD, and whatever the dispatcher still holds of the larger grid skews that placement.
#define AUX_GRID_CAP 1024u
// the same with the factor taken from device memory: scal = {sum of the film's luminance over all ranks, sum of the ranks' b}
// (drmlt_exchange_tiled without a host round trip: the exchange of a step is enqueued behind its chain kernel)
__global__ void __launch_bounds__(256) k_develop_dev(const float *film, const float *importance, const double *scal, float inv_world, float inv_pixels,
                                                     int acceptance_map, uint32_t n, float *out) {
    const float factor = acceptance_map ? 1.f : (float) ((scal[1] * (double) inv_world) / (scal[0] * (double) inv_pixels));
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = film[i] * (importance ? factor * importance[i / 3u] : factor);
}

// dst += src (film tiles of ranks that share a device: drmlt_node.cpp's loopback transport)
__global__ void __launch_bounds__(256) k_accumulate(float *dst, const float *src, size_t n) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) dst[i] += src[i];
}
void launch_accumulate(float *dst, const float *src, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_accumulate, dim3((unsigned) std::min<size_t>((n + 255) / 256, AUX_GRID_CAP)), dim3(256), 0, st, dst, src, n);
}

// ---- host-callable launchers (C++ linkage, used by drmlt_capi.cpp) --------------------------
void launch_bootstrap(const DParams &P, uint32_t n, float *lum_out, hipStream_t st) {
    hipLaunchKernelGGL(k_bootstrap, dim3((n + 63) / 64), dim3(64), 0, st, P, n, lum_out);
}
void launch_init_chains(const DParams &P, const uint32_t *seed_index, const float *seed_lum, hipStream_t st) {
    hipLaunchKernelGGL(k_init_chains, dim3((P.n_chains + 63) / 64), dim3(64), 0, st, P, seed_index, seed_lum);
}
void launch_mutate_pssmlt(const DParams &P, uint32_t n_mut, uint32_t mut_base, hipStream_t st) {
    hipLaunchKernelGGL(k_mutate_pssmlt, dim3((P.n_chains + CHAIN_BLOCK - 1) / CHAIN_BLOCK), dim3(CHAIN_BLOCK), (size_t) P.eff_dim * 64 * sizeof(float), st, P,
                       n_mut, mut_base);
}
void launch_mutate(const DParams &P, uint32_t n_mut, uint32_t mut_base, hipStream_t st) {
    const size_t D = (size_t) P.eff_dim, D4 = (D + 3) & ~(size_t) 3;
    const dim3 block(CHAIN_BLOCK);
    if (P.kernel_variant == 5) { // ray pool, 64 chains per wave
        const bool flat = (P.features & 8) == 0;
        const bool rows_mem = P.rows != nullptr && (!flat || P.tables_in_lds); // (drmlt_capi.cpp: chains for more than two waves per SIMD)
        size_t lds = v5_lds_bytes(rows_mem ? 0u : (uint32_t) D, (flat || P.bvh_stack16) ? V5_QCAP : V5_QCAP_STACK32, flat || P.bvh_stack16);
        if (flat && P.tables_in_lds) lds += (size_t) P.n_shade * 64 + (size_t) P.n_bsdfs * 48 + (size_t) P.n_emitters * 32;
        if (!flat && P.small_tables_lds) lds += ((size_t) P.n_bsdfs * 12 + (size_t) P.n_emitters * 24) * sizeof(float);
        if (getenv("DRMLT_VERBOSE")) fprintf(stderr, "[drmlt] k_mutate_v5: %zu B of LDS per wave%s%s\n", lds, flat ? "" : " (+ the traversal stack)", rows_mem ? "; proposal rows in device memory, three waves per SIMD" : "");
        const dim3 g5((P.n_chains + 63) / 64);
        const bool diffuse = P.features == 8;
        if (flat && rows_mem) { // three waves per SIMD, as on traversed scenes
            if (P.features == 0) hipLaunchKernelGGL((k_mutate_v5<0, true, false, false, true, true>), g5, block, lds, st, P, n_mut, mut_base);
            else if (P.features == 1) hipLaunchKernelGGL((k_mutate_v5<1, true, false, false, true, true>), g5, block, lds, st, P, n_mut, mut_base);
            else if ((P.features & ~3) == 0) hipLaunchKernelGGL((k_mutate_v5<3, true, false, false, true, true>), g5, block, lds, st, P, n_mut, mut_base);
            else hipLaunchKernelGGL((k_mutate_v5<7, true, false, false, true, true>), g5, block, lds, st, P, n_mut, mut_base);
        }
        else if (flat) { // brute-force loop as the trace phase; tables in LDS when they are small (they are, for scenes this small)
            if (!P.tables_in_lds) hipLaunchKernelGGL((k_mutate_v5<7, true, false, false, false>), g5, block, lds, st, P, n_mut, mut_base);
            else if (P.features == 0 && (P.debug & 128)) hipLaunchKernelGGL((k_mutate_v5<0, true, false, true, true>), g5, block, lds, st, P, n_mut, mut_base); // diagnostic stamps
            else if (P.features == 0) hipLaunchKernelGGL((k_mutate_v5<0, true, false, false, true>), g5, block, lds, st, P, n_mut, mut_base);
            else if (P.features == 1) hipLaunchKernelGGL((k_mutate_v5<1, true, false, false, true>), g5, block, lds, st, P, n_mut, mut_base); // rough conductors, no dielectric (config 3)
            else if ((P.features & ~3) == 0) hipLaunchKernelGGL((k_mutate_v5<3, true, false, false, true>), g5, block, lds, st, P, n_mut, mut_base);
            else hipLaunchKernelGGL((k_mutate_v5<7, true, false, false, true>), g5, block, lds, st, P, n_mut, mut_base);
        }
        else if (rows_mem) {
            if (!P.bvh_stack16) { if (diffuse) hipLaunchKernelGGL((k_mutate_v5<8, false, true, false, false, true>), g5, block, lds, st, P, n_mut, mut_base); else hipLaunchKernelGGL((k_mutate_v5<15, false, true, false, false, true>), g5, block, lds, st, P, n_mut, mut_base); }
            else if (P.bvh_overflow) { if (diffuse) hipLaunchKernelGGL((k_mutate_v5<8, true, true, false, false, true>), g5, block, lds, st, P, n_mut, mut_base); else hipLaunchKernelGGL((k_mutate_v5<15, true, true, false, false, true>), g5, block, lds, st, P, n_mut, mut_base); }
            else if (diffuse) hipLaunchKernelGGL((k_mutate_v5<8, true, false, false, false, true>), g5, block, lds, st, P, n_mut, mut_base);
            else hipLaunchKernelGGL((k_mutate_v5<15, true, false, false, false, true>), g5, block, lds, st, P, n_mut, mut_base);
        }
        else if (!P.bvh_stack16) { if (diffuse) hipLaunchKernelGGL((k_mutate_v5<8, false, true>), g5, block, lds, st, P, n_mut, mut_base); else hipLaunchKernelGGL((k_mutate_v5<15, false, true>), g5, block, lds, st, P, n_mut, mut_base); }
        else if (P.bvh_overflow) { if (diffuse) hipLaunchKernelGGL((k_mutate_v5<8, true, true>), g5, block, lds, st, P, n_mut, mut_base); else hipLaunchKernelGGL((k_mutate_v5<15, true, true>), g5, block, lds, st, P, n_mut, mut_base); }
        else if (diffuse && (P.debug & 128)) hipLaunchKernelGGL((k_mutate_v5<8, true, false, true>), g5, block, lds, st, P, n_mut, mut_base); // diagnostic stamps
        else if (diffuse) hipLaunchKernelGGL((k_mutate_v5<8, true, false>), g5, block, lds, st, P, n_mut, mut_base);
        else hipLaunchKernelGGL((k_mutate_v5<15, true, false>), g5, block, lds, st, P, n_mut, mut_base);
    } else if (P.kernel_variant == 4) { // free-running chains, flattened bookkeeping, queued splats (rows of 33 floats)
        const size_t qcap = (P.features & 8) || !P.tables_in_lds ? V4_QCAP_BVH : V4_QCAP; // as the kernel variants below
        size_t lds = ((D + 2 * D4 + 4) * V4_STRIDE + 32 + 5 * qcap + 3) / 4 * 4 * sizeof(float);
        if (P.tables_in_lds) lds += (size_t) P.n_shade * 64 + (size_t) P.n_bsdfs * 48 + (size_t) P.n_emitters * 32;
        if (getenv("DRMLT_VERBOSE")) fprintf(stderr, "[drmlt] k_mutate_v4: %zu B of LDS per wave\n", lds);
        const dim3 g4((P.n_chains + 31) / 32);
        if (P.tables_in_lds) {
            if (P.features == 0 && (P.debug & 128)) hipLaunchKernelGGL((k_mutate_v4<0, true, true>), g4, block, lds, st, P, n_mut, mut_base);
            else if (P.features == 0) hipLaunchKernelGGL((k_mutate_v4<0, true, false>), g4, block, lds, st, P, n_mut, mut_base);
            else if ((P.features & ~3) == 0 && (P.debug & 128)) hipLaunchKernelGGL((k_mutate_v4<3, true, true>), g4, block, lds, st, P, n_mut, mut_base); // diagnostic stamps
            else if ((P.features & ~3) == 0) hipLaunchKernelGGL((k_mutate_v4<3, true, false>), g4, block, lds, st, P, n_mut, mut_base);
            else if ((P.features & 8) == 0) hipLaunchKernelGGL((k_mutate_v4<7, true, false>), g4, block, lds, st, P, n_mut, mut_base);
            // BVH: 32-bit stacks always run the build with the spill / refill paths (short LDS column), 16-bit stacks only for
            // trees deeper than their column
            else if (!P.bvh_stack16) hipLaunchKernelGGL((k_mutate_v4<15, true, false, false, true>), g4, block, lds, st, P, n_mut, mut_base);
            else if (P.bvh_overflow) hipLaunchKernelGGL((k_mutate_v4<15, true, false, true, true>), g4, block, lds, st, P, n_mut, mut_base);
            else hipLaunchKernelGGL((k_mutate_v4<15, true, false, true>), g4, block, lds, st, P, n_mut, mut_base);
        } else if (!P.bvh_stack16 && P.features == 8) hipLaunchKernelGGL((k_mutate_v4<8, false, false, false, true>), g4, block, lds, st, P, n_mut, mut_base);
        else if (!P.bvh_stack16) hipLaunchKernelGGL((k_mutate_v4<15, false, false, false, true>), g4, block, lds, st, P, n_mut, mut_base);
        else if (P.bvh_overflow) hipLaunchKernelGGL((k_mutate_v4<15, false, false, true, true>), g4, block, lds, st, P, n_mut, mut_base);
        else if (P.debug & 128) hipLaunchKernelGGL((k_mutate_v4<15, false, true, true>), g4, block, lds, st, P, n_mut, mut_base); // diagnostic stamps
        else if (P.features == 8) hipLaunchKernelGGL((k_mutate_v4<8, false, false, true>), g4, block, lds, st, P, n_mut, mut_base); // triangle meshes with diffuse surfaces only
        else hipLaunchKernelGGL((k_mutate_v4<15, false, false, true>), g4, block, lds, st, P, n_mut, mut_base);
    } else { // k_mutate_v3, the cross-check: 32 chains per wave, rows of 32 floats
        size_t lds = (D + 2 * D4) * 32 * sizeof(float);
        if (P.tables_in_lds) lds += (size_t) P.n_shade * 64 + (size_t) P.n_bsdfs * 48 + (size_t) P.n_emitters * 32;
        if (getenv("DRMLT_VERBOSE")) fprintf(stderr, "[drmlt] k_mutate_v3: %zu B of LDS per wave\n", lds);
        // specialisations: 0 = diffuse polygons (Cornell configs); 3 = + rough conductor / dielectric, still flat primitives
        // under the brute-force loop (door config); 7 = + spheres; 15 = everything (BVH traversal, with its 6 KB LDS stack)
        const dim3 g3((P.n_chains + 31) / 32);
        if (P.tables_in_lds) {
            if (P.features == 0) hipLaunchKernelGGL((k_mutate_v3<0, true>), g3, block, lds, st, P, n_mut, mut_base);
            else if ((P.features & ~3) == 0) hipLaunchKernelGGL((k_mutate_v3<3, true>), g3, block, lds, st, P, n_mut, mut_base);
            else if ((P.features & 8) == 0) hipLaunchKernelGGL((k_mutate_v3<7, true>), g3, block, lds, st, P, n_mut, mut_base);
            else hipLaunchKernelGGL((k_mutate_v3<15, true>), g3, block, lds, st, P, n_mut, mut_base);
        } else { // large scenes (BVH, tables in HBM/L2): one general variant
            hipLaunchKernelGGL((k_mutate_v3<15, false>), g3, block, lds, st, P, n_mut, mut_base);
        }
    }
}
void launch_eval_paths(const DParams &P, const float *u, uint32_t n, uint32_t dim, float *out8, hipStream_t st) {
    hipLaunchKernelGGL(k_eval_paths, dim3((n + 63) / 64), dim3(64), 0, st, P, u, n, dim, out8);
}
void launch_render_pt(const DParams &P, uint64_t n_samples, uint32_t stream, float scale, hipStream_t st) {
    hipLaunchKernelGGL(k_render_pt, dim3(16384), dim3(64), 0, st, P, n_samples, stream, scale);
}
void launch_lum_sum(const float *film, const float *importance, uint32_t n_pixels, double *sum, hipStream_t st) {
    hipLaunchKernelGGL(k_lum_sum, dim3(256), dim3(256), 0, st, film, importance, n_pixels, sum);
}
void launch_develop_dev(const float *film, const float *importance, const double *scal, float inv_world, float inv_pixels, int acceptance_map, uint32_t n,
                        float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_develop_dev, dim3(std::min((n + 255) / 256, AUX_GRID_CAP)), dim3(256), 0, st, film, importance, scal, inv_world, inv_pixels, acceptance_map, n, out);
}
void launch_develop(const float *film, const float *direct, const float *importance, float factor, uint32_t n, float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_develop, dim3(std::min((n + 255) / 256, AUX_GRID_CAP)), dim3(256), 0, st, film, direct, importance, factor, n, out);
}
