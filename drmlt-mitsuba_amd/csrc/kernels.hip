// HIP kernels of the DRMLT hot path for gfx950 (MI355X). One Markov chain per lane.
//
//   k_bootstrap     luminance samples of PathSampler::generateSeeds (pathsampler.cpp:879-920)
//   k_init_chains   seed replay + fillReplay + luminance sanity check (drmlt_proc.cpp:467-514)
//   k_mutate        DRMLTRenderer::process / processMixture chain loop (drmlt_proc.cpp:161-380,518-770)
//   k_eval_paths    PathSampler::sampleSplats on caller-supplied PSS points (pathsampler.cpp:529-567)
//   k_render_pt     independent samples of the same integrand (validation image)
//   k_lum_sum / k_develop   DRMLTProcess::develop (drmlt_proc.cpp:824-849)
#include <cstdlib>
#include <cstdio>
#include "device_path.h"

#include "kernel_common.h"

__global__ void __launch_bounds__(64) k_bootstrap(DParams P, uint32_t n, float *lum_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Sampler smp;
    smp.key0 = P.key0; smp.key1 = P.key1; smp.chain = P.boot_stream; smp.major = i;
    smp.mode = SM_BOOT; smp.type = P.type; smp.large = false; smp.sigma2 = P.sigma2; smp.lane = 0u; smp.arr = nullptr;
    uint32_t nr, nd;
    DSplat s = eval_path(P, smp, nr, nd);
    lum_out[i] = s.lum;
}

__global__ void __launch_bounds__(64) k_init_chains(DParams P, const uint32_t *seed_index, const float *seed_lum) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.n_chains) return;
    Sampler smp;
    smp.key0 = P.key0; smp.key1 = P.key1; smp.chain = P.boot_stream; smp.major = seed_index[c];
    smp.mode = SM_BOOT; smp.type = P.type; smp.large = false; smp.sigma2 = P.sigma2; smp.lane = 0u; smp.arr = nullptr;
    uint32_t nr, nd;
    DSplat s = eval_path(P, smp, nr, nd);
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

__global__ void __launch_bounds__(CHAIN_BLOCK) k_mutate(DParams P, uint32_t n_mut, uint32_t mut_base) {
    const uint32_t lane = threadIdx.x;
    const uint32_t c = blockIdx.x * CHAIN_BLOCK + lane;
    const bool live = c < P.n_chains;
    const uint32_t cc = live ? c : P.n_chains - 1;
    const int D = P.eff_dim;
    for (int k = 0; k < D; ++k) lds_x[k * 64 + lane] = P.x[(size_t) k * P.n_chains + cc];

    DSplat cur;
    cur.lum = P.cur_lum[cc]; cur.px = P.cur_px[cc]; cur.py = P.cur_py[cc];
    cur.r = P.cur_r[cc]; cur.g = P.cur_g[cc]; cur.b = P.cur_b[cc];

    Sampler smp;
    smp.key0 = P.key0; smp.key1 = P.key1; smp.chain = P.chain_offset + cc;
    smp.type = P.type; smp.sigma2 = P.sigma2; smp.lane = lane; smp.arr = nullptr;
    Counters ct = {0u, 0u, 0u, 0u, 0u};
    const bool amap = P.acceptance_map != 0;

    if (live && !(P.debug & 8)) for (uint32_t it = 0; it < n_mut; ++it) {
        const uint32_t m = mut_base + it;
        const u4 coins = philox4x32_10(P.key0, P.key1, 0u, m, smp.chain, TAG_COIN);
        const bool large = u32_to_unit(coins.x) < P.p_large;
        smp.major = m;
        smp.large = large;
        uint32_t nd1 = 0, nd2 = 0;
        DSplat y, z;
        y.lum = 0.f; y.px = y.py = y.r = y.g = y.b = 0.f;
        z = y;
        float a1 = 0.f, a2 = 0.f;
        bool acc1 = false, acc2 = false, doSecond = false;
        const bool mix = P.use_mixture != 0;

        // Stage loop with ONE path-evaluation site: 0 = first stage, 1 = second stage,
        // 2 = Green's reverse move. Lanes leave the loop as soon as their mutation is decided.
#pragma nounroll
        for (int stage = 0; stage < 3; ++stage) {
            smp.mode = stage == 0 ? SM_STAGE1 : (stage == 1 ? SM_STAGE2 : SM_REVERSE);
            uint32_t nr, nd;
            DSplat res = eval_path(P, smp, nr, nd);
            ct.rays += nr;
            normalize_splat(res, P);
            if (stage == 0) {
                y = res; nd1 = nd;
                if (!(mix ? lum_invalid_mix(y.lum) : lum_invalid(y.lum))) { // Eq. 5, drmlt_proc.cpp:544-550 / :285-293
                    a1 = fminf(1.f, y.lum / cur.lum);
                    acc1 = a1 >= 1.f || u32_to_unit(coins.y) < a1;
                }
                if (!mix) doSecond = !acc1 && (P.timid_after_large || !large);   // :553-558
                else doSecond = !large && u32_to_unit(coins.w) < 0.5f;           // :296-299
                if (!doSecond) break;
            } else if (stage == 1) {
                z = res; nd2 = nd;
                if (mix) { // processMixture: the second-stage proposal replaces the first (:313-324)
                    acc1 = false;
                    a1 = 0.f;
                    if (!lum_invalid_mix(z.lum)) {
                        a2 = fminf(1.f, z.lum / cur.lum);
                        acc2 = a2 >= 1.f || u32_to_unit(coins.z) < a2;
                    }
                    break;
                }
                if (lum_invalid(z.lum)) break;
                if (P.type == 0) continue; // Green & Mira (2001): needs the reverse path y* = z - (y - x)
                if (P.type == 1) { // Tierney & Mira (1999), drmlt_proc.cpp:625-650
                    float aRev = fminf(1.f, y.lum / z.lum);
                    if (!(aRev >= 1.f)) {
                        float ratio = 1.f;
                        if (!large) { // Q1(y|z) / Q1(y|x) over the used dimensions (drmlt_sampler.cpp:400-414)
                            uint32_t dimStage = max(nd1, nd2) - 1u;
                            float num = 0.f, den = 0.f;
                            for (uint32_t i = 0; i < dimStage; ++i) {
                                float yi = smp.y_raw(i);
                                num += kelemen_logpdf(smp.z_raw(i) - yi);
                                den += kelemen_logpdf(smp.x(i) - yi);
                            }
                            ratio = __expf(num - den);
                        }
                        if (!lum_invalid(ratio)) {
                            a2 = fminf(1.f, (z.lum / cur.lum) * ratio * (1.f - aRev) / (1.f - a1));
                            acc2 = a2 >= 1.f || u32_to_unit(coins.z) < a2;
                        }
                    }
                } else { // pairwise orbital, DRMLT Eq. 11 (drmlt_proc.cpp:655-669)
                    if (z.lum < y.lum) { a2 = 0.f; }
                    else if (z.lum >= cur.lum) { a2 = 1.f; acc2 = true; }
                    else {
                        a2 = (z.lum - y.lum) / (cur.lum - y.lum);
                        acc2 = a2 >= 1.f || u32_to_unit(coins.z) < a2;
                    }
                }
                break;
            } else { // Green's second-stage acceptance, Eq. 13-14 (drmlt_proc.cpp:599-615)
                ct.acc2b_rev += 1u << 16;
                float aRev = lum_invalid(res.lum) ? 0.f : fminf(1.f, res.lum / z.lum);
                if (aRev != 1.f) {
                    a2 = fminf(1.f, (z.lum / cur.lum) * (1.f - aRev) / (1.f - a1));
                    acc2 = a2 >= 1.f || u32_to_unit(coins.z) < a2;
                }
            }
        }

        if (!mix) {
            // expectation weights, drmlt_proc.cpp:677-688
            float w1 = a1, w2 = (1.f - a1) * a2, w0 = 1.f - w1 - w2;
            if (!amap) {
                if (w0 > 0.f) film_put(P, cur.px, cur.py, mk3(cur.r * w0, cur.g * w0, cur.b * w0));
                if (w1 > 0.f) film_put(P, y.px, y.py, mk3(y.r * w1, y.g * w1, y.b * w1));
                if (doSecond && w2 > 0.f) film_put(P, z.px, z.py, mk3(z.r * w2, z.g * w2, z.b * w2));
            }
        } else {
            // processMixture splats, drmlt_proc.cpp:327-333: a = acceptance of whichever proposal was tested
            const float a = doSecond ? a2 : a1;
            const DSplat &pr = doSecond ? z : y;
            if (1.f - a > 0.f) film_put(P, cur.px, cur.py, mk3(cur.r * (1.f - a), cur.g * (1.f - a), cur.b * (1.f - a)));
            if (a > 0.f) film_put(P, pr.px, pr.py, mk3(pr.r * a, pr.g * a, pr.b * a));
        }

        // bookkeeping (event counts; the 7 ratios are assembled on the host)
        if (large) {
            ct.large_acc1l += 1u + (acc1 ? 1u << 16 : 0u);
            if (doSecond) ct.acc1b_secl += 1u << 16;
            if (acc2) ct.secb_acc2l += 1u << 16;
        } else {
            if (acc1) ct.acc1b_secl += 1u;
            if (doSecond) ct.secb_acc2l += 1u;
            if (acc2) ct.acc2b_rev += 1u;
        }

        if (acc1 || acc2) {
            // DRMLTSampler::accept: uCurrent = wrap(chosen proposal), every kept dimension
            if (acc1) { for (int k = 0; k < D; ++k) lds_x[k * 64 + lane] = wrap01(smp.y_raw((uint32_t) k)); cur = y; }
            else { for (int k = 0; k < D; ++k) lds_x[k * 64 + lane] = wrap01(smp.z_raw((uint32_t) k)); cur = z; }
            if (amap) { // drmlt_proc.cpp:697-709
                if (acc1) { if (!large && !P.use_mixture) film_put(P, cur.px, cur.py, mk3(1.f, 0.f, 0.f)); }
                else if (!P.use_mixture) film_put(P, cur.px, cur.py, mk3(0.f, 1.f, 0.f));
            }
        }
    }

    if (live && !(P.debug & 4)) {
        for (int k = 0; k < D; ++k) P.x[(size_t) k * P.n_chains + c] = lds_x[k * 64 + lane];
        P.cur_lum[c] = cur.lum; P.cur_px[c] = cur.px; P.cur_py[c] = cur.py;
        P.cur_r[c] = cur.r; P.cur_g[c] = cur.g; P.cur_b[c] = cur.b;
    }
    // wave-reduce the event counters, one atomic per counter per wave
    unsigned long long v[9];
    v[0] = wave_sum(ct.large_acc1l & 0xffffu); v[1] = wave_sum(ct.large_acc1l >> 16);
    v[2] = wave_sum(ct.acc1b_secl & 0xffffu);  v[3] = wave_sum(ct.acc1b_secl >> 16);
    v[4] = wave_sum(ct.secb_acc2l & 0xffffu);  v[5] = wave_sum(ct.secb_acc2l >> 16);
    v[6] = wave_sum(ct.acc2b_rev & 0xffffu);   v[7] = wave_sum(ct.acc2b_rev >> 16);
    v[8] = wave_sum(ct.rays);
    if (lane == 0 && !(P.debug & 2))
        for (int i = 0; i < 9; ++i) atomicAdd(P.stats + i, v[i]);
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
            if (P.kelemen_weights) { // :197-203
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
        if (large) ct.large_acc1l += 1u + (accept ? 1u << 16 : 0u);
        else if (accept) ct.acc1b_secl += 1u;
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
    unsigned long long v[9];
    v[0] = wave_sum(ct.large_acc1l & 0xffffu); v[1] = wave_sum(ct.large_acc1l >> 16);
    v[2] = wave_sum(ct.acc1b_secl & 0xffffu);  v[3] = 0; v[4] = 0; v[5] = 0; v[6] = 0; v[7] = 0;
    v[8] = wave_sum(ct.rays);
    if (lane == 0)
        for (int i = 0; i < 9; ++i) if (v[i]) atomicAdd(P.stats + i, v[i]);
}

// ------------------------------------------------------------------------------------------
// k_mutate_v2: the same chain loop as k_mutate, restructured for the wave.
//
// k_mutate nests "for every stage: run the path to completion": a wave then runs as long as its
// longest path, and the second stage (needed by a few lanes only) costs the whole wave a second
// full evaluation -- measured VALU lane utilisation 22 %. Here every lane is an independent
// state machine and one loop iteration is ONE ray step for all lanes, whatever path, stage or
// mutation each of them is in. Lanes whose path has finished park until at least
// `P.mh_batch` of them can take the Metropolis-Hastings bookkeeping branch together (the branch
// is divergent, so its cost is amortised over the lanes that share it). Per lane the arithmetic
// is identical to k_mutate: both kernels produce the same chains.
struct ChainState {
    DSplat cur, y, z;
    float a1, coin_acc1, coin_acc2, coin_mix;
    uint32_t it, nd1, nd2;
    int stage;       // -1: no mutation in flight, 0/1/2: evaluating first / second / reverse
    bool large, do_second;
};

struct MhStamps { unsigned long long digest, splat, commit, start; };

// The bookkeeping branch in four pieces so that k_mutate_v3 can share the two heavy ones
// (committing D_eff dimensions, drawing the next mutation's uniforms) between a chain lane and
// its helper lane:
//   mh_decide  digest the finished evaluation; if the mutation is decided: splats + counters,
//              returns the commit mode (0 none, SM_STAGE1 = adopt y, SM_STAGE2 = adopt z)
//   commit     x[k] = wrap(proposal[k]) for a range of dimensions           (shareable)
//   mh_start   advance to the next evaluation (next stage or next mutation); returns which
//              uniforms must be drawn (0 none, 1 first stage, 2 second stage)
//   fill       Philox draws into LDS for a range of blocks                    (shareable)
DEV int mh_decide(const DParams &P, ChainState &cs, LdsSampler &smp, PathState &ps, Counters &ct) {
    const bool mix = P.use_mixture != 0;
    const bool amap = P.acceptance_map != 0;
    if (cs.stage < 0) return 0; // nothing evaluated yet (first call of a launch)
    bool decided = false;
    float a2 = 0.f;
    bool acc1 = false, acc2 = false;
    DSplat res;
    res.px = ps.px; res.py = ps.py; res.r = ps.Li.x; res.g = ps.Li.y; res.b = ps.Li.z;
    res.lum = luminance3(ps.Li);
    normalize_splat(res, P);
    ct.rays += ps.nrays;
    if (cs.stage == 0) {
        cs.y = res; cs.nd1 = ps.k;
        cs.a1 = 0.f;
        if (!(mix ? lum_invalid_mix(res.lum) : lum_invalid(res.lum))) cs.a1 = fminf(1.f, res.lum / cs.cur.lum);
        acc1 = cs.a1 >= 1.f || (cs.a1 > 0.f && cs.coin_acc1 < cs.a1);
        if (!mix) cs.do_second = !acc1 && (P.timid_after_large || !cs.large);
        else cs.do_second = !cs.large && cs.coin_mix < 0.5f;
        if (cs.do_second) { cs.stage = 1; return 0; }
        decided = true;
    } else if (cs.stage == 1) {
        cs.z = res; cs.nd2 = ps.k;
        acc1 = false;
        if (mix) {
            cs.a1 = 0.f;
            if (!lum_invalid_mix(res.lum)) {
                a2 = fminf(1.f, res.lum / cs.cur.lum);
                acc2 = a2 >= 1.f || cs.coin_acc2 < a2;
            }
        } else if (!lum_invalid(res.lum)) {
            if (P.type == 0) { cs.stage = 2; return 0; } // Green: evaluate the reverse move first
            if (P.type == 1) {
                float aRev = fminf(1.f, cs.y.lum / res.lum);
                if (!(aRev >= 1.f)) {
                    float ratio = 1.f;
                    if (!cs.large) {
                        uint32_t dimStage = max(cs.nd1, cs.nd2) - 1u;
                        float num = 0.f, den = 0.f;
                        for (uint32_t i = 0; i < dimStage; ++i) {
                            float yi = smp.y_raw(i);
                            num += kelemen_logpdf(smp.z_raw(i) - yi);
                            den += kelemen_logpdf(smp.x(i) - yi);
                        }
                        ratio = __expf(num - den);
                    }
                    if (!lum_invalid(ratio)) {
                        a2 = fminf(1.f, (res.lum / cs.cur.lum) * ratio * (1.f - aRev) / (1.f - cs.a1));
                        acc2 = a2 >= 1.f || cs.coin_acc2 < a2;
                    }
                }
            } else {
                if (res.lum < cs.y.lum) { a2 = 0.f; }
                else if (res.lum >= cs.cur.lum) { a2 = 1.f; acc2 = true; }
                else {
                    a2 = (res.lum - cs.y.lum) / (cs.cur.lum - cs.y.lum);
                    acc2 = a2 >= 1.f || cs.coin_acc2 < a2;
                }
            }
        }
        decided = true;
    } else {
        ct.acc2b_rev += 1u << 16;
        float aRev = lum_invalid(res.lum) ? 0.f : fminf(1.f, res.lum / cs.z.lum);
        if (aRev != 1.f) {
            a2 = fminf(1.f, (cs.z.lum / cs.cur.lum) * (1.f - aRev) / (1.f - cs.a1));
            acc2 = a2 >= 1.f || cs.coin_acc2 < a2;
        }
        decided = true;
    }
    if (!decided) return 0;
    // splats of this mutation through ONE film_put site (the call expands to ~150 instructions; six inlined copies
    // of it were a sixth of the kernel's code): slot 0 current state, 1 first-stage, 2 second-stage proposal
    float w0, w1, w2;
    if (!mix) { // expectation weights, drmlt_proc.cpp:677-688
        w1 = cs.a1; w2 = (1.f - cs.a1) * a2; w0 = 1.f - w1 - w2;
        if (!cs.do_second) w2 = 0.f;
        if (amap) w0 = w1 = w2 = 0.f;
    } else { // processMixture, :327-333: a = acceptance of whichever proposal was tested
        const float a = cs.do_second ? a2 : cs.a1;
        w0 = 1.f - a; w1 = cs.do_second ? 0.f : a; w2 = cs.do_second ? a : 0.f;
    }
#pragma nounroll
    for (int i = 0; i < 3; ++i) {
        const float w = i == 0 ? w0 : (i == 1 ? w1 : w2);
        const DSplat sp = select_splat(i == 0, cs.cur, select_splat(i == 1, cs.y, cs.z));
        if (w > 0.f) film_put(P, sp.px, sp.py, mk3(sp.r * w, sp.g * w, sp.b * w));
    }
    if (cs.large) {
        ct.large_acc1l += 1u + (acc1 ? 1u << 16 : 0u);
        if (cs.do_second) ct.acc1b_secl += 1u << 16;
        if (acc2) ct.secb_acc2l += 1u << 16;
    } else {
        if (acc1) ct.acc1b_secl += 1u;
        if (cs.do_second) ct.secb_acc2l += 1u;
        if (acc2) ct.acc2b_rev += 1u;
    }
    int commit = 0;
    if (acc1 || acc2) {
        commit = acc1 ? SM_STAGE1 : SM_STAGE2;
        cs.cur = select_splat(acc1, cs.y, cs.z);
        if (amap && !mix) {
            if (acc1) { if (!cs.large) film_put(P, cs.cur.px, cs.cur.py, mk3(1.f, 0.f, 0.f)); }
            else film_put(P, cs.cur.px, cs.cur.py, mk3(0.f, 1.f, 0.f));
        }
    }
    cs.it++;
    cs.stage = -1;
    return commit;
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

// one lane does everything (k_mutate_v2)
DEV void mh_advance(const DParams &P, ChainState &cs, LdsSampler &smp, PathState &ps, Counters &ct, uint32_t n_mut,
                    uint32_t mut_base, uint32_t lane, MhStamps &ms, bool stamps) {
    const unsigned long long m0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    const int commit = mh_decide(P, cs, smp, ps, ct);
    const unsigned long long m1 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    if (commit) commit_range(smp, commit, 0u, (uint32_t) P.eff_dim);
    const unsigned long long m2 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    const int fill = mh_start(P, cs, smp, ps, n_mut, mut_base);
    const uint32_t D4 = ((uint32_t) P.eff_dim + 3u) & ~3u;
    if (fill == 1) smp.fill_stage1(0u, D4 / 4u);
    else if (fill == 2) smp.fill_stage2(D4, 0u, 1u);
    ms.digest += m1 - m0; ms.commit += m2 - m1;
    ms.start += (stamps ? __builtin_amdgcn_s_memtime() : 0ull) - m2;
}

__global__ void __launch_bounds__(CHAIN_BLOCK) k_mutate_v2(DParams P, uint32_t n_mut, uint32_t mut_base) {
    const uint32_t lane = threadIdx.x;
    // experiment (DRMLT_DEBUG bit 512): 32 chains per wave in lanes 0..31, twice the waves
    const uint32_t per_wave = (P.debug & 512) ? 32u : 64u;
    const uint32_t c = blockIdx.x * per_wave + lane;
    const bool live = lane < per_wave && c < P.n_chains;
    const uint32_t cc = live ? c : P.n_chains - 1;
    const int D = P.eff_dim;
    for (int k = 0; k < D; ++k) lds_x[(uint32_t) k * per_wave + lane] = P.x[(size_t) k * P.n_chains + cc];

    ChainState cs;
    cs.cur.lum = P.cur_lum[cc]; cs.cur.px = P.cur_px[cc]; cs.cur.py = P.cur_py[cc];
    cs.cur.r = P.cur_r[cc]; cs.cur.g = P.cur_g[cc]; cs.cur.b = P.cur_b[cc];
    cs.y = cs.cur; cs.z = cs.cur;
    cs.a1 = 0.f; cs.coin_acc1 = cs.coin_acc2 = cs.coin_mix = 0.f;
    cs.it = 0u; cs.nd1 = cs.nd2 = 0u; cs.stage = -1; cs.large = false; cs.do_second = false;

    LdsSampler smp;
    smp.key0 = P.key0; smp.key1 = P.key1; smp.chain = P.chain_offset + cc; smp.major = 0u;
    smp.mode = SM_STAGE1; smp.type = P.type; smp.large = false; smp.sigma2 = P.sigma2; smp.lane = lane;
    const uint32_t D4 = ((uint32_t) D + 3u) & ~3u;
    smp.stride = per_wave;
    smp.u1_off = (uint32_t) D * per_wave;
    smp.s2_off = smp.u1_off + D4 * per_wave;
    smp.timing_probe = (P.debug & 256) != 0;
    Counters ct = {0u, 0u, 0u, 0u, 0u};
    PathState ps;
    path_init(P, ps);
    ps.phase = (live && n_mut > 0u) ? PH_DONE : PH_IDLE; // PH_DONE with stage -1: "start the first mutation"
    Hit h{-1, 0.f, 0.f, 0.f};
    ShadowRay sr_unused;
    const int batch = P.mh_batch;
    // scene tables: staged in LDS behind the sampler rows when they are small
    LdsTables LT;
    LT.shade_off = smp.s2_off + D4 * per_wave;
    LT.bsdf_off = LT.shade_off + (uint32_t) P.n_shade * 16u;
    LT.emit_off = LT.bsdf_off + (uint32_t) P.n_bsdfs * 12u;
    const GlobalTables GT{P.shade, P.bsdfs, P.emitters};
    const bool lds_tables = P.tables_in_lds != 0;
    if (lds_tables) stage_tables(P, LT, lane);

    // diagnostic stamps (DRMLT_DEBUG bit 128): per-wave cycle shares of the three loop sections
    const bool stamps = (P.debug & 128) != 0;
    unsigned long long t_mh = 0, t_trace = 0, t_step = 0, n_iter = 0, n_mh = 0, n_busy = 0;
    MhStamps ms = {0, 0, 0, 0};
#define STAMP() (stamps ? __builtin_amdgcn_s_memtime() : 0ull)
    for (;;) {
        const bool parked = ps.phase == PH_DONE;
        const unsigned long long pmask = __ballot(parked);
        const unsigned long long rmask = __ballot(ps.phase != PH_DONE && ps.phase != PH_IDLE);
        if (!pmask && !rmask) break;
        const unsigned long long s0 = STAMP();
        if (pmask && (__popcll(pmask) >= batch || !rmask)) {
            if (parked) mh_advance(P, cs, smp, ps, ct, n_mut, mut_base, lane, ms, stamps);
            n_mh++;
        }
        const unsigned long long s1 = STAMP();
        const bool tracing = ps.phase == PH_CLOSEST || ps.phase == PH_SHADOW;
        if (stamps) n_busy += __popcll(__ballot(tracing));
        if (tracing) h = trace(P, ps.o, ps.d, ps.tmin, ps.tmax, ps.phase == PH_SHADOW);
        const unsigned long long s2 = STAMP();
        if (ps.phase != PH_DONE && ps.phase != PH_IDLE) {
            if (lds_tables) path_step<false, 15>(P, LT, ps, smp, h, false, sr_unused);
            else path_step<false, 15>(P, GT, ps, smp, h, false, sr_unused);
        }
        const unsigned long long s3 = STAMP();
        t_mh += s1 - s0; t_trace += s2 - s1; t_step += s3 - s2; n_iter++;
    }
#undef STAMP
    if (stamps && lane == 0) {
        atomicAdd(P.stats + 16, t_mh); atomicAdd(P.stats + 17, t_trace); atomicAdd(P.stats + 18, t_step);
        atomicAdd(P.stats + 19, n_iter); atomicAdd(P.stats + 20, n_mh); atomicAdd(P.stats + 21, n_busy);
        atomicAdd(P.stats + 22, ms.digest); atomicAdd(P.stats + 23, ms.splat); atomicAdd(P.stats + 24, ms.commit); atomicAdd(P.stats + 25, ms.start);
    }

    if (live) {
        for (int k = 0; k < D; ++k) P.x[(size_t) k * P.n_chains + c] = lds_x[(uint32_t) k * per_wave + lane];
        P.cur_lum[c] = cs.cur.lum; P.cur_px[c] = cs.cur.px; P.cur_py[c] = cs.cur.py;
        P.cur_r[c] = cs.cur.r; P.cur_g[c] = cs.cur.g; P.cur_b[c] = cs.cur.b;
    }
    unsigned long long v[9];
    v[0] = wave_sum(ct.large_acc1l & 0xffffu); v[1] = wave_sum(ct.large_acc1l >> 16);
    v[2] = wave_sum(ct.acc1b_secl & 0xffffu);  v[3] = wave_sum(ct.acc1b_secl >> 16);
    v[4] = wave_sum(ct.secb_acc2l & 0xffffu);  v[5] = wave_sum(ct.secb_acc2l >> 16);
    v[6] = wave_sum(ct.acc2b_rev & 0xffffu);   v[7] = wave_sum(ct.acc2b_rev >> 16);
    v[8] = wave_sum(ct.rays);
    if (lane == 0)
        for (int i = 0; i < 9; ++i) atomicAdd(P.stats + i, v[i]);
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
            if (parked) commit = mh_decide(P, cs, smp, ps, ct);
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
    unsigned long long v[9];
    v[0] = wave_sum(ct.large_acc1l & 0xffffu); v[1] = wave_sum(ct.large_acc1l >> 16);
    v[2] = wave_sum(ct.acc1b_secl & 0xffffu);  v[3] = wave_sum(ct.acc1b_secl >> 16);
    v[4] = wave_sum(ct.secb_acc2l & 0xffffu);  v[5] = wave_sum(ct.secb_acc2l >> 16);
    v[6] = wave_sum(ct.acc2b_rev & 0xffffu);   v[7] = wave_sum(ct.acc2b_rev >> 16);
    v[8] = wave_sum(ct.rays);
    if (lane == 0)
        for (int i = 0; i < 9; ++i) atomicAdd(P.stats + i, v[i]);
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
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = film[i] * (importance ? factor * importance[i / 3u] : factor) + (direct ? direct[i] : 0.f); // :841-847
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
    dim3 grid((P.n_chains + CHAIN_BLOCK - 1) / CHAIN_BLOCK), block(CHAIN_BLOCK);
    if (P.kernel_variant == 1) {
        hipLaunchKernelGGL(k_mutate, grid, block, D * 64 * sizeof(float), st, P, n_mut, mut_base);
    } else if (P.kernel_variant == 3) { // 32 chains per wave, rows of 32 floats
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
    } else { // x + first-stage uniforms + second-stage values, one 256 B row per dimension
        size_t lds = (D + 2 * D4) * 64 * sizeof(float);
        if (P.debug & 512) { grid = dim3((P.n_chains + 31) / 32); lds /= 2; }
        if (P.tables_in_lds) lds += (size_t) P.n_shade * 64 + (size_t) P.n_bsdfs * 48 + (size_t) P.n_emitters * 32;
        hipLaunchKernelGGL(k_mutate_v2, grid, block, lds, st, P, n_mut, mut_base);
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
void launch_develop(const float *film, const float *direct, const float *importance, float factor, uint32_t n, float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_develop, dim3((n + 255) / 256), dim3(256), 0, st, film, direct, importance, factor, n, out);
}
