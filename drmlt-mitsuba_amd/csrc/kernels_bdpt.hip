// HIP kernels of the DRMLT hot path for technique=bdpt (gfx950). One Markov chain per lane, one wave per workgroup.
//
//   k_bootstrap_bdpt    luminance samples of generateSeeds (pathsampler.cpp:879-920)
//   k_init_chains_bdpt  seed replay + fillReplay of the sensor / emitter samplers (drmlt_proc.cpp:467-514)
//   k_mutate_bdpt       DRMLTRenderer::process / processMixture over sampleSplats(EBidirectional) (drmlt_proc.cpp:161-380,518-770)
//   k_eval_lists_bdpt   sampleSplats(EBidirectional) on caller-supplied PSS points, full splat lists
//
// LDS rows ([row][lane]): the walks' part of the chain state [0, S + E) (bdpt_dims_sensor / _emitter), then the two row
// groups of eval_bdpt and its row of segment heads: 77 rows = 19.25 KB for maxDepth 8 -- under the 20 KB that put eight waves on a CU. The direct
// sampler's Dd components (directSampling = true) stay in memory: a sample reads at most a few of them. Splat lists live in HBM (`bd_lists`, slot 0 = current state), unnormalised next to their luminance.
#include <cstdlib>
#include "device_bdpt.h"
#include "kernel_common.h"

DEV uint32_t bdpt_nx(const DParams &P) { return (uint32_t) (P.mmlt_S + P.mmlt_E + P.bd_Dd); } // components of a chain's state
DEV uint32_t bdpt_nx_lds(const DParams &P) { return (uint32_t) (P.mmlt_S + P.mmlt_E); }        // ... of which LDS holds the two walks'

DEV void bsampler_setup(MSampler &smp, const DParams &P, uint32_t lane) {
    smp.key0 = P.key0; smp.key1 = P.key1;
    smp.type = P.type; smp.sigma2 = P.sigma2; smp.large = false;
    smp.lane = lane; smp.arr = nullptr;
    smp.S = (uint32_t) P.mmlt_S; smp.E = (uint32_t) P.mmlt_E;
    smp.base_e = 2u * (uint32_t) P.mmlt_dmax; smp.base_d = 4u * (uint32_t) P.mmlt_dmax;
    smp.emitter_ident2 = false; smp.direct_ident = false; smp.x_dir = nullptr; smp.x_dir_n = 0u;
    smp.reset_caches();
    smp.select(SEG_SENSOR);
}

DEV float *list_col(const DParams &P, int slot, uint32_t chain) {
    return P.bd_lists + (size_t) slot * (size_t) bdpt_list_rows(P.max_depth) * P.n_chains_alloc + chain;
}

// SplatList::normalize with a two-stage importance map (pathsampler.cpp:1001-1020): weigh every splat, recompute the
// list luminance. Lists stay unnormalised otherwise; the 1 / luminance factor is applied when splatting.
DEV float list_finalize(const DParams &P, float *list, float lum) {
    if (!P.importance) return lum;
    const size_t n = P.n_chains_alloc;
    const int meta = __float_as_int(list[(size_t) BL_META * n]);
    const int cnt = meta >> 1;
    float total = 0.f;
    for (int k = -1; k < cnt; ++k) {
        if (k < 0 && !(meta & 1)) continue;
        const int r0 = k < 0 ? BL_MAIN : BL_MORE + 5 * k;
        float r = list[(size_t) (r0 + 2) * n], g = list[(size_t) (r0 + 3) * n], b = list[(size_t) (r0 + 4) * n];
        if (r == 0.f && g == 0.f && b == 0.f) continue;
        const int ix = min(max(0, (int) list[(size_t) r0 * n]), P.width - 1), iy = min(max(0, (int) list[(size_t) (r0 + 1) * n]), P.height - 1);
        const float lv = P.importance[ix + iy * P.width];
        r /= lv; g /= lv; b /= lv;
        list[(size_t) (r0 + 2) * n] = r; list[(size_t) (r0 + 3) * n] = g; list[(size_t) (r0 + 4) * n] = b;
        total += luminance3(mk3(r, g, b));
    }
    list[(size_t) BL_LUM * n] = total;
    return total;
}

// splat every entry of a list with weight w / luminance (the list is stored unnormalised)
DEV void list_splat(const DParams &P, const float *list, float lum, float w) {
    if (!(w > 0.f) || !(lum > 0.f)) return;
    const size_t n = P.n_chains_alloc;
    const float sc = w / lum;
    const int meta = __float_as_int(list[(size_t) BL_META * n]);
    if (meta & 1)
        film_put(P, list[(size_t) BL_MAIN * n], list[(size_t) (BL_MAIN + 1) * n],
                 mk3(list[(size_t) (BL_MAIN + 2) * n] * sc, list[(size_t) (BL_MAIN + 3) * n] * sc, list[(size_t) (BL_MAIN + 4) * n] * sc));
    const int cnt = meta >> 1;
    for (int k = 0; k < cnt; ++k) {
        const int r0 = BL_MORE + 5 * k;
        film_put(P, list[(size_t) r0 * n], list[(size_t) (r0 + 1) * n],
                 mk3(list[(size_t) (r0 + 2) * n] * sc, list[(size_t) (r0 + 3) * n] * sc, list[(size_t) (r0 + 4) * n] * sc));
    }
}
DEV void list_splat_const(const DParams &P, const float *list, f3 c) { // acceptance map: every splat position (:443-450)
    const size_t n = P.n_chains_alloc;
    const int meta = __float_as_int(list[(size_t) BL_META * n]);
    if (meta & 1) film_put(P, list[(size_t) BL_MAIN * n], list[(size_t) (BL_MAIN + 1) * n], c);
    for (int k = 0; k < (meta >> 1); ++k) film_put(P, list[(size_t) (BL_MORE + 5 * k) * n], list[(size_t) (BL_MORE + 5 * k + 1) * n], c);
}
DEV void list_copy(const DParams &P, float *dst, const float *src) {
    const size_t n = P.n_chains_alloc;
    const int rows = BL_MORE + 5 * (__float_as_int(src[(size_t) BL_META * n]) >> 1);
    for (int r = 0; r < rows; ++r) dst[(size_t) r * n] = src[(size_t) r * n];
}

__global__ void __launch_bounds__(CHAIN_BLOCK) k_bootstrap_bdpt(DParams P, uint32_t n, float *lum_out) {
    const uint32_t lane = threadIdx.x;
    const uint32_t c = blockIdx.x * CHAIN_BLOCK + lane; // workspace column
    const bool has_col = c < P.n_chains_alloc;
    const uint32_t cc = has_col ? c : P.n_chains_alloc - 1u;
    MSampler smp;
    bsampler_setup(smp, P, lane);
    smp.chain = P.boot_stream; smp.mode = SM_BOOT;
    const GlobalTables T{P.shade, P.bsdfs, P.emitters};
    for (uint32_t i0 = blockIdx.x * CHAIN_BLOCK; i0 < n; i0 += P.n_chains_alloc) { // the whole wave makes every call (eval_bdpt)
        const uint32_t i = i0 + lane;
        const bool active = has_col && i < n;
        smp.major = i;
        BdptResult R;
        eval_bdpt(P, T, smp, active, cc, bdpt_nx_lds(P), list_col(P, 1, cc), R);
        if (active) {
            lum_out[i] = R.lum;
            if (P.boot_weighted) lum_out[(size_t) n + i] = list_finalize(P, list_col(P, 1, cc), R.lum); // luminance of f / importance
        }
    }
}

__global__ void __launch_bounds__(CHAIN_BLOCK) k_init_chains_bdpt(DParams P, const uint32_t *seed_index, const float *seed_lum) {
    const uint32_t lane = threadIdx.x;
    const uint32_t c0 = blockIdx.x * CHAIN_BLOCK + lane;
    const bool live = c0 < P.n_chains;
    const uint32_t c = live ? c0 : P.n_chains - 1u;
    MSampler smp;
    bsampler_setup(smp, P, lane);
    smp.chain = P.boot_stream; smp.major = seed_index[c]; smp.mode = SM_BOOT;
    const GlobalTables T{P.shade, P.bsdfs, P.emitters};
    BdptResult R;
    float *cur = list_col(P, 0, c);
    eval_bdpt(P, T, smp, live, c, bdpt_nx_lds(P), cur, R);
    if (!live) return;
    if (!(fabsf((R.lum - seed_lum[c]) / seed_lum[c]) <= EPSILON_F)) atomicExch(P.error_flag, 1); // drmlt_proc.cpp:509-512
    const float lum = list_finalize(P, cur, R.lum);
    P.cur_lum[c] = lum;
    // replayed stream, in call order: emitter (n_e), sensor (n_s), direct (n_d); fillReplay then tops up the sensor and
    // the emitter sampler to D and the direct sampler to Dd, in that order (drmlt_proc.cpp:506-509)
    const uint32_t D = (uint32_t) P.mmlt_dmax, ne = R.n_emitter, ns = R.n_sensor, nd = R.n_direct;
    const uint32_t S = (uint32_t) P.mmlt_S, E = (uint32_t) P.mmlt_E, Dd = (uint32_t) P.bd_Dd;
    smp.b1_idx = 0xffffffffu;
    for (uint32_t k = 0; k < S; ++k) P.x[(size_t) k * P.n_chains + c] = smp.u_boot(k < ns ? ne + k : ne + nd + k);
    for (uint32_t k = 0; k < E; ++k) P.x[(size_t) (S + k) * P.n_chains + c] = smp.u_boot(k < ne ? k : nd + D + k);
    for (uint32_t k = 0; k < Dd; ++k) P.x[(size_t) (S + E + k) * P.n_chains + c] = smp.u_boot(k < nd ? ne + ns + k : 2u * D + k);
}

// FEAT 7: scenes without a BVH -- no traversal stack in LDS (6 KB), which is what keeps four waves on a CU with
// directSampling = true (35 KB of rows per wave; measured 4.3e7 -> 8e7 mutations/s, all of it occupancy)
// OCC 2: registers capped at 256 (the rest spills to scratch) so that two waves share a SIMD -- chosen by the launcher when
// there are waves to fill them (more than 65 536 chains): a single wave64 issues a vector instruction every 4 cycles at
// best, the SIMD one every 2.
// LDS_BSDFS (round 4): BSDF and emitter records staged in LDS behind the kernel's rows (device_path.h: MixedTables) -- every walk step and
// every connection cell gathers one or two BSDF records per lane; the shading table does not fit beside 19.25 KB of rows at two waves per SIMD.
template <int FEAT, int OCC, bool LDS_BSDFS = false> __global__ void __launch_bounds__(CHAIN_BLOCK, OCC) k_mutate_bdpt(DParams P, uint32_t n_mut, uint32_t mut_base) {
    const uint32_t lane = threadIdx.x;
    MixedTables MT;
    MT.sh = P.shade;
    MT.L.shade_off = ((uint32_t) P.mmlt_S + (uint32_t) P.mmlt_E) * 64u + (uint32_t) bdpt_eval_lds_floats(P.max_depth); // (the emitters' shape records)
    MT.L.bsdf_off = MT.L.shade_off + (uint32_t) P.n_emitters * 16u;
    MT.L.emit_off = MT.L.bsdf_off + (uint32_t) P.n_bsdfs * 12u;
    if (LDS_BSDFS) stage_bsdfs_emitters(P, MT.L, lane);
    // Execution order (drmlt_capi.cpp: regroup_chains): which chain a lane runs. Between the launches of a call the host groups the
    // chains by the evaluations they needed in the launch just done, so that chains parked on a glint (two evaluations per mutation,
    // launch after launch) share waves instead of holding sixty-three finished lanes each. Chain ids -- state, streams, workspace
    // column, lists -- are untouched: the same chains bit for bit, in other lanes.
    const uint32_t slot = blockIdx.x * CHAIN_BLOCK + lane;
    const uint32_t c = P.exec_order ? P.exec_order[slot] : slot;
    const bool live = c < P.n_chains;
    const uint32_t cc = live ? c : P.n_chains - 1;
    const uint32_t NX = bdpt_nx_lds(P);
    for (uint32_t k = 0; k < NX; ++k) lds_x[k * 64u + lane] = P.x[(size_t) k * P.n_chains + cc];
    float cur_lum = P.cur_lum[cc];
    float *L0 = list_col(P, 0, cc), *L1 = list_col(P, 1, cc), *L2 = list_col(P, 2, cc);
    float cum = 0.f; // weight the current state has gathered since it was adopted

    MSampler smp;
    bsampler_setup(smp, P, lane);
    smp.chain = P.chain_offset + cc;
    float *const xdir = P.x + (size_t) NX * P.n_chains + cc; // the direct sampler's components of this chain
    smp.x_dir = xdir; smp.x_dir_n = P.n_chains;
    const GlobalTables T{P.shade, P.bsdfs, P.emitters};
    Counters ct = {0u, 0u, 0u, 0u, 0u};
    const bool amap = P.acceptance_map != 0;
    const bool mix = P.use_mixture != 0;

    const bool dbg = (P.debug & 128) != 0;
    unsigned long long t_stages = 0, t_splat = 0, t_commit = 0;
#define BSTAMP() (dbg ? __builtin_amdgcn_s_memtime() : 0ull)
    const unsigned long long k0 = BSTAMP();
    // Chains run free: every pass of the loop evaluates ONE path per lane -- the first stage of the lane's next mutation, or the
    // second stage / Green's reverse move of the one it is in. (A lockstep loop over mutations with the stages inside ran a whole
    // wave pass for the two or three lanes in sixty-four that go to a second stage -- 2 % of bdpt's mutations, so three passes in
    // four had one: 1.8 passes per mutation instead of 1.02.)
    uint32_t it = 0u, work = 0u; // work: path evaluations of this launch
    int stage = 0;
    float y_lum = 0.f, z_lum = 0.f, a1 = 0.f;
    uint32_t ns1 = 0, ne1 = 0, nd1 = 0, ns2 = 0, ne2 = 0, nd2 = 0;
    for (;;) {
        const bool run = live && it < n_mut;
        if (!__builtin_amdgcn_ballot_w64(run)) break;
        const uint32_t m = mut_base + it;
        const u4 coins = philox4x32_10(P.key0, P.key1, 0u, m, smp.chain, TAG_COIN);
        const bool large = u32_to_unit(coins.x) < P.p_large;
        smp.major = m;
        smp.large = large;
        const unsigned long long b0 = BSTAMP();
        float a2 = 0.f;
        bool acc1 = false, acc2 = false, decided = true;
        smp.mode = stage == 0 ? SM_STAGE1 : (stage == 1 ? SM_STAGE2 : SM_REVERSE);
        BdptResult R;
        // Green's reverse path only needs its luminance; it is written over the first-stage list, which is rejected
        // for good by then (acc1 = false) and has already been splatted (see below)
        float *target = stage == 1 ? L2 : L1;
        if (LDS_BSDFS) eval_bdpt<FEAT>(P, MT, smp, run, cc, NX, target, R); // the whole wave: the connections of all chains go to all lanes
        else eval_bdpt<FEAT>(P, T, smp, run, cc, NX, target, R);
        if (!run) continue;
        ++work;
        {
            ct.rays += R.nrays;
            const float lum = list_finalize(P, target, R.lum);
            if (stage == 0) {
                bool doSecond = false;
                y_lum = lum; ns1 = R.n_sensor; ne1 = R.n_emitter; nd1 = R.n_direct;
                z_lum = 0.f; ns2 = ne2 = nd2 = 0u;
                mh_first(mix, P.timid_after_large != 0, large, y_lum, cur_lum, u32_to_unit(coins.y), u32_to_unit(coins.w), a1, acc1, doSecond);
                if (doSecond) { stage = 1; decided = false; }
            } else if (stage == 1) {
                z_lum = lum; ns2 = R.n_sensor; ne2 = R.n_emitter; nd2 = R.n_direct;
                if (mix) { // the second proposal replaces the first
                    a1 = 0.f;
                    mh_second_mixture(z_lum, cur_lum, u32_to_unit(coins.z), a2, acc2);
                } else if (lum_invalid(z_lum)) {
                } else if (P.type == 0) {
                    // Green: the first-stage splats must reach the film before the reverse move reuses their list
                    if (!amap && a1 > 0.f) list_splat(P, L1, y_lum, a1);
                    stage = 2; decided = false;
                } else if (P.type == 1) {
                    float ratio = 1.f;
                    if (!large && !(fminf(1.f, y_lum / z_lum) >= 1.f)) { // (a large step -- here only with timidAfterLarge -- has no kernel ratio: uniform proposals both times)
                        float num = 0.f, den = 0.f;
                        for (int sg = 0; sg < 3; ++sg) {
                            const uint32_t nmax = sg == 0 ? max(ns1, ns2) : (sg == 1 ? max(ne1, ne2) : max(nd1, nd2));
                            const uint32_t dimStage = nmax > 0u ? nmax - 1u : 0u;
                            smp.select(sg);
                            for (uint32_t i = 0; i < dimStage; ++i) {
                                float yi = smp.y_raw(i);
                                num += kelemen_logpdf(smp.z_raw(i) - yi);
                                den += kelemen_logpdf(smp.x(i) - yi);
                            }
                        }
                        ratio = __expf(num - den);
                    }
                    mh_second_mira(y_lum, z_lum, cur_lum, a1, ratio, u32_to_unit(coins.z), a2, acc2);
                } else {
                    mh_second_orbital(y_lum, z_lum, cur_lum, u32_to_unit(coins.z), a2, acc2);
                }
            } else {
                ct.acc2b_rev += 1u << 16;
                mh_second_green(lum, z_lum, cur_lum, a1, u32_to_unit(coins.z), a2, acc2);
            }
        }
        const unsigned long long b1 = BSTAMP();
        if (!decided) { t_stages += b1 - b0; continue; }
        const bool doSecond = stage != 0;
        const bool y_splatted = stage == 2 && !amap; // Green went on to the reverse move

        // Expectation weights (device_mh.h). The current state's share is accumulated and its list splatted once, when the
        // state is replaced or the launch ends (the reference's pssmlt loop does the same, pssmlt_proc.cpp:205-228): the
        // same film, one list splat per mutation instead of two or three. An adopted proposal carries its weight into
        // `cum`. In an acceptance-map run of the delayed-rejection loop all three weights are zero.
        const MhWeights w = mh_weights(mix, amap, doSecond, a1, a2);
        cum += w.w0;
        if (acc1 || acc2) { list_splat(P, L0, cur_lum, cum); cum = acc1 ? w.w1 : w.w2; }
        if (!acc1 && !y_splatted) list_splat(P, L1, y_lum, w.w1);
        if (!acc2) list_splat(P, L2, z_lum, w.w2);

        const unsigned long long b2 = BSTAMP();
        mh_count(ct, large, acc1, acc2, doSecond);

        if (acc1 || acc2) {
            for (int sg = 0; sg < 3; ++sg) { // every component of the three samplers (DRMLTSampler::accept, drmlt_sampler.cpp:189-199)
                smp.select(sg);
                const uint32_t nk = sg == 0 ? smp.S : (sg == 1 ? smp.E : (uint32_t) P.bd_Dd);
                for (uint32_t k = 0; k < nk; ++k) {
                    const float v = wrap01(acc1 ? smp.y_raw(k) : smp.z_raw(k));
                    if (sg == 2) xdir[(size_t) k * P.n_chains] = v;
                    else lds_x[(smp.x_off + k) * 64u + lane] = v;
                    if (smp.type == 2 && (k & 1u)) smp.pair_base = 0xffffffffu;
                }
            }
            { float *t = L0; if (acc1) { L0 = L1; L1 = t; } else { L0 = L2; L2 = t; } } // the accepted list becomes the current one: swap, do not copy
            cur_lum = acc1 ? y_lum : z_lum;
            // acceptance map: every splat position of the list that WAS current -- after the swap that is the proposal's
            // slot, exactly as in the reference (drmlt_proc.cpp:693-709; device_mh.h)
            const int mark = mh_amap_mark(mix, amap, large, acc1, acc2);
            if (mark) list_splat_const(P, acc1 ? L1 : L2, mh_amap_colour(mark));
        }
        ++it;
        stage = 0;
        const unsigned long long b3 = BSTAMP();
        t_stages += b1 - b0; t_splat += b2 - b1; t_commit += b3 - b2;
    }

    if (live) {
        list_splat(P, L0, cur_lum, cum); // "perform the last splat" (cum stays 0 through an acceptance-map run)
        if (L0 != list_col(P, 0, cc)) list_copy(P, list_col(P, 0, cc), L0); // slot 0 is where the next launch (and drmlt_chain_state) look
        for (uint32_t k = 0; k < NX; ++k) P.x[(size_t) k * P.n_chains + c] = lds_x[k * 64u + lane];
        P.cur_lum[c] = cur_lum;
        if (P.chain_done) P.chain_done[c] = work; // (this kernel has no run-ahead: the array is free for the count)
    }
    flush_counters(P, ct, lane);
    if (dbg && lane == 0) { atomicAdd(P.stats + 18, __builtin_amdgcn_s_memtime() - k0); atomicAdd(P.stats + 20, t_stages); atomicAdd(P.stats + 21, t_splat); atomicAdd(P.stats + 22, t_commit); } // per wave
#undef BSTAMP
}

// u: [sensor S | emitter E | direct Dd] per point (dim >= S + E + Dd); out: rows of `stride` floats:
// [lum, hasMain, px, py, r, g, b, nMore, nDims, nRays, nMore x (px, py, r, g, b)]
__global__ void __launch_bounds__(CHAIN_BLOCK) k_eval_lists_bdpt(DParams P, const float *u, uint32_t n, uint32_t dim, float *out, uint32_t stride) {
    const uint32_t lane = threadIdx.x;
    const uint32_t c0 = blockIdx.x * CHAIN_BLOCK + lane;
    const bool has_col = c0 < P.n_chains_alloc;
    const uint32_t c = has_col ? c0 : P.n_chains_alloc - 1u;
    MSampler smp;
    bsampler_setup(smp, P, lane);
    smp.chain = 0u; smp.major = 0u; smp.mode = SM_ARRAY;
    const GlobalTables T{P.shade, P.bsdfs, P.emitters};
    const size_t na = P.n_chains_alloc;
    for (uint32_t i0 = blockIdx.x * CHAIN_BLOCK; i0 < n; i0 += P.n_chains_alloc) { // the whole wave makes every call (eval_bdpt)
        const uint32_t i = i0 + lane;
        const bool active = has_col && i < n;
        smp.arr = u + (size_t) (active ? i : 0u) * dim;
        BdptResult R;
        float *L = list_col(P, 1, c);
        eval_bdpt(P, T, smp, active, c, bdpt_nx_lds(P), L, R);
        if (!active) continue;
        float *o = out + (size_t) i * stride;
        for (uint32_t k = 0; k < stride; ++k) o[k] = 0.f;
        o[0] = R.lum; o[1] = R.has_main ? 1.f : 0.f;
        for (int k = 0; k < 5; ++k) o[2 + k] = L[(size_t) (BL_MAIN + k) * na];
        o[7] = (float) R.n_more; o[8] = (float) (R.n_sensor + R.n_emitter + R.n_direct); o[9] = (float) R.nrays;
        for (int k = 0; k < R.n_more && 10 + 5 * (k + 1) <= (int) stride; ++k)
            for (int q = 0; q < 5; ++q) o[10 + 5 * k + q] = L[(size_t) (BL_MORE + 5 * k + q) * na];
    }
}

static size_t bdpt_lds_bytes(const DParams &P) {
    static const size_t pad = getenv("DRMLT_BDPT_LDS_PAD") ? (size_t) atoi(getenv("DRMLT_BDPT_LDS_PAD")) : 0; // diagnostic: occupancy experiments
    return pad + (((size_t) P.mmlt_S + P.mmlt_E) * 64 + (size_t) bdpt_eval_lds_floats(P.max_depth)) * sizeof(float);
}
void launch_bootstrap_bdpt(const DParams &P, uint32_t n, float *lum_out, hipStream_t st) {
    hipLaunchKernelGGL(k_bootstrap_bdpt, dim3((P.n_chains_alloc + CHAIN_BLOCK - 1) / CHAIN_BLOCK), dim3(CHAIN_BLOCK), bdpt_lds_bytes(P), st, P, n, lum_out);
}
void launch_init_chains_bdpt(const DParams &P, const uint32_t *seed_index, const float *seed_lum, hipStream_t st) {
    hipLaunchKernelGGL(k_init_chains_bdpt, dim3((P.n_chains + CHAIN_BLOCK - 1) / CHAIN_BLOCK), dim3(CHAIN_BLOCK), bdpt_lds_bytes(P), st, P, seed_index,
                       seed_lum);
}
void launch_mutate_bdpt(const DParams &P, uint32_t n_mut, uint32_t mut_base, hipStream_t st) {
    const dim3 grid((P.n_chains + CHAIN_BLOCK - 1) / CHAIN_BLOCK), block(CHAIN_BLOCK);
    static const int force_occ = getenv("DRMLT_BDPT_OCC") ? atoi(getenv("DRMLT_BDPT_OCC")) : 0; // diagnostic
    const bool two = force_occ ? force_occ == 2 : (!P.use_bvh && grid.x > 1024u + 256u && bdpt_lds_bytes(P) <= 20480);
    const size_t tb = ((size_t) P.n_bsdfs * 12 + (size_t) P.n_emitters * (8 + 16)) * sizeof(float); // BSDFs, emitters, the emitters' shape records
    static const bool global_tables = getenv("DRMLT_BDPT_TABLES_GLOBAL") != nullptr; // A/B
    if (P.use_bvh) hipLaunchKernelGGL((k_mutate_bdpt<15, 1>), grid, block, bdpt_lds_bytes(P), st, P, n_mut, mut_base);
    else if (two && !global_tables && bdpt_lds_bytes(P) + tb <= 20480) hipLaunchKernelGGL((k_mutate_bdpt<7, 2, true>), grid, block, bdpt_lds_bytes(P) + tb, st, P, n_mut, mut_base);
    else if (two) hipLaunchKernelGGL((k_mutate_bdpt<7, 2>), grid, block, bdpt_lds_bytes(P), st, P, n_mut, mut_base);
    else hipLaunchKernelGGL((k_mutate_bdpt<7, 1>), grid, block, bdpt_lds_bytes(P), st, P, n_mut, mut_base);
}
void launch_eval_lists_bdpt(const DParams &P, const float *u, uint32_t n, uint32_t dim, float *out, uint32_t stride, hipStream_t st) {
    hipLaunchKernelGGL(k_eval_lists_bdpt, dim3((P.n_chains_alloc + CHAIN_BLOCK - 1) / CHAIN_BLOCK), dim3(CHAIN_BLOCK), bdpt_lds_bytes(P), st, P, u, n, dim, out,
                       stride);
}
