// HIP kernels of the DRMLT hot path for technique=mmlt (gfx950). One Markov chain per lane, one wave per workgroup.
//
//   k_bootstrap_mmlt    luminance samples of generateSeeds with depth = (i % maxDepth) + 1 (pathsampler.cpp:879-920)
//   k_init_chains_mmlt  seed replay + fillReplay of the three samplers (drmlt_proc.cpp:467-514)
//   k_mutate_mmlt       DRMLTRenderer::process / processMixture over sampleSplats(EMMLT) (drmlt_proc.cpp:161-380,518-770)
//   k_eval_paths_mmlt   sampleSplats(EMMLT) on caller-supplied PSS points
//
// LDS rows (256 B each, [row][lane]): chain state [0, NX), NX = S + E + 1; then the MIS scratch of eval_mmlt.
#include "device_bidir.h"
#include "kernel_common.h"

DEV uint32_t mmlt_nx(const DParams &P) { return (uint32_t) (P.mmlt_S + P.mmlt_E + 1); }

DEV void msampler_setup(MSampler &smp, const DParams &P, uint32_t lane) {
    smp.key0 = P.key0; smp.key1 = P.key1;
    smp.type = P.type; smp.sigma2 = P.sigma2; smp.large = false;
    smp.lane = lane; smp.arr = nullptr;
    smp.S = (uint32_t) P.mmlt_S; smp.E = (uint32_t) P.mmlt_E;
    smp.base_e = 2u * (uint32_t) P.mmlt_dmax; smp.base_d = 4u * (uint32_t) P.mmlt_dmax;
    smp.emitter_ident2 = false; smp.direct_ident = true; smp.x_dir = nullptr; smp.x_dir_n = 0u;
    smp.reset_caches();
    smp.select(SEG_SENSOR);
}

__global__ void __launch_bounds__(CHAIN_BLOCK) k_bootstrap_mmlt(DParams P, uint32_t n, float *lum_out) {
    const uint32_t lane = threadIdx.x;
    const uint32_t i = blockIdx.x * CHAIN_BLOCK + lane;
    if (i >= n) return;
    MSampler smp;
    msampler_setup(smp, P, lane);
    smp.chain = P.boot_stream; smp.major = i; smp.mode = SM_BOOT;
    const GlobalTables T{P.shade, P.bsdfs, P.emitters};
    MmltResult R;
    eval_mmlt(P, T, smp, (int) (i % (uint32_t) P.max_depth) + 1, mmlt_nx(P), R);
    lum_out[i] = R.splat.lum;
    if (P.boot_weighted) { DSplat w = R.splat; normalize_splat(w, P); lum_out[(size_t) n + i] = w.lum; } // luminance of f / importance
}

__global__ void __launch_bounds__(CHAIN_BLOCK) k_init_chains_mmlt(DParams P, const uint32_t *seed_index, const float *seed_lum) {
    const uint32_t lane = threadIdx.x;
    const uint32_t c = blockIdx.x * CHAIN_BLOCK + lane;
    if (c >= P.n_chains) return;
    const uint32_t idx = seed_index[c];
    const int depth = (int) (idx % (uint32_t) P.max_depth) + 1;
    MSampler smp;
    msampler_setup(smp, P, lane);
    smp.chain = P.boot_stream; smp.major = idx; smp.mode = SM_BOOT;
    const GlobalTables T{P.shade, P.bsdfs, P.emitters};
    MmltResult R;
    eval_mmlt(P, T, smp, depth, mmlt_nx(P), R);
    DSplat s = R.splat;
    // drmlt_proc.cpp:509-512: relative tolerance Epsilon. (The replay is inlined into another kernel than the
    // bootstrap, so contraction may differ in the last bit; the reference's check is relative as well.)
    if (!(fabsf((s.lum - seed_lum[c]) / seed_lum[c]) <= EPSILON_F)) atomicExch(P.error_flag, 1);
    normalize_splat(s, P);
    P.cur_lum[c] = s.lum; P.cur_px[c] = s.px; P.cur_py[c] = s.py;
    P.cur_r[c] = s.r; P.cur_g[c] = s.g; P.cur_b[c] = s.b;
    P.cur_t[c] = R.t;
    P.chain_depth[c] = depth;
    // The replayed stream, in call order: direct (1), sensor (n_s), emitter (n_e); fillReplay then tops up the
    // sensor state to D components, the emitter state to D, the direct state is complete (drmlt_proc.cpp:506-509)
    const uint32_t D = (uint32_t) mmlt_max_dim(depth), ns = R.n_sensor, ne = R.n_emitter;
    const uint32_t S = (uint32_t) P.mmlt_S, E = (uint32_t) P.mmlt_E;
    smp.b1_idx = 0xffffffffu;
    for (uint32_t k = 0; k < S; ++k) P.x[(size_t) k * P.n_chains + c] = smp.u_boot(k < ns ? 1u + k : 1u + ne + k);
    for (uint32_t k = 0; k < E; ++k) P.x[(size_t) (S + k) * P.n_chains + c] = smp.u_boot(k < ne ? 1u + ns + k : 1u + D + k);
    P.x[(size_t) (S + E) * P.n_chains + c] = smp.u_boot(0u);
}

// Two waves per SIMD is what this kernel lives on (7.3e8 -> 1.3e9 mutations/s at 131 072 chains): the launch bounds keep it
// at 256 registers whatever else the translation unit grows, and flat scenes run the build without any BVH code (FEAT 7).
// LDS_TABLES (round 4): the scene's shading / BSDF / emitter records staged in LDS behind the kernel's own rows, as the path kernels have
// them -- every walk step gathers two or three of them per lane, and from device memory each gather is a dependent round trip through the
// vector cache that the wave parks on (35 % of its time on config 5). Small scenes only (P.tables_in_lds and the rows still fit eight waves).
template <int FEAT, bool LDS_TABLES = false> __global__ void __launch_bounds__(CHAIN_BLOCK, 2) k_mutate_mmlt(DParams P, uint32_t n_mut, uint32_t mut_base) {
    const uint32_t lane = threadIdx.x;
    LdsTables LT;
    LT.shade_off = ((uint32_t) P.mmlt_S + (uint32_t) P.mmlt_E + 1u + 3u * ((uint32_t) P.max_depth + 3u)) * 64u;
    LT.bsdf_off = LT.shade_off + (uint32_t) P.n_shade * 16u;
    LT.emit_off = LT.bsdf_off + (uint32_t) P.n_bsdfs * 12u;
    if (LDS_TABLES) stage_tables(P, LT, lane);
    // Execution order: a wave's work is its DEEPEST chain's (every lane walks in lock step), so waves are made of chains of
    // one depth, deepest first. With more waves than the device holds (262 144 chains = two rounds) the slots that shallow
    // waves free early are taken by the next ones -- depths 6 5 4 | 3 2 1 pair up to equal sums -- and the kernel costs the
    // MEAN wave instead of the deepest. Chain ids (state, RNG streams) are untouched: the same chains, bit for bit.
    const uint32_t slot = blockIdx.x * CHAIN_BLOCK + lane;
    const uint32_t c = P.exec_order ? P.exec_order[slot] : slot;
    const bool live = c < P.n_chains;
    const uint32_t cc = live ? c : P.n_chains - 1;
    const uint32_t NX = mmlt_nx(P);
    for (uint32_t k = 0; k < NX; ++k) lds_x[k * 64u + lane] = P.x[(size_t) k * P.n_chains + cc];

    DSplat cur;
    cur.lum = P.cur_lum[cc]; cur.px = P.cur_px[cc]; cur.py = P.cur_py[cc];
    cur.r = P.cur_r[cc]; cur.g = P.cur_g[cc]; cur.b = P.cur_b[cc];
    int cur_t = P.cur_t[cc];
    const int depth = P.chain_depth[cc];
    const uint32_t usedS = 2u * (uint32_t) (depth + 1), usedE = 2u * (uint32_t) depth; // components a depth-`depth` path can consume

    MSampler smp;
    msampler_setup(smp, P, lane);
    smp.chain = P.chain_offset + cc;
    const GlobalTables T{P.shade, P.bsdfs, P.emitters};
    Counters ct = {0u, 0u, 0u, 0u, 0u};
    const bool amap = P.acceptance_map != 0;
    const bool mix = P.use_mixture != 0;

    // Chains run free: every pass of the loop evaluates ONE path per lane -- the first stage of the lane's next mutation, or the
    // second stage / Green's reverse move of the one it is in. (In a lockstep loop over mutations with the stages inside, the few
    // lanes in sixty-four that go to a second stage -- 8 % of the mutations -- cost the wave a whole second pass nearly every time.)
    uint32_t it = 0u, work = 0u; // work: path evaluations of this launch (the host groups chains of similar work into waves, drmlt_capi.cpp)
    int stage = 0;
    DSplat y, z;
    y.lum = 0.f; y.px = y.py = y.r = y.g = y.b = 0.f;
    z = y;
    int y_t = 0, z_t = 0;
    uint32_t ns1 = 0, ne1 = 0, ns2 = 0, ne2 = 0;
    float a1 = 0.f;
    for (;;) {
        const bool run = live && it < n_mut;
        if (!__builtin_amdgcn_ballot_w64(run)) break;
        if (!run) continue;
        const uint32_t m = mut_base + it;
        const u4 coins = philox4x32_10(P.key0, P.key1, 0u, m, smp.chain, TAG_COIN);
        const bool large = u32_to_unit(coins.x) < P.p_large;
        smp.major = m;
        smp.large = large;
        // fixEmitterPath: the emitter sampler moves in the second stage only for pure light tracing (drmlt_proc.cpp:566-573)
        smp.emitter_ident2 = P.fix_emitter_path != 0 && cur_t != 1;
        float a2 = 0.f;
        bool acc1 = false, acc2 = false, decided = true;
        {
            smp.mode = stage == 0 ? SM_STAGE1 : (stage == 1 ? SM_STAGE2 : SM_REVERSE);
            MmltResult R;
            if (LDS_TABLES) eval_mmlt<FEAT>(P, LT, smp, depth, NX, R);
            else eval_mmlt<FEAT>(P, T, smp, depth, NX, R);
            ++work;
            ct.rays += R.nrays;
            DSplat res = R.splat;
            normalize_splat(res, P);
            if (stage == 0) {
                bool doSecond = false;
                y = res; y_t = R.t; ns1 = R.n_sensor; ne1 = R.n_emitter;
                z.lum = 0.f; z.px = z.py = z.r = z.g = z.b = 0.f; z_t = 0; ns2 = ne2 = 0u;
                mh_first(mix, false, large, y.lum, cur.lum, u32_to_unit(coins.y), u32_to_unit(coins.w), a1, acc1, doSecond); // timidAfterLarge is refused for mmlt
                if (doSecond) { stage = 1; decided = false; }
            } else if (stage == 1) {
                z = res; z_t = R.t; ns2 = R.n_sensor; ne2 = R.n_emitter;
                if (mix) { // the second proposal replaces the first
                    a1 = 0.f;
                    mh_second_mixture(z.lum, cur.lum, u32_to_unit(coins.z), a2, acc2);
                } else if (lum_invalid(z.lum)) {
                } else if (P.type == 0) { // Green: the reverse move first
                    stage = 2; decided = false;
                } else if (P.type == 1) { // Tierney & Mira: product of the three samplers' ratios (drmlt_proc.cpp:633-637)
                    float ratio = 1.f;
                    if (!(fminf(1.f, y.lum / z.lum) >= 1.f)) { // (a large step never gets here: no second stage after it)
                        float num = 0.f, den = 0.f;
                        for (int sg = 0; sg < 2; ++sg) { // the direct sampler's first stage is the identity: ratio 1
                            const uint32_t nmax = sg == 0 ? max(ns1, ns2) : max(ne1, ne2);
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
                    mh_second_mira(y.lum, z.lum, cur.lum, a1, ratio, u32_to_unit(coins.z), a2, acc2);
                } else {
                    mh_second_orbital(y.lum, z.lum, cur.lum, u32_to_unit(coins.z), a2, acc2);
                }
            } else {
                ct.acc2b_rev += 1u << 16;
                mh_second_green(res.lum, z.lum, cur.lum, a1, u32_to_unit(coins.z), a2, acc2);
            }
        }
        if (!decided) continue;
        const bool doSecond = stage != 0;

        const MhWeights w = mh_weights(mix, amap, doSecond, a1, a2);
        if (w.w0 > 0.f) film_put(P, cur.px, cur.py, mk3(cur.r * w.w0, cur.g * w.w0, cur.b * w.w0));
        if (w.w1 > 0.f) film_put(P, y.px, y.py, mk3(y.r * w.w1, y.g * w.w1, y.b * w.w1));
        if (w.w2 > 0.f) film_put(P, z.px, z.py, mk3(z.r * w.w2, z.g * w.w2, z.b * w.w2));
        mh_count(ct, large, acc1, acc2, doSecond);

        if (acc1 || acc2) {
            // DRMLTSampler::accept on the three samplers: every component a path of this depth can consume
            for (int sg = 0; sg < 3; ++sg) {
                smp.select(sg);
                const uint32_t nk = sg == 0 ? usedS : (sg == 1 ? usedE : 1u);
                for (uint32_t k = 0; k < nk; ++k) {
                    const float v = wrap01(acc1 ? smp.y_raw(k) : smp.z_raw(k));
                    lds_x[(smp.x_off + k) * 64u + lane] = v;
                    if (smp.type == 2 && (k & 1u)) smp.pair_base = 0xffffffffu; // the pair cache holds the old x
                }
            }
            const int mark = mh_amap_mark(mix, amap, large, acc1, acc2);
            if (mark) film_put(P, cur.px, cur.py, mh_amap_colour(mark)); // at the state that is LEFT (device_mh.h)
            cur = select_splat(acc1, y, z);
            cur_t = acc1 ? y_t : z_t;
        }
        ++it;
        stage = 0;
    }

    if (live) {
        for (uint32_t k = 0; k < NX; ++k) P.x[(size_t) k * P.n_chains + c] = lds_x[k * 64u + lane];
        P.cur_lum[c] = cur.lum; P.cur_px[c] = cur.px; P.cur_py[c] = cur.py;
        P.cur_r[c] = cur.r; P.cur_g[c] = cur.g; P.cur_b[c] = cur.b;
        P.cur_t[c] = cur_t;
        if (P.chain_done) P.chain_done[c] = work; // (this kernel has no run-ahead: the array is free for the count)
    }
    flush_counters(P, ct, lane);
}

// u: [sensor S | emitter E | direct | depth] per point, dim >= S + E + 2
__global__ void __launch_bounds__(CHAIN_BLOCK) k_eval_paths_mmlt(DParams P, const float *u, uint32_t n, uint32_t dim, float *out8) {
    const uint32_t lane = threadIdx.x;
    const uint32_t i = blockIdx.x * CHAIN_BLOCK + lane;
    if (i >= n) return;
    MSampler smp;
    msampler_setup(smp, P, lane);
    smp.chain = 0u; smp.major = 0u; smp.mode = SM_ARRAY;
    smp.arr = u + (size_t) i * dim;
    const int depth = (int) smp.arr[P.mmlt_S + P.mmlt_E + 1];
    const GlobalTables T{P.shade, P.bsdfs, P.emitters};
    MmltResult R;
    R.s = R.t = 0;
    if (depth >= 1 && depth <= P.max_depth) eval_mmlt(P, T, smp, depth, mmlt_nx(P), R);
    else { R.splat.lum = R.splat.px = R.splat.py = R.splat.r = R.splat.g = R.splat.b = 0.f; R.nrays = R.n_sensor = R.n_emitter = R.n_direct = 0u; }
    float *o = out8 + (size_t) i * 8;
    o[0] = R.splat.lum; o[1] = R.splat.px; o[2] = R.splat.py; o[3] = R.splat.r; o[4] = R.splat.g; o[5] = R.splat.b;
    // n_dims carries the strategy as well: dims | s << 8 | t << 16
    o[6] = __int_as_float((int) (R.n_sensor + R.n_emitter + R.n_direct) | (R.s << 8) | (R.t << 16));
    o[7] = __int_as_float((int) R.nrays);
}

static size_t mmlt_lds_bytes(const DParams &P) {
    return ((size_t) P.mmlt_S + P.mmlt_E + 1 + 3 * ((size_t) P.max_depth + 3)) * 64 * sizeof(float);
}
void launch_bootstrap_mmlt(const DParams &P, uint32_t n, float *lum_out, hipStream_t st) {
    hipLaunchKernelGGL(k_bootstrap_mmlt, dim3((n + CHAIN_BLOCK - 1) / CHAIN_BLOCK), dim3(CHAIN_BLOCK), mmlt_lds_bytes(P), st, P, n, lum_out);
}
void launch_init_chains_mmlt(const DParams &P, const uint32_t *seed_index, const float *seed_lum, hipStream_t st) {
    hipLaunchKernelGGL(k_init_chains_mmlt, dim3((P.n_chains + CHAIN_BLOCK - 1) / CHAIN_BLOCK), dim3(CHAIN_BLOCK), mmlt_lds_bytes(P), st, P,
                       seed_index, seed_lum);
}
void launch_mutate_mmlt(const DParams &P, uint32_t n_mut, uint32_t mut_base, hipStream_t st) {
    const size_t table_bytes = ((size_t) P.n_shade * 16 + (size_t) P.n_bsdfs * 12 + (size_t) P.n_emitters * 8) * sizeof(float);
    static const bool no_lds_tables = getenv("DRMLT_MMLT_TABLES_GLOBAL") != nullptr; // A/B
    if (!P.use_bvh && P.tables_in_lds && !no_lds_tables && mmlt_lds_bytes(P) + table_bytes <= 20480)
        hipLaunchKernelGGL((k_mutate_mmlt<7, true>), dim3((P.n_chains + CHAIN_BLOCK - 1) / CHAIN_BLOCK), dim3(CHAIN_BLOCK), mmlt_lds_bytes(P) + table_bytes, st, P, n_mut, mut_base);
    else if (P.use_bvh) hipLaunchKernelGGL(k_mutate_mmlt<15>, dim3((P.n_chains + CHAIN_BLOCK - 1) / CHAIN_BLOCK), dim3(CHAIN_BLOCK), mmlt_lds_bytes(P), st, P, n_mut, mut_base);
    else hipLaunchKernelGGL(k_mutate_mmlt<7>, dim3((P.n_chains + CHAIN_BLOCK - 1) / CHAIN_BLOCK), dim3(CHAIN_BLOCK), mmlt_lds_bytes(P), st, P, n_mut,
                       mut_base);
}
void launch_eval_paths_mmlt(const DParams &P, const float *u, uint32_t n, uint32_t dim, float *out8, hipStream_t st) {
    hipLaunchKernelGGL(k_eval_paths_mmlt, dim3((n + CHAIN_BLOCK - 1) / CHAIN_BLOCK), dim3(CHAIN_BLOCK), mmlt_lds_bytes(P), st, P, u, n, dim, out8);
}


// ------------------------------------------------------------------------------------------ regrouping on the device
// The execution order of the bidirectional chain kernels between the launches of a call (drmlt_capi.cpp: regroup): chains sorted
// by (path depth, deepest first; evaluations of the launch just done, most first), stable in the chain id -- the counting sort the
// host did in round 3 behind a device-to-host copy, two stream synchronisations and a host-to-device copy per launch (VERDICT r03
// #14). Three small kernels on the chains' stream, nothing waits for the host:
//   k_regroup_hist     one wave per segment of RG_SEG chains: key of every chain, histogram of the segment (LDS atomics) -> hist[key][seg]
//   k_regroup_scan     one wave: hist[key][seg] := number of chains that come BEFORE the first chain of that key in that segment
//   k_regroup_scatter  one wave per segment, its chains in id order, 64 at a time; lanes with equal keys are numbered by ballot +
//                      prefix count, so equal keys keep their id order: the same permutation as the host's stable sort.
#define RG_SEG 1024u
#define RG_BUCKETS 16u
DEV uint32_t regroup_key(const uint32_t *work, const int32_t *depth, uint32_t j, uint32_t n_mut, uint32_t md) {
    const uint32_t d = depth ? (uint32_t) (depth[j] - 1) : 0u;                       // (seed index % maxDepth: pathsampler.cpp:889)
    const uint32_t w = work[j], extra = w > n_mut ? w - n_mut : 0u;                   // second stages and reverse moves
    const uint32_t b = min(RG_BUCKETS - 1u, (uint32_t) ((unsigned long long) extra * RG_BUCKETS / max(1u, n_mut)));
    return (md - 1u - min(d, md - 1u)) * RG_BUCKETS + (RG_BUCKETS - 1u - b);
}
__global__ void __launch_bounds__(64) k_regroup_hist(const uint32_t *work, const int32_t *depth, uint32_t n, uint32_t n_mut, uint32_t md, uint32_t nseg, uint32_t *hist) {
    __shared__ uint32_t cnt[24u * RG_BUCKETS];
    const uint32_t lane = threadIdx.x, seg = blockIdx.x, nkeys = md * RG_BUCKETS;
    for (uint32_t k = lane; k < nkeys; k += 64u) cnt[k] = 0u;
    __syncthreads();
    for (uint32_t i = 0; i < RG_SEG; i += 64u) {
        const uint32_t j = seg * RG_SEG + i + lane;
        if (j < n) atomicAdd(&cnt[regroup_key(work, depth, j, n_mut, md)], 1u);
    }
    __syncthreads();
    for (uint32_t k = lane; k < nkeys; k += 64u) hist[(size_t) k * nseg + seg] = cnt[k];
}
__global__ void __launch_bounds__(64) k_regroup_scan(uint32_t nkeys, uint32_t nseg, uint32_t *hist) {
    const uint32_t lane = threadIdx.x;
    uint32_t running = 0u; // wave-uniform: chains before the current (key, first segment of this pass)
    for (uint32_t k = 0; k < nkeys; ++k)
        for (uint32_t s0 = 0; s0 < nseg; s0 += 64u) {
            const uint32_t s = s0 + lane;
            const uint32_t v = s < nseg ? hist[(size_t) k * nseg + s] : 0u;
            uint32_t incl = v; // inclusive prefix sum over the wave
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = (uint32_t) __shfl_up((int) incl, off, 64);
                if ((int) lane >= off) incl += up;
            }
            if (s < nseg) hist[(size_t) k * nseg + s] = running + incl - v;
            running += (uint32_t) __shfl((int) incl, 63, 64);
        }
}
__global__ void __launch_bounds__(64) k_regroup_scatter(const uint32_t *work, const int32_t *depth, uint32_t n, uint32_t n_mut, uint32_t md, uint32_t nseg, const uint32_t *offs,
                                                        uint32_t *order, uint32_t padded) {
    __shared__ uint32_t cur[24u * RG_BUCKETS];
    const uint32_t lane = threadIdx.x, seg = blockIdx.x, nkeys = md * RG_BUCKETS;
    for (uint32_t k = lane; k < nkeys; k += 64u) cur[k] = offs[(size_t) k * nseg + seg];
    __syncthreads();
    for (uint32_t i = 0; i < RG_SEG; i += 64u) {
        const uint32_t j = seg * RG_SEG + i + lane;
        const bool have = j < n;
        const uint32_t key = have ? regroup_key(work, depth, j, n_mut, md) : 0xffffffffu;
        unsigned long long todo = __ballot(have);
        while (todo) { // one distinct key per round, lowest pending lane first: every lane of the wave reaches the end
            const uint32_t k = (uint32_t) __shfl((int) key, __builtin_ctzll(todo), 64);
            const unsigned long long m = __ballot(have && key == k);
            if (have && key == k) order[cur[k] + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u))] = j;
            __syncthreads();
            if (lane == 0) cur[k] += (uint32_t) __popcll(m);
            __syncthreads();
            todo &= ~m;
        }
    }
    if (seg == 0) for (uint32_t j = n + lane; j < padded; j += 64u) order[j] = n; // whole waves: the tail runs no chain
}
void launch_regroup(const uint32_t *work, const int32_t *depth_or_null, uint32_t n, uint32_t n_mut, uint32_t md, uint32_t *order, uint32_t padded, uint32_t *scratch, hipStream_t st) {
    const uint32_t nseg = (n + RG_SEG - 1u) / RG_SEG, nkeys = md * RG_BUCKETS;
    hipLaunchKernelGGL(k_regroup_hist, dim3(nseg), dim3(64), 0, st, work, depth_or_null, n, n_mut, md, nseg, scratch);
    hipLaunchKernelGGL(k_regroup_scan, dim3(1), dim3(64), 0, st, nkeys, nseg, scratch);
    hipLaunchKernelGGL(k_regroup_scatter, dim3(nseg), dim3(64), 0, st, work, depth_or_null, n, n_mut, md, nseg, scratch, order, padded);
}
size_t regroup_scratch_words(uint32_t n, uint32_t md) { return (size_t) md * RG_BUCKETS * ((n + RG_SEG - 1u) / RG_SEG); }
