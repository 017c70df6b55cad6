// Device code of the DRMLT hot path: PSS samplers (transition kernels evaluated lazily as
// pure functions of the addressed RNG), ray queries, and the unidirectional estimator as a
// resumable state machine with exactly one ray query per step.
//
// Reference behaviour restated here (paths relative to the reference checkout):
//   transition kernels       src/integrators/drmlt/tools/transition.h:54-190
//   fillSpace / wrap         src/integrators/drmlt/drmlt_sampler.cpp:313-394, drmlt_sampler.h:140-144
//   sampleSplats (path)      src/libbidir/pathsampler.cpp:529-567
//   MIPathTracer::Li         src/integrators/path/path.cpp:123-321
//   ray epsilons             src/librender/skdtree.cpp:125-129,213-218, scene.cpp:891-893
//   emitter sampling         src/librender/scene.cpp:879-904, shape.cpp:102-127, area.cpp:111-189
//   diffuse / dielectric     src/bsdfs/diffuse.cpp:110-149, dielectric.cpp:228-333
//   perspective sensor       src/sensors/perspective.cpp:271-300
#pragma once
#include "device_bsdf.h"
#include "device_math.h"
#include "device_types.h"

// Current PSS states of the 64 chains of this wave, [dim][lane]: lane-contiguous rows, so any
// per-lane dimension pattern is bank-conflict free (64 == 0 mod 32 banks). Addressed directly
// (ds_read), never through a generic pointer: LDS offset 0 casts to the flat null pointer.
extern __shared__ __attribute__((aligned(16))) float lds_x[];

// The proposal arithmetic is evaluated at several sites (on demand in k_mutate / k_mutate_v2 / v3, for whole rows in
// k_mutate_v4, pair-wise at a commit): without this the compiler contracts a*b+c into an fma at some sites and not at
// others, and the kernels' chains drift apart in the last bit. Explicit fmaf() calls stay fused everywhere.
#define FP_STRICT _Pragma("clang fp contract(off)")

// ------------------------------------------------------------------ PSS sampler
enum { SM_BOOT = 0, SM_ARRAY = 1, SM_STAGE1 = 2, SM_STAGE2 = 3, SM_REVERSE = 4, SM_PT = 5 };

// drmlt_sampler.h:140-144: y > 1 ? 2 - y : (y <= 0 ? |y| : y). For 0 < y <= 1, |y| = y, so two selects collapse into
// one select on an |.|-modified operand (the nested form compiled to two exec-mask branches per component).
DEV float wrap01(float y) { return y > 1.f ? 2.f - y : fabsf(y); }

#define KELEMEN_S1 (1.0f / 1024.0f)
#define KELEMEN_S2 (1.0f / 64.0f)
#define ORBITAL_SCALE 1.9f
#define LOG2_S1_OVER_S2 (-4.0f) // log2((1/1024)/(1/64))
// wrapped Cauchy: rho = exp(-1/4), dispersion c = 2 rho / (1 + rho^2)
#define WC_DISPERSION 0.96954361f

// Kelemen kernel (transition.h:97-111): sign * s2 * (s1/s2)^(1 - xi')
DEV float kelemen_sample(float xi, float s2) {
        FP_STRICT;
    float sign = 1.f;
    if (xi < 0.5f) { xi *= 2.f; } else { sign = -1.f; xi = 2.f * (xi - 0.5f); }
    return sign * s2 * fast_exp2((1.f - xi) * LOG2_S1_OVER_S2);
}
// Gaussian kernel (transition.h:61-66), Box-Muller cosine branch
DEV float gaussian_sample(float u1, float u2, float sigma) {
        FP_STRICT;
    float tmp = sqrtf(-1.3862943611198906f * fast_log2(1.f - u1)); // -2 ln(v) = -2 ln2 log2(v)
    return tmp * cos_rev(u2) * sigma;
}
// log pdf of the (unscaled) Kelemen kernel, transition.h:113-122
DEV float kelemen_logpdf(float du) {
    float d = fabsf(du);
    if (d < KELEMEN_S1 || d > KELEMEN_S2) return -INFINITY;
    return -__logf(2.f * d * 2.772588722239781f); // ln(s2/s1) = ln 16
}

// STRIDE: floats per row of the chain-state rows in LDS (x[k] of the chain in column `lane` = lds_x[k * STRIDE + lane])
template <uint32_t STRIDE = 64u> struct SamplerT {
    // addressing
    uint32_t key0, key1, chain, major; // major: mutation index (chains) or sample index (boot / pt)
    int mode, type;
    bool large;
    float sigma2;
    uint32_t lane;    // chain state lives in LDS: x[k] = lds_x[k * 64 + lane]
    const float *arr; // SM_ARRAY: explicit PSS vector
    // one-block caches
    u4 b1, b2;
    uint32_t b1_idx, b2_idx;
    // orbital pair cache
    uint32_t pair_base;
    float pair_y0, pair_y1, pair_z0, pair_z1;
    bool pair_has_z;

    DEV void reset_caches() { b1_idx = b2_idx = 0xffffffffu; pair_base = 0xffffffffu; }

    DEV float u_boot(uint32_t k, uint32_t tag) {
        uint32_t blk = k >> 2;
        if (blk != b1_idx) { b1 = philox4x32_10(key0, key1, blk, major, chain, tag); b1_idx = blk; }
        return pick4(b1, k & 3u);
    }
    DEV float u_s1(uint32_t idx) {
        uint32_t blk = idx >> 2;
        if (blk != b1_idx) { b1 = philox4x32_10(key0, key1, blk, major, chain, TAG_S1); b1_idx = blk; }
        return pick4(b1, idx & 3u);
    }
    DEV float u_s2(uint32_t idx) {
        uint32_t blk = idx >> 2;
        if (blk != b2_idx) { b2 = philox4x32_10(key0, key1, blk, major, chain, TAG_S2); b2_idx = blk; }
        return pick4(b2, idx & 3u);
    }
    DEV float x(uint32_t k) const { return lds_x[k * STRIDE + lane]; }

    // first-stage proposal, unwrapped (fillSpace with isFirst = true)
    DEV float y_raw(uint32_t k) {
        FP_STRICT;
        if (large) return u_s1(k);
        if (type != 2 /*orbital*/) return x(k) + kelemen_sample(u_s1(k), KELEMEN_S2);
        ensure_pair(k & ~1u, false);
        return (k & 1u) ? pair_y1 : pair_y0;
    }
    // second-stage proposal, unwrapped (fillSpace with isFirst = false)
    DEV float z_raw(uint32_t k) {
        FP_STRICT;
        // second stage of a large step (timidAfterLarge): the reference's fillSpace takes its
        // uniform branch again (drmlt_sampler.cpp:319-321 behind a debug-only assertion)
        if (large) return u_s2(k);
        if (type != 2) return x(k) + gaussian_sample(u_s2(2u * k), u_s2(2u * k + 1u), sigma2);
        ensure_pair(k & ~1u, true);
        return (k & 1u) ? pair_z1 : pair_z0;
    }
    // orbital pair (k0, k0+1): y = x + d (cos a, sin a); z = y + R(theta) (x - y)
    DEV void ensure_pair(uint32_t k0, bool need_z) {
        FP_STRICT;
        if (pair_base != k0) {
            pair_base = k0;
            float x0 = x(k0), x1 = x(k0 + 1u);
            if (large) {
                pair_y0 = u_s1(k0);
                pair_y1 = u_s1(k0 + 1u);
            } else {
                float d = kelemen_sample(u_s1(k0), KELEMEN_S2 * ORBITAL_SCALE);
                float a = u_s1(k0 + 1u);
                pair_y0 = fmaf(d, cos_rev(a), x0);
                pair_y1 = fmaf(d, cos_rev(a - 0.25f), x1); // sin(2 pi a) as the row samplers evaluate it (v_sin and v_cos differ in the last bit)
            }
            pair_has_z = false;
        }
        if (need_z && !pair_has_z) {
            pair_has_z = true;
            float x0 = x(k0), x1 = x(k0 + 1u);
            // theta ~ wrapped Cauchy by inverse CDF (transition.h:157-173): cos(theta) = A
            float xi = u_s2(k0 >> 1);
            float sign = 1.f;
            if (xi < 0.5f) { xi *= 2.f; } else { sign = -1.f; xi = 2.f * (xi - 0.5f); }
            float V = cos_rev(xi);
            float A = fminf(1.f, fmaxf(-1.f, (V + WC_DISPERSION) / (1.f + WC_DISPERSION * V)));
            float ct = A, st = sign * sqrtf(fmaxf(0.f, 1.f - A * A));
            // (x - y) rotated by theta about y: identical to y + |x-y| (cos, sin)(theta + mu),
            // mu the polar angle of x - y (drmlt_sampler.cpp:374-391), without the acos round trip
            float dx0 = x0 - pair_y0, dx1 = x1 - pair_y1;
            pair_z0 = pair_y0 + (ct * dx0 - st * dx1);
            pair_z1 = pair_y1 + (st * dx0 + ct * dx1);
        }
    }

    // value handed to the path code for PSS dimension k (primarySample)
    DEV float next(uint32_t k) {
        FP_STRICT;
        switch (mode) {
            case SM_BOOT: return u_boot(k, TAG_BOOT);
            case SM_PT: return u_boot(k, TAG_PT);
            case SM_ARRAY: return arr[k];
            case SM_STAGE1: return wrap01(y_raw(k));
            case SM_STAGE2: return wrap01(z_raw(k));
            default: { // Green reverse: y* = z - (y - x)
                float du = y_raw(k) - x(k);
                return wrap01(z_raw(k) - du);
            }
        }
    }
};
typedef SamplerT<64u> Sampler;

// ------------------------------------------------------------------ PSS sampler of k_mutate_v2
// Same proposals as `Sampler`, but the Philox draws of a mutation are produced up front, by all
// lanes that start an evaluation together (convergent code in the bookkeeping branch), and parked
// in LDS next to the chain state:
//   lds_x  [0 .. D)          current state x
//   lds_u1 = lds_x + D*64    first-stage uniforms, draw k of the TAG_S1 stream (row k, D rounded to 4)
//   lds_s2 = lds_u1 + D4*64  second-stage values: orbital: theta uniform of pair q in row q;
//                            iid: the Gaussian perturbation of dim k in row k; large: uniform k
// The per-dimension code below is then pure arithmetic on LDS operands -- no RNG, no caches, no
// divergent Philox in the ray loop (measured: the on-demand sampler was ~2/3 of the VALU work).
struct LdsSampler {
    uint32_t key0, key1, chain, major;
    int mode, type;
    bool large;
    float sigma2;
    uint32_t lane;
    uint32_t u1_off, s2_off; // row offsets (in floats) of lds_u1 / lds_s2
    uint32_t stride;         // floats per row (64; 32 in the half-wave experiment)
    bool timing_probe;       // DRMLT_DEBUG bit 256: replace stage-1 Philox by a trivial hash (timing experiments only)

    DEV void reset_caches() {}
    DEV float x(uint32_t k) const { return lds_x[k * stride + lane]; }
    DEV float u1(uint32_t k) const { return lds_x[u1_off + k * stride + lane]; }
    DEV float s2(uint32_t k) const { return lds_x[s2_off + k * stride + lane]; }

    // first-stage draws of this mutation, Philox blocks [b0, b1): rows 4b .. 4b+3 of lds_u1
    DEV void fill_stage1(uint32_t b0, uint32_t b1) {
        for (uint32_t b = b0; b < b1; ++b) {
            u4 r = timing_probe ? u4{b * 2654435761u ^ major, chain * 40503u + b, major * 2246822519u, b + chain}
                                : philox4x32_10(key0, key1, b, major, chain, TAG_S1);
            float *dst = &lds_x[u1_off + b * 4u * stride + lane];
            dst[0] = u32_to_unit(r.x); dst[stride] = u32_to_unit(r.y); dst[2u * stride] = u32_to_unit(r.z); dst[3u * stride] = u32_to_unit(r.w);
        }
    }
    // second-stage values (see layout above); blocks first, first + step, ... (step 2 = shared by two lanes)
    DEV void fill_stage2(uint32_t D4, uint32_t first, uint32_t step) {
        FP_STRICT;
        if (large || type == 2) {
            const uint32_t rows = large ? D4 : (D4 / 2u + 3u) & ~3u; // uniforms: one per dim, or one per pair
            for (uint32_t b = first; b < rows / 4u; b += step) {
                u4 r = philox4x32_10(key0, key1, b, major, chain, TAG_S2);
                float *dst = &lds_x[s2_off + b * 4u * stride + lane];
                dst[0] = u32_to_unit(r.x); dst[stride] = u32_to_unit(r.y); dst[2u * stride] = u32_to_unit(r.z); dst[3u * stride] = u32_to_unit(r.w);
            }
        } else {
            for (uint32_t b = first; b < D4 / 2u; b += step) { // draws (2k, 2k+1) -> Gaussian sample of dim k
                u4 r = philox4x32_10(key0, key1, b, major, chain, TAG_S2);
                float *dst = &lds_x[s2_off + b * 2u * stride + lane];
                dst[0] = gaussian_sample(u32_to_unit(r.x), u32_to_unit(r.y), sigma2);
                dst[stride] = gaussian_sample(u32_to_unit(r.z), u32_to_unit(r.w), sigma2);
            }
        }
    }

    DEV float y_raw(uint32_t k) const {
        FP_STRICT;
        if (large) return u1(k);
        if (type != 2) return x(k) + kelemen_sample(u1(k), KELEMEN_S2);
        const uint32_t k0 = k & ~1u;
        float d = kelemen_sample(u1(k0), KELEMEN_S2 * ORBITAL_SCALE), a = u1(k0 + 1u);
        return fmaf(d, cos_rev(a - ((k & 1u) ? 0.25f : 0.f)), x(k));
    }
    DEV float z_raw(uint32_t k) const {
        FP_STRICT;
        if (large) return s2(k);
        if (type != 2) return x(k) + s2(k);
        const uint32_t k0 = k & ~1u;
        float x0 = x(k0), x1 = x(k0 + 1u);
        float d = kelemen_sample(u1(k0), KELEMEN_S2 * ORBITAL_SCALE), a = u1(k0 + 1u);
        float y0 = fmaf(d, cos_rev(a), x0), y1 = fmaf(d, cos_rev(a - 0.25f), x1);
        float xi = s2(k0 >> 1);
        float sign = 1.f;
        if (xi < 0.5f) { xi *= 2.f; } else { sign = -1.f; xi = 2.f * (xi - 0.5f); }
        float V = cos_rev(xi);
        float A = fminf(1.f, fmaxf(-1.f, (V + WC_DISPERSION) / (1.f + WC_DISPERSION * V)));
        float ct = A, st = sign * sqrtf(fmaxf(0.f, 1.f - A * A));
        float dx0 = x0 - y0, dx1 = x1 - y1;
        return (k & 1u) ? y1 + (st * dx0 + ct * dx1) : y0 + (ct * dx0 - st * dx1);
    }
    DEV float next(uint32_t k) const {
        FP_STRICT;
        if (type == 2 && mode == SM_STAGE1) {
            // The common case (orbital, first stage) without divergent branches: unconditional LDS reads, the
            // large-step case as a select, sin(2 pi a) as cos(2 pi (a - 1/4)) so that one v_cos serves both components.
            const uint32_t k0 = k & ~1u;
            const bool odd = (k & 1u) != 0u;
            const float ua = u1(k0), ub = u1(k0 + 1u);
            const float d = kelemen_sample(ua, KELEMEN_S2 * ORBITAL_SCALE);
            const float y = fmaf(d, cos_rev(ub - (odd ? 0.25f : 0.f)), x(k));
            return wrap01(large ? (odd ? ub : ua) : y);
        }
        if (type == 2) {
            // orbital second stage (the reverse mode belongs to Green and never gets here): the pair once, the
            // large-step case (timidAfterLarge) as a select on an unconditional read
            float z0, z1;
            orbital_pair(k & ~1u, true, z0, z1);
            const float zl = s2(k);
            return wrap01(large ? zl : ((k & 1u) ? z1 : z0));
        }
        // iid kernels (Green, Mira): all three modes from unconditional reads and selects
        const float xk = x(k), a = u1(k), g = s2(k);
        const float y = large ? a : xk + kelemen_sample(a, KELEMEN_S2);
        const float z = large ? g : xk + g;
        const float v = mode == SM_STAGE1 ? y : (mode == SM_STAGE2 ? z : z - (y - xk)); // reverse: y* = z - (y - x)
        return wrap01(v);
    }
    // Orbital pair (k0, k0 + 1), both components of the first- or second-stage proposal at once (accept(): the
    // per-component form above evaluates the shared radius / angle / rotation twice per pair). Same arithmetic.
    DEV void orbital_pair(uint32_t k0, bool second, float &o0, float &o1) const {
        FP_STRICT;
        const float x0 = x(k0), x1 = x(k0 + 1u);
        const float d = kelemen_sample(u1(k0), KELEMEN_S2 * ORBITAL_SCALE), a = u1(k0 + 1u);
        const float y0 = fmaf(d, cos_rev(a), x0), y1 = fmaf(d, cos_rev(a - 0.25f), x1);
        if (!second) { o0 = y0; o1 = y1; return; }
        float xi = s2(k0 >> 1);
        float sign = 1.f;
        if (xi < 0.5f) { xi *= 2.f; } else { sign = -1.f; xi = 2.f * (xi - 0.5f); }
        const float V = cos_rev(xi);
        const float A = fminf(1.f, fmaxf(-1.f, (V + WC_DISPERSION) / (1.f + WC_DISPERSION * V)));
        const float ct = A, st = sign * sqrtf(fmaxf(0.f, 1.f - A * A));
        const float dx0 = x0 - y0, dx1 = x1 - y1;
        o0 = y0 + (ct * dx0 - st * dx1);
        o1 = y1 + (st * dx0 + ct * dx1);
    }
};

// ------------------------------------------------------------------ PSS sampler of k_mutate_v4
// The proposals themselves are parked in LDS, not the uniforms they are made of: the bookkeeping branch of k_mutate_v4
// evaluates the transition kernels for ALL dimensions of the chains that start an evaluation, flattened over the 64
// lanes of the wave (items = chain x Philox block, every lane busy), so that inside the divergent path step a PSS
// component is one LDS read and a reflection. Rows (stride `stride` floats, column = chain):
//   x  [0 .. D)    current state
//   y  [D4 rows]   first-stage proposal, unwrapped (a large step: the uniforms themselves)
//   z  [D4 rows]   second-stage proposal, unwrapped
// Same arithmetic per component as LdsSampler (and Sampler): the chains are bit-identical.
struct RowSampler {
    static constexpr bool batch_draws = true; // path_step: the (up to) five components of a step are read together, ahead of the hit's digestion
    uint32_t key0, key1;
    int mode, type;
    float sigma2;
    uint32_t lane, stride, y_off, z_off;

    DEV void reset_caches() {}
    DEV float x(uint32_t k) const { return lds_x[k * stride + lane]; }
    DEV float y_raw(uint32_t k) const { return lds_x[y_off + k * stride + lane]; }
    DEV float z_raw(uint32_t k) const { return lds_x[z_off + k * stride + lane]; }
    DEV float next(uint32_t k) const {
        FP_STRICT;
        float v = y_raw(k);
        if (mode != SM_STAGE1) {
            const float z = z_raw(k);
            v = mode == SM_STAGE2 ? z : z - (v - x(k)); // Green's reverse move: y* = z - (y - x)
        }
        return wrap01(v);
    }
    // first-stage proposal of chain column `col`, dimensions 4b .. 4b+3, from Philox block b of (major, chain)
    DEV void fill_first(uint32_t col, uint32_t b, uint32_t major, uint32_t chain, bool large) const {
        FP_STRICT;
        const u4 r = philox4x32_10(key0, key1, b, major, chain, TAG_S1);
        const float u0 = u32_to_unit(r.x), u1 = u32_to_unit(r.y), u2 = u32_to_unit(r.z), u3 = u32_to_unit(r.w);
        const float *xs = &lds_x[4u * b * stride + col];
        float *ys = &lds_x[y_off + 4u * b * stride + col];
        const float x0 = xs[0], x1 = xs[stride], x2 = xs[2u * stride], x3 = xs[3u * stride];
        float y0, y1, y2, y3;
        if (type == 2) { // pairwise orbital: radius from the Kelemen kernel (x 1.9), uniform angle (drmlt_sampler.cpp:354-361)
            const float d0 = kelemen_sample(u0, KELEMEN_S2 * ORBITAL_SCALE), d1 = kelemen_sample(u2, KELEMEN_S2 * ORBITAL_SCALE);
            y0 = fmaf(d0, cos_rev(u1), x0); y1 = fmaf(d0, cos_rev(u1 - 0.25f), x1);
            y2 = fmaf(d1, cos_rev(u3), x2); y3 = fmaf(d1, cos_rev(u3 - 0.25f), x3);
        } else {
            y0 = x0 + kelemen_sample(u0, KELEMEN_S2); y1 = x1 + kelemen_sample(u1, KELEMEN_S2);
            y2 = x2 + kelemen_sample(u2, KELEMEN_S2); y3 = x3 + kelemen_sample(u3, KELEMEN_S2);
        }
        ys[0] = large ? u0 : y0; ys[stride] = large ? u1 : y1; ys[2u * stride] = large ? u2 : y2; ys[3u * stride] = large ? u3 : y3;
    }
    // second-stage proposal of chain column `col` from Philox block b of the TAG_S2 stream: a large step (timidAfterLarge)
    // -> dims 4b..4b+3 (uniforms); orbital -> the angles of pairs 4b..4b+3 = dims 8b..8b+7; iid kernels -> the Gaussian
    // perturbations of dims 2b, 2b+1 (draws 2k, 2k+1 belong to dim k). Blocks beyond the chain's kind of stage do nothing.
    DEV void fill_second(uint32_t col, uint32_t b, uint32_t D4, uint32_t major, uint32_t chain, bool large) const {
        FP_STRICT;
        const uint32_t nblk = large ? D4 / 4u : (type == 2 ? (D4 / 2u + 3u) / 4u : D4 / 2u);
        if (b >= nblk) return;
        const u4 r = philox4x32_10(key0, key1, b, major, chain, TAG_S2);
        const float u[4] = {u32_to_unit(r.x), u32_to_unit(r.y), u32_to_unit(r.z), u32_to_unit(r.w)};
        if (large) {
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) lds_x[z_off + (4u * b + i) * stride + col] = u[i];
        } else if (type == 2) {
            // (all reads first: the writes below are to LDS too and may alias them for the compiler -- pair after pair would wait for its
            // own reads behind the previous pair's writes)
            float xa[4], xb[4], ya[4], yb[4];
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                const uint32_t k0 = 2u * (4u * b + i), kk = k0 + 1u < D4 ? k0 : 0u;
                xa[i] = lds_x[kk * stride + col]; xb[i] = lds_x[(kk + 1u) * stride + col];
                ya[i] = lds_x[y_off + kk * stride + col]; yb[i] = lds_x[y_off + (kk + 1u) * stride + col];
            }
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                const uint32_t k0 = 2u * (4u * b + i);
                if (k0 + 1u < D4) {
                    const float x0 = xa[i], x1 = xb[i], y0 = ya[i], y1 = yb[i];
                    // theta ~ wrapped Cauchy by inverse CDF (transition.h:157-173); z = y + R(theta)(x - y) (drmlt_sampler.cpp:374-391)
                    float xi = u[i], sign = 1.f;
                    if (xi < 0.5f) { xi *= 2.f; } else { sign = -1.f; xi = 2.f * (xi - 0.5f); }
                    const float V = cos_rev(xi);
                    const float A = fminf(1.f, fmaxf(-1.f, (V + WC_DISPERSION) / (1.f + WC_DISPERSION * V)));
                    const float ct = A, st = sign * sqrtf(fmaxf(0.f, 1.f - A * A));
                    const float dx0 = x0 - y0, dx1 = x1 - y1;
                    lds_x[z_off + k0 * stride + col] = y0 + (ct * dx0 - st * dx1);
                    lds_x[z_off + (k0 + 1u) * stride + col] = y1 + (st * dx0 + ct * dx1);
                }
            }
        } else {
            const uint32_t k = 2u * b;
            const float x0 = lds_x[k * stride + col], x1 = lds_x[(k + 1u) * stride + col];
            lds_x[z_off + k * stride + col] = x0 + gaussian_sample(u[0], u[1], sigma2);
            lds_x[z_off + (k + 1u) * stride + col] = x1 + gaussian_sample(u[2], u[3], sigma2);
        }
    }
};

// ------------------------------------------------------------------ ray queries
struct Hit {
    int prim;
    float t, u, v;
};

// Records fetched per lane from device memory, read through a GLOBAL-address-space pointer whatever the origin of `p`:
// a pointer that passed through a copied parameter block (k_mutate_v4's per-section copies) is generic to the compiler,
// which then emits FLAT loads -- both wait counters, the LDS aperture check. T: 16-byte multiple in a 16-byte aligned array.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <class T> DEV T load_global16(const T *p) {
    static_assert(sizeof(T) % 16 == 0, "records are read in 16-byte words");
    typedef const u32x4_t __attribute__((address_space(1))) *GQ;
    const GQ q = (GQ) (uintptr_t) p;
    union { T v; u32x4_t w[sizeof(T) / 16]; } u;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16u; ++i) u.w[i] = q[i];
    return u.v;
}
// Samplers whose components are plain reads (proposal rows in LDS or in device memory) have the draws of a path step requested
// together: a loop over `need` reads, each waited for, was `need` LDS round trips per step even with the rows in LDS (round 4: config 2
// 2.11e9 -> 2.24e9, the 2000-triangle soup 5.82e8 -> 6.01e8; with the rows in device memory it is the difference between one and five
// trips through the L2). The lazy samplers (Philox inside next()) keep the loop: one copy of their code.
template <class S, class = void> struct draws_batched : std::false_type {};
template <class S> struct draws_batched<S, std::void_t<decltype(S::batch_draws)>> : std::integral_constant<bool, S::batch_draws> {};
DEV float load_global_f32(const float *p) { return *(const float __attribute__((address_space(1))) *) (uintptr_t) p; }
DEV void atomic_add_global_f32(float *p, float v) { // no-return float add on device memory (global_atomic_add_f32)
    (void) __hip_atomic_fetch_add((float __attribute__((address_space(1))) *) (uintptr_t) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One primitive against one ray. `P` is wave-uniform in the brute-force loop (SGPR operands).
// Flat primitives are branch-free (selects only): no exec-mask traffic in the hot loop.
template <int FEAT = 15> DEV void intersect_prim(const DPrim &P, int idx, f3 o, f3 d, float tmin, Hit &h) {
    f3 lo = mk3(fmaf(P.m[0], o.x, fmaf(P.m[1], o.y, fmaf(P.m[2], o.z, P.m[3]))),
                fmaf(P.m[4], o.x, fmaf(P.m[5], o.y, fmaf(P.m[6], o.z, P.m[7]))),
                fmaf(P.m[8], o.x, fmaf(P.m[9], o.y, fmaf(P.m[10], o.z, P.m[11]))));
    f3 ld = mk3(fmaf(P.m[0], d.x, fmaf(P.m[1], d.y, P.m[2] * d.z)), fmaf(P.m[4], d.x, fmaf(P.m[5], d.y, P.m[6] * d.z)),
                fmaf(P.m[8], d.x, fmaf(P.m[9], d.y, P.m[10] * d.z)));
    if (!(FEAT & 4) || P.type != PRIM_SPHERE) { // wave-uniform branch (compiled out for sphere-free scenes)
        float t = -lo.z * fast_rcp(ld.z);
        float u = fmaf(t, ld.x, lo.x), v = fmaf(t, ld.y, lo.y);
        // triangle: u, v, 1-u-v >= 0; parallelogram (rectangle or merged triangle pair): u, v, 1-u, 1-v >= 0.
        // The kind is wave-uniform (SGPR), so this is selects on a scalar condition, no exec-mask traffic.
        const bool tri = P.type == PRIM_TRIANGLE;
        const float w = tri ? 1.f - (u + v) : fminf(1.f - u, 1.f - v);
        bool hit = fminf(fminf(u, v), w) >= 0.f && t >= tmin && t <= h.t;
        int id = idx;
        if (P.type == PRIM_QUAD2) { // sub-triangle (a,b,c) for v <= u, (a,c,d) otherwise; its own barycentrics
            const bool second = v > u;
            id = idx + (second ? 1 : 0);
            const float uu = second ? u : u - v, vv = second ? v - u : v;
            u = uu; v = vv;
        }
        h.prim = hit ? id : h.prim;
        h.t = hit ? t : h.t;
        h.u = hit ? u : h.u;
        h.v = hit ? v : h.v;
    } else {
        // unit sphere; discriminant from the closest-approach vector (stable in fp32)
        float A = dot3(ld, ld), invA = fast_rcp(A);
        float b = dot3(lo, ld);
        f3 l = fma3(ld, -b * invA, lo);
        float disc = A * (1.f - dot3(l, l));
        if (disc >= 0.f) {
            float sq = sqrtf(disc);
            float q = (b < 0.f) ? -(b - sq) : -(b + sq); // = -0.5 (B -/+ sqrt(discrim)) with B = 2b
            float c = dot3(lo, lo) - 1.f;
            float t0 = q * invA, t1 = c / q;
            float nearT = fminf(t0, t1), farT = fmaxf(t0, t1);
            // sphere.cpp:176-188
            if (nearT <= h.t && farT >= tmin) {
                float t = nearT;
                bool ok = true;
                if (nearT < tmin) { t = farT; ok = farT <= h.t; }
                if (ok) { h.prim = idx; h.t = t; h.u = 0.f; h.v = 0.f; }
            }
        }
    }
}

// The same test for one record PER LANE (a BVH leaf), as straight-line code: with the kind in a vector register the
// selects of intersect_prim become branches over the lane mask. `ok`: the lane holds a leaf (the others computed on zeros).
template <int FEAT = 15> DEV void intersect_leaf(const DPrim &P, bool ok, f3 o, f3 d, float tmin, Hit &h) {
    const f3 lo = mk3(fmaf(P.m[0], o.x, fmaf(P.m[1], o.y, fmaf(P.m[2], o.z, P.m[3]))),
                      fmaf(P.m[4], o.x, fmaf(P.m[5], o.y, fmaf(P.m[6], o.z, P.m[7]))),
                      fmaf(P.m[8], o.x, fmaf(P.m[9], o.y, fmaf(P.m[10], o.z, P.m[11]))));
    const f3 ld = mk3(fmaf(P.m[0], d.x, fmaf(P.m[1], d.y, P.m[2] * d.z)), fmaf(P.m[4], d.x, fmaf(P.m[5], d.y, P.m[6] * d.z)),
                      fmaf(P.m[8], d.x, fmaf(P.m[9], d.y, P.m[10] * d.z)));
    const int type = P.type;
    const float t = -lo.z * fast_rcp(ld.z);
    const float u = fmaf(t, ld.x, lo.x), v = fmaf(t, ld.y, lo.y);
    const float w_tri = 1.f - (u + v), w_par = fminf(1.f - u, 1.f - v);
    const float w = type == PRIM_TRIANGLE ? w_tri : w_par;
    bool hit = ok & (fminf(fminf(u, v), w) >= 0.f) & (t >= tmin) & (t <= h.t);
    if (FEAT & 4) hit = hit & (type != PRIM_SPHERE);
    const bool quad2 = type == PRIM_QUAD2, second = quad2 & (v > u); // sub-triangle (a,b,c) for v <= u, (a,c,d) otherwise; its own barycentrics
    const float uq = second ? u : u - v, vq = second ? v - u : v;
    h.prim = hit ? P.shade + (second ? 1 : 0) : h.prim;
    h.t = hit ? t : h.t;
    h.u = hit ? (quad2 ? uq : u) : h.u;
    h.v = hit ? (quad2 ? vq : v) : h.v;
    if (FEAT & 4) {
        const bool sph = ok & (type == PRIM_SPHERE);
        if (__ballot(sph)) { // wave-uniform: scenes with spheres keep the general test for them
            if (sph) intersect_prim<FEAT>(P, P.shade, o, d, tmin, h);
        }
    }
}

// closest hit in [tmin, tmax] over all primitives. The loop index is wave-uniform, so the 64 B
// primitive record is fetched through the SCALAR cache (constant address space => s_load_dwordx16)
// and feeds the VALU ops as SGPR operands: no VGPRs, no vector-memory latency in the loop.
typedef const DPrim __attribute__((address_space(4))) *ScalarPrimPtr;

template <int FEAT = 15> DEV Hit trace_brute(const DParams &P, f3 o, f3 d, float tmin, float tmax) {
    Hit h{-1, tmax, 0.f, 0.f};
    const int n = P.n_prims;
    if (P.debug & 64) { // A/B: vector-memory path
        for (int i = 0; i < n; ++i) intersect_prim(P.prims[i], P.prims[i].shade, o, d, tmin, h);
        return h;
    }
    ScalarPrimPtr sp = (ScalarPrimPtr) (uintptr_t) P.prims;
    // ping-pong software pipeline: while primitive i is tested (~35 VALU ops) the record of
    // primitive i+1 is already in flight through the scalar cache, and vice versa
    auto load = [&](int i, DPrim &G) {
#pragma unroll
        for (int k = 0; k < 12; ++k) G.m[k] = sp[i].m[k];
        const int ks = sp[i].kind_shade; // one scalar dword: kind in the low byte, shading record index above
        G.type = ks & 0xff;
        G.shade = ks >> 8;
    };
    // SMEM returns out of order, so the only wait is lgkmcnt(0) = "everything outstanding". The
    // empty asm pins that wait on record X BEFORE the request for the other record is issued;
    // otherwise the compiler's wait for X (placed at X's first use) would also wait for the
    // record just requested and the pipeline would collapse to one load at a time.
#define PIN(G) asm volatile("" ::"s"(G.m[0]), "s"(G.m[4]), "s"(G.m[8]), "s"(G.type), "s"(G.shade))
    DPrim A, B;
    load(0, A);
    for (int i = 0; i < n; i += 2) {
        PIN(A);
        load(i + 1 < n ? i + 1 : i, B);
        intersect_prim<FEAT>(A, A.shade, o, d, tmin, h);
        PIN(B);
        load(i + 2 < n ? i + 2 : i, A);
        if (i + 1 < n) intersect_prim<FEAT>(B, B.shade, o, d, tmin, h);
    }
#undef PIN
    return h;
}

// 4-wide BVH traversal, one ray per lane, RESUMABLE: the traversal state of a lane (`Trav` + its column of the LDS stack)
// survives a return, so a caller whose lanes need very different numbers of node visits (k_mutate_v4 on BVH scenes: the
// rays of 32 unrelated paths and their shadow rays) can take the lanes that have finished out of the loop, let them
// shade and come back with new rays, while the long traversals simply continue -- the wave-level regrouping of live rays
// the lock-step form lacks (there a wave waits for its longest ray, lane utilisation ~10 %).
//   * node = 128 B, four child boxes (DBvh4Node); the hit children are ordered near-to-far by a 5-comparator network on
//     (entry distance | child slot) keys and the far ones pushed;
//   * per-lane stack in LDS, column layout [slot][lane]: a push or pop is one ds access without bank conflicts (a stack
//     in registers is indexed by a divergent sp: the compiler turns every push and pop into a compare-and-select over all
//     entries, ~100 VALU per node); the builder bounds the tree depth so that a push never overflows;
//   * leaf references and node indices share the stack entries (32 bit, or 16 bit for scenes that fit);
//   * "while-while" with a vote: per iteration the wave either advances the lanes that hold inner nodes or the lanes that
//     hold leaves, whichever are more.
struct Trav {
    f3 o, d, inv, oi;
    float tmin;
    Hit h;
    int cur, sp;      // cur >= 0 inner node, < 0 leaf reference
    int ovf;          // entries of this lane's stack that sit in the overflow area (0 for trees the LDS column holds)
    uint32_t rx, ry, rz; // 1 where the ray runs against the axis: the NEAR plane of a box is then its upper one (row lo/hi swapped at the load)
    bool active, any_hit;
    uint32_t n_nodes, n_prims; // fetched so far by this lane (k_mutate_v4 reports them: the scene part of the algorithmic bytes)
    uint32_t it_inner, it_leaf; // wave-uniform: traversal iterations of each kind (lane occupancy = n_nodes / (64 it_inner) ...)
};

DEV void trav_begin(Trav &T, f3 o, f3 d, float tmin, float tmax, bool any_hit) {
    T.o = o; T.d = d;
    T.inv = mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);
    T.oi = mk3(-o.x * T.inv.x, -o.y * T.inv.y, -o.z * T.inv.z);
    T.tmin = tmin;
    T.h = Hit{-1, tmax, 0.f, 0.f};
    T.cur = 0; T.sp = 0; T.ovf = 0;
    T.rx = T.inv.x < 0.f ? 1u : 0u; T.ry = T.inv.y < 0.f ? 1u : 0u; T.rz = T.inv.z < 0.f ? 1u : 0u;
    T.active = true; T.any_hit = any_hit;
}
DEV void trav_reset_counters(Trav &T) { T.n_nodes = T.n_prims = T.it_inner = T.it_leaf = 0u; }

DEV unsigned umin2(unsigned a, unsigned b) { return a < b ? a : b; }
DEV unsigned umax2(unsigned a, unsigned b) { return a < b ? b : a; }

// One wave's traversal machinery: the LDS stack column of this lane, the buffer resources of the node / primitive arrays, and
// `step` = ONE iteration for the whole wave (nodes or leaves, by vote). Two loops drive it: trav_run (every lane keeps the
// ray it was given until a slice ends) and k_mutate_v5's pool loop (a lane that finishes a ray takes the next one from
// the wave's ray queue inside the loop).
template <class StackT, class PT, bool OVF = true, int CAP = BVH_STACK, int FEAT = 15> struct TravLoop {
    static constexpr int SPILL = CAP / 2; // entries moved at a time
    static constexpr unsigned TRAV_NO_FETCH = 0xfffff000u;
    const PT &P;
    StackT *const stk;
    const size_t ovf_col;
    const bool has_ovf;
    const __amdgpu_buffer_rsrc_t r_bvh, r_prm;
    // `column`: this lane's column of a (CAP + 3) x 64 array in LDS. The column holds CAP entries (+ 3 spare: the branch-free
    // pushes write up to three entries above the top; lanes that hold no node: all three, from row CAP). A tree deeper than
    // CAP / 3 levels can need more. Slow paths: before a node's pushes could run past the column its SPILL OLDEST entries move
    // to this lane's column of an overflow area in memory and the rest slides down; a pop that finds the column empty brings
    // the newest SPILL back. Node and primitive records are raw buffers (byte offsets; drmlt_create refuses arrays of 2 GiB
    // and more).
    DEV TravLoop(const PT &P_, StackT *column)
        : P(P_), stk(column), ovf_col((size_t) blockIdx.x * 64u + (threadIdx.x & 63u)), has_ovf(OVF && P_.bvh_overflow != nullptr),
          r_bvh(__builtin_amdgcn_make_buffer_rsrc((void *) (uintptr_t) P_.bvh, 0, 0x80000000u, 0x00020000)),
          r_prm(__builtin_amdgcn_make_buffer_rsrc((void *) (uintptr_t) P_.prims, 0, 0x80000000u, 0x00020000)) {}
    DEV void spill(Trav &T) const {
        for (int i = 0; i < SPILL; ++i) P.bvh_overflow[(size_t) (T.ovf + i) * P.bvh_ovf_lanes + ovf_col] = (int) stk[i * 64];
        for (int i = SPILL; i < T.sp; ++i) stk[(i - SPILL) * 64] = stk[i * 64];
        T.sp -= SPILL; T.ovf += SPILL;
    }
    DEV void refill(Trav &T) const {
        T.ovf -= SPILL;
        for (int i = 0; i < SPILL; ++i) stk[i * 64] = (StackT) P.bvh_overflow[(size_t) (T.ovf + i) * P.bvh_ovf_lanes + ovf_col];
        T.sp = SPILL;
    }
    // Advance the lanes with `mine` set by one iteration. Returns, per lane, whether its traversal FINISHED in this iteration;
    // `any` (wave-uniform): some lane still had a traversal to advance (false: nothing was done).
    DEV bool step(Trav &T, bool mine, bool &any) const {
        const bool run = mine && T.active;
        const unsigned long long m_inner = __ballot(run && T.cur >= 0), m_leaf = __ballot(run && T.cur < 0);
        any = (m_inner | m_leaf) != 0ull;
        if (!any) return false;
        bool done_now = false;
        // Both blocks below are STRAIGHT-LINE code for the whole wave: a lane that does not hold the kind being advanced
        // computes on whatever its registers hold and keeps nothing of it (its pushes land in the spare rows above the
        // column, every state update is a select on `ok`). Only the fetches sit under the lane mask. As per-lane branches
        // the same code cost ~30 register copies and ~50 scalar instructions per iteration at the merges of its nested
        // conditions -- about as much as the node test itself.
        if (__popcll(m_inner) * P.trace_vote >= __popcll(m_leaf) * 16) { // a leaf test costs about 0.4 node tests: see drmlt_capi.cpp
            T.it_inner++;
            const bool ok = run && T.cur >= 0;
            if (has_ovf) { if (ok && T.sp > CAP - 3) spill(T); }
            // The node's rows (16 bytes each: lox, hix, loy, hiy, loz, hiz, child) are fetched with the lo / hi rows of an
            // axis SWAPPED where the ray runs against it: the first of each pair then holds the four NEAR planes, the second
            // the four FAR ones, and the slab test needs no per-axis min / max (24 of ~110 vector instructions per node).
            // Fetched through a buffer resource: a lane that holds no node offers an offset beyond the resource's range, for which
            // the load returns zeros without a memory access -- no lane mask, no branch, no register to initialise.
            const unsigned nb = ok ? (unsigned) T.cur << 7 : TRAV_NO_FETCH;
            const u32x4_t nxr = __builtin_amdgcn_raw_buffer_load_b128(r_bvh, nb + (T.rx << 4), 0, 0), fxr = __builtin_amdgcn_raw_buffer_load_b128(r_bvh, nb + ((T.rx ^ 1u) << 4), 0, 0);
            const u32x4_t nyr = __builtin_amdgcn_raw_buffer_load_b128(r_bvh, nb + 32u + (T.ry << 4), 0, 0), fyr = __builtin_amdgcn_raw_buffer_load_b128(r_bvh, nb + 32u + ((T.ry ^ 1u) << 4), 0, 0);
            const u32x4_t nzr = __builtin_amdgcn_raw_buffer_load_b128(r_bvh, nb + 64u + (T.rz << 4), 0, 0), fzr = __builtin_amdgcn_raw_buffer_load_b128(r_bvh, nb + 64u + ((T.rz ^ 1u) << 4), 0, 0);
            const u32x4_t chr = __builtin_amdgcn_raw_buffer_load_b128(r_bvh, nb + 96u, 0, 0);
            T.n_nodes += ok ? 1u : 0u;
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 ix = {T.inv.x, T.inv.x}, iy = {T.inv.y, T.inv.y}, iz = {T.inv.z, T.inv.z};
            const f2 ox = {T.oi.x, T.oi.x}, oy = {T.oi.y, T.oi.y}, oz = {T.oi.z, T.oi.z};
            float tn[4];
            bool hitc[4];
            const int childc[4] = {(int) chr.x, (int) chr.y, (int) chr.z, (int) chr.w};
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) { // children (0, 1), then (2, 3): v_pk_fma_f32
                auto pr = [&](const u32x4_t &r) { return h2 == 0 ? (f2){__uint_as_float(r.x), __uint_as_float(r.y)} : (f2){__uint_as_float(r.z), __uint_as_float(r.w)}; };
                const f2 t0x = __builtin_elementwise_fma(pr(nxr), ix, ox), t1x = __builtin_elementwise_fma(pr(fxr), ix, ox);
                const f2 t0y = __builtin_elementwise_fma(pr(nyr), iy, oy), t1y = __builtin_elementwise_fma(pr(fyr), iy, oy);
                const f2 t0z = __builtin_elementwise_fma(pr(nzr), iz, oz), t1z = __builtin_elementwise_fma(pr(fzr), iz, oz);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float a = fmaxf(fmaxf(t0x[q], t0y[q]), fmaxf(t0z[q], T.tmin));
                    const float f = fminf(fminf(t1x[q], t1y[q]), fminf(t1z[q], T.h.t));
                    tn[2 * h2 + q] = a;
                    hitc[2 * h2 + q] = a <= f;
                }
            }
            // Sort keys, nearest first. 16-bit child references: the key carries the child itself -- entry distance in the
            // upper half (its bit pattern orders like an unsigned: tn >= tmin >= 0; 7 mantissa bits are plenty for an ORDER),
            // reference in the lower half. 32-bit references: the key carries the SLOT in its two low mantissa bits.
            unsigned key[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned kk = sizeof(StackT) == 2 ? ((__float_as_uint(tn[c]) & 0xffff0000u) | ((unsigned) childc[c] & 0xffffu))
                                                        : ((__float_as_uint(tn[c]) & 0x7ffffffcu) | (unsigned) c);
                key[c] = hitc[c] ? kk : 0xffffffffu;
            }
            // sorting network (0,1)(2,3)(0,2)(1,3)(1,2)
            const unsigned a0 = umin2(key[0], key[1]), a1 = umax2(key[0], key[1]), a2 = umin2(key[2], key[3]), a3 = umax2(key[2], key[3]);
            const unsigned b0 = umin2(a0, a2), b2 = umax2(a0, a2), b1 = umin2(a1, a3), b3 = umax2(a1, a3);
            const unsigned k0 = b0, k1 = umin2(b1, b2), k2 = umax2(b1, b2), k3 = b3;
            auto child_of = [&](unsigned k) -> int {
                if constexpr (sizeof(StackT) == 2) return (int) (short) (k & 0xffffu);
                else {
                    const int lo = (k & 1u) ? childc[1] : childc[0], hi = (k & 1u) ? childc[3] : childc[2];
                    return (k & 2u) ? hi : lo;
                }
            };
            // far ones first, so that the nearest pending child ends on top; a write above the top of the stack is harmless
            int sp = ok ? T.sp : CAP;
            stk[sp * 64] = (StackT) child_of(k3); sp += k3 != 0xffffffffu ? 1 : 0;
            stk[sp * 64] = (StackT) child_of(k2); sp += k2 != 0xffffffffu ? 1 : 0;
            stk[sp * 64] = (StackT) child_of(k1); sp += k1 != 0xffffffffu ? 1 : 0;
            const bool has0 = k0 != 0xffffffffu;
            const int below = sp > 0 ? sp - 1 : 0;
            const int top = stk[below * 64]; // the pop, should the node have no child to descend into
            bool fin = !has0 && sp == 0;
            T.cur = ok && !fin ? (has0 ? child_of(k0) : top) : T.cur;
            T.sp = ok ? (has0 ? sp : below) : T.sp;
            if (has_ovf) { if (ok && fin && T.ovf > 0) { refill(T); T.cur = stk[--T.sp * 64]; fin = false; } }
            done_now = ok && fin;
            T.active = T.active && !done_now;
        } else {
            T.it_leaf++;
            const bool ok = run && T.cur < 0;
            // leaf reference: ~(first << shift | count); shift = 0 when every leaf holds one primitive (the default build)
            const int first = ~T.cur >> P.bvh_leaf_shift, n = P.bvh_leaf_shift ? (~T.cur & 7) : 1;
            T.n_prims += ok ? (uint32_t) n : 0u;
            if (P.bvh_leaf_shift == 0) { // wave-uniform
                union { DPrim v; u32x4_t w[4]; } G;
                const unsigned pb = ok ? (unsigned) first << 6 : TRAV_NO_FETCH;
#pragma unroll
                for (unsigned k = 0; k < 4u; ++k) G.w[k] = __builtin_amdgcn_raw_buffer_load_b128(r_prm, pb + 16u * k, 0, 0);
                intersect_leaf<FEAT>(G.v, ok, T.o, T.d, T.tmin, T.h);
            } else if (ok) {
                for (int i = 0; i < n; ++i) {
                    const DPrim G = load_global16(P.prims + first + i);
                    intersect_prim(G, G.shade, T.o, T.d, T.tmin, T.h);
                }
            }
            const bool found = T.any_hit && T.h.prim >= 0;
            const int sp = ok ? T.sp : 0;
            const int below = sp > 0 ? sp - 1 : 0;
            const int top = stk[below * 64];
            bool fin = found || sp == 0;
            T.cur = ok && !fin ? top : T.cur;
            T.sp = ok ? below : T.sp;
            if (has_ovf) { if (ok && !found && sp == 0 && T.ovf > 0) { refill(T); T.cur = stk[--T.sp * 64]; fin = false; } }
            done_now = ok && fin;
            T.active = T.active && !done_now;
        }
        return done_now;
    }
};

// Advance the traversals of the lanes with `mine` set until all of them are done or, if yield_lanes > 0, at least that
// many lanes of the wave have finished during this call.
// StackT: int, or short when every node index and leaf reference of the scene fits 15 bits (drmlt_create decides): half
// the LDS, which is what lets k_mutate_v4 keep 8 waves per CU on BVH scenes (measured: 1.13e8 -> 2.0e8 mutations/s on the
// 2000-triangle soup, all of it occupancy).
// OVF: compile the spill / refill paths in (k_mutate_v4 has a build without them for trees that fit the column: they cost
// 4 % there even when never taken).
// CAP: entries of the LDS column (k_mutate_v4 gives its 32-bit stacks 12 instead of 24 -- with the overflow paths the column
// only has to hold the hot top of the stack, and 3.5 KB instead of 6.5 KB keep eight waves on a CU).
template <class StackT, class PT, bool OVF = true, int CAP = BVH_STACK, int FEAT = 15> DEV void trav_run(const PT &P, Trav &T, bool mine, int yield_lanes) {
    __shared__ StackT bvh_stack[(CAP + 3) * 64];
    const TravLoop<StackT, PT, OVF, CAP, FEAT> L(P, bvh_stack + (threadIdx.x & 63u)); // every kernel that traces runs one wave per workgroup
    int finished = 0;
    for (;;) {
        if (yield_lanes > 0 && finished >= yield_lanes) break;
        bool any;
        const bool done_now = L.step(T, mine, any);
        if (!any) break;
        finished += __popcll(__ballot(done_now));
    }
}

// closest (or any) hit in [tmin, tmax], traversal run to completion
template <class PT> DEV Hit trace_bvh(const PT &P, f3 o, f3 d, float tmin, float tmax, bool any_hit) {
    Trav T;
    trav_reset_counters(T);
    trav_begin(T, o, d, tmin, tmax, any_hit);
    trav_run<int>(P, T, true, 0);
    return T.h;
}

// Brute-force loop for scenes of flat primitives only (rectangles, triangles, merged pairs). Same tests as
// intersect_prim, restructured around the scalar unit, which was issuing almost as many instructions as the VALU:
//   * DPrimFlat: the (u, v) rows arrive as aligned SGPR pairs -> packed FMAs without scalar shuffles;
//   * two sentinel records end the array, so the software pipeline reads ahead without clamping its index and the
//     loads take immediate offsets from one running pointer;
//   * the sub-triangle of a merged pair is resolved once, for the final hit, not per record.
typedef float float2v __attribute__((ext_vector_type(2)));
typedef const DPrimFlat __attribute__((address_space(4))) *ScalarFlatPtr;

struct FlatRec {
    float2v cx, cy, cz, cw;
    float z0, z1, z2, z3;
    int ks;
};

template <bool TRI> DEV void test_flat(const FlatRec &G, f3 o, f3 d, float tmin, float &best_t, float2v &best_uv, int &best_ks) {
    const float ldz = fmaf(G.z0, d.x, fmaf(G.z1, d.y, G.z2 * d.z));
    const float loz = fmaf(G.z0, o.x, fmaf(G.z1, o.y, fmaf(G.z2, o.z, G.z3)));
    const float t = -loz * fast_rcp(ldz);
    const float2v lo = G.cx * o.x + (G.cy * o.y + (G.cz * o.z + G.cw));
    const float2v ld = G.cx * d.x + (G.cy * d.y + G.cz * d.z);
    const float2v uv = ld * t + lo;
    float w;
    if (TRI && (G.ks & 0xff) == PRIM_TRIANGLE) w = 1.f - (uv.x + uv.y);
    else { const float2v om = 1.f - uv; w = fminf(om.x, om.y); }
    const bool hit = fminf(fminf(uv.x, uv.y), w) >= 0.f && t >= tmin && t <= best_t;
    best_t = hit ? t : best_t;
    best_uv.x = hit ? uv.x : best_uv.x;
    best_uv.y = hit ? uv.y : best_uv.y;
    best_ks = hit ? G.ks : best_ks;
}

// ---- cuboid records (host side and the argument why this equals the loop over the separate faces: box_merge.h)
// The ray in the cuboid's own coordinates b in [0, 1]^3 (the affine map of a flat record), the three slabs, then the face the
// ray ENTERS through (entry distance >= tmin) or, failing that -- the ray starts inside, or the scene has no face there: the
// open side of a room -- the face it LEAVES through. Which of the two is needed is mostly the same for a whole wave (rays start
// inside a room and outside a box), so each half sits behind a wave-uniform test. The hit is handed on as the face's half-word
// (exists | code << 1 | kind << 4 | shade << 6) and the in-face coordinates (p, q) = (b[j], b[k]), j < k the other two axes;
// trace_flat turns the winner's into the face's own (u, v) and record once, after the loops.
#define BOX_FACE_FLAG 0x40000000
typedef const DPrimBox __attribute__((address_space(4))) *ScalarBoxPtr;
struct BoxRec {
    float2v cx, cy, cz, cw;
    float z0, z1, z2, z3;
    uint32_t f0, f1, f2; // face words of axes 0, 1, 2: low half side 0 (b = 0), high half side 1 (b = 1)
};
// ONE candidate face for a lane: the face the ray enters through (`leave` false) or leaves through. Moving along +axis a ray
// enters through side 0 and leaves through side 1, along -axis the other way round: side = (ld[axis] < 0) != leave. Everything by
// value (arguments and result): through references the choice among the three face words compiles to a load through a selected
// address, and the words then live in scratch memory.
struct BoxCand { float t, p, q; uint32_t hc; };
DEV BoxCand box_candidate(bool leave, float nx, float ny, float fx, float fy, float tnear, float tfar, float2v lo, float2v ld, float loz, float ldz,
                          uint32_t f0, uint32_t f1, uint32_t f2) {
    BoxCand c;
    c.t = leave ? tfar : tnear;
    const float cx = leave ? fx : nx, cy = leave ? fy : ny;
    const bool a0 = cx == c.t, a1 = cy == c.t;                        // the axis whose plane the ray crosses at t
    const float lda = a0 ? ld.x : (a1 ? ld.y : ldz);
    const uint32_t w = a0 ? f0 : (a1 ? f1 : f2);
    c.hc = ((lda < 0.f) != leave) ? w >> 16 : w & 0xffffu;
    const float2v bxy = ld * c.t + lo;
    const float bzz = fmaf(ldz, c.t, loz);
    c.p = a0 ? bxy.y : bxy.x; c.q = (a0 || a1) ? bzz : bxy.y;       // in-face coordinates: axis 0 (y, z), 1 (x, z), 2 (x, y)
    return c;
}
DEV void test_box(const BoxRec &G, f3 o, f3 d, float tmin, float &best_t, float2v &best_uv, int &best_ks) {
    const uint32_t f0 = G.f0, f1 = G.f1, f2 = G.f2;
    const float ldz = fmaf(G.z0, d.x, fmaf(G.z1, d.y, G.z2 * d.z));
    const float loz = fmaf(G.z0, o.x, fmaf(G.z1, o.y, fmaf(G.z2, o.z, G.z3)));
    const float2v lo = G.cx * o.x + (G.cy * o.y + (G.cz * o.z + G.cw));
    const float2v ld = G.cx * d.x + (G.cy * d.y + G.cz * d.z);
    const float ix = fast_rcp(ld.x), iy = fast_rcp(ld.y), iz = fast_rcp(ldz);
    // slab distances. (1 - lo) * inv, not t0 + inv: a ray parallel to a slab (inv = inf) must see (-inf, +inf) inside it and
    // two equal infinities outside
    const float ax = -lo.x * ix, bx = (1.f - lo.x) * ix, ay = -lo.y * iy, by = (1.f - lo.y) * iy, az = -loz * iz, bz = (1.f - loz) * iz;
    const float nx = fminf(ax, bx), fx = fmaxf(ax, bx), ny = fminf(ay, by), fy = fmaxf(ay, by), nz = fminf(az, bz), fz = fmaxf(az, bz);
    const float tnear = fmaxf(fmaxf(nx, ny), nz), tfar = fminf(fminf(fx, fy), fz);
    // One candidate per lane: the entry face if the ray comes from outside (entry distance >= tmin), the exit face otherwise. All
    // selects on values (a flag that lives across a branch costs a v_cndmask and a v_cmp at every merge).
    const bool through = tnear <= tfar && tfar >= tmin && tnear <= best_t;
    const bool inside = !(tnear >= tmin);
    BoxCand c = box_candidate(inside, nx, ny, fx, fy, tnear, tfar, lo, ld, loz, ldz, f0, f1, f2);
    // the scene has no face where the ray enters (the open side of a room seen from outside): the face it leaves through. Rare, and
    // then for many lanes at once: behind a wave-uniform test.
    const bool hole = through && !inside && (c.hc & 1u) == 0u;
    if (__ballot(hole)) {
        const BoxCand c2 = box_candidate(true, nx, ny, fx, fy, tnear, tfar, lo, ld, loz, ldz, f0, f1, f2);
        c.t = hole ? c2.t : c.t; c.p = hole ? c2.p : c.p; c.q = hole ? c2.q : c.q; c.hc = hole ? c2.hc : c.hc;
    }
    const bool hit = through && (c.hc & 1u) != 0u && c.t >= tmin && c.t <= best_t;
    best_t = hit ? c.t : best_t;
    best_uv.x = hit ? c.p : best_uv.x;
    best_uv.y = hit ? c.q : best_uv.y;
    best_ks = hit ? (int) (c.hc | BOX_FACE_FLAG) : best_ks;
}

template <bool TRI> DEV Hit trace_flat(const DParams &P, f3 o, f3 d, float tmin, float tmax) {
    ScalarFlatPtr p = (ScalarFlatPtr) (uintptr_t) P.prims_flat;
    // VOLATILE scalar loads (three per record: 8 + 4 + 1 dwords): they stay where the pipeline below puts them. Left to its
    // scheduler the compiler sinks the read-ahead of a record to just before its first use and the wave then waits out a
    // scalar-cache access per record.
    auto load = [](ScalarFlatPtr q, FlatRec &G) {
#ifdef DRMLT_FLAT_PLAIN_LOADS
        G.cx = float2v{q->c[0], q->c[1]}; G.cy = float2v{q->c[2], q->c[3]}; G.cz = float2v{q->c[4], q->c[5]}; G.cw = float2v{q->c[6], q->c[7]};
        G.z0 = q->rz[0]; G.z1 = q->rz[1]; G.z2 = q->rz[2]; G.z3 = q->rz[3];
        G.ks = q->kind_shade;
#else
        typedef float f8v __attribute__((ext_vector_type(8)));
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f8v c = *(const volatile f8v __attribute__((address_space(4))) *) &q->c[0];
        const f4v z = *(const volatile f4v __attribute__((address_space(4))) *) &q->rz[0];
        G.ks = *(const volatile int __attribute__((address_space(4))) *) &q->kind_shade;
        G.cx = float2v{c[0], c[1]}; G.cy = float2v{c[2], c[3]}; G.cz = float2v{c[4], c[5]}; G.cw = float2v{c[6], c[7]};
        G.z0 = z[0]; G.z1 = z[1]; G.z2 = z[2]; G.z3 = z[3];
#endif
    };
#define PINF(G) asm volatile("" ::"s"(G.cx.x), "s"(G.cz.x), "s"(G.z0), "s"(G.ks))
    float best_t = tmax;
    float2v best_uv = {0.f, 0.f};
    int best_ks = -1;
    // ---- cuboids first (record n_box is a sentinel: the read-ahead needs no clamp)
    if (P.n_box > 0) {
        auto loadb = [](ScalarBoxPtr q, BoxRec &G) {
            typedef float f8v __attribute__((ext_vector_type(8)));
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f8v c = *(const volatile f8v __attribute__((address_space(4))) *) &q->c[0];
            const f4v z = *(const volatile f4v __attribute__((address_space(4))) *) &q->rz[0];
            // (three separate scalar loads: as elements of one loaded vector the per-lane choice among them in test_box became an
            // indexed load from a copy of the vector in scratch memory)
            typedef const volatile uint32_t __attribute__((address_space(4))) *SU;
            G.f0 = *(SU) &q->fw[0]; G.f1 = *(SU) &q->fw[1]; G.f2 = *(SU) &q->fw[2];
            G.cx = float2v{c[0], c[1]}; G.cy = float2v{c[2], c[3]}; G.cz = float2v{c[4], c[5]}; G.cw = float2v{c[6], c[7]};
            G.z0 = z[0]; G.z1 = z[1]; G.z2 = z[2]; G.z3 = z[3];
        };
        ScalarBoxPtr pb = (ScalarBoxPtr) (uintptr_t) P.prims_box;
        BoxRec A, B;
        loadb(pb, A);
        for (int i = 0; i < P.n_box; i += 2, pb += 2) {
            asm volatile("" ::"s"(A.cx.x), "s"(A.z0), "s"(A.f0));
            loadb(pb + 1, B);
            test_box(A, o, d, tmin, best_t, best_uv, best_ks);
            if (i + 1 < P.n_box) {
                asm volatile("" ::"s"(B.cx.x), "s"(B.z0), "s"(B.f0));
                loadb(pb + 2, A);
                test_box(B, o, d, tmin, best_t, best_uv, best_ks);
            }
        }
    }
    const int n = P.n_flat_rec;
    if (n > 0) {
        FlatRec A, B;
        load(p, A);
        for (int i = 0; i < n; i += 2, p += 2) { // records n and n + 1 are sentinels: the read-ahead needs no clamp
            PINF(A);
            load(p + 1, B);
            test_flat<TRI>(A, o, d, tmin, best_t, best_uv, best_ks);
            if (i + 1 < n) { // (with the cuboids most flat scenes keep one or two flat records -- the light: an odd count no longer pays for testing a sentinel)
                PINF(B);
                load(p + 2, A);
                test_flat<TRI>(B, o, d, tmin, best_t, best_uv, best_ks);
            }
        }
    }
#undef PINF
    Hit h{-1, best_t, best_uv.x, best_uv.y};
    if (best_ks >= 0) {
        if (best_ks & BOX_FACE_FLAG) { // a cuboid's face: its own (u, v) from the in-face coordinates, its own record
            const uint32_t hc = (uint32_t) best_ks & 0xffffu;
            const float uu = (hc & 2u) ? h.v : h.u, vv = (hc & 2u) ? h.u : h.v;
            h.u = (hc & 4u) ? 1.f - uu : uu;
            h.v = (hc & 8u) ? 1.f - vv : vv;
            best_ks = (int) (((hc >> 4) & 3u) | ((hc >> 6) << 8));
        }
        h.prim = best_ks >> 8;
        if ((best_ks & 0xff) == PRIM_QUAD2) { // sub-triangle (a,b,c) for v <= u, (a,c,d) otherwise; its own barycentrics
            const bool second = h.v > h.u;
            const float uu = second ? h.u : h.u - h.v, vv = second ? h.v - h.u : h.v;
            h.prim += second ? 1 : 0;
            h.u = uu; h.v = vv;
        }
    }
    return h;
}

template <int FEAT = 15> DEV Hit trace(const DParams &P, f3 o, f3 d, float tmin, float tmax, bool any_hit) {
    if ((FEAT & 8) && P.use_bvh) return trace_bvh(P, o, d, tmin, tmax, any_hit);
    if (P.prims_flat) {
        Hit h = P.has_plain_tri ? trace_flat<true>(P, o, d, tmin, tmax) : trace_flat<false>(P, o, d, tmin, tmax);
        if (FEAT & 4) // the spheres of the scene follow the flat records; same wave-uniform scalar loads, one at a time
            for (int i = P.n_flat; i < P.n_prims; ++i) { // (P.prims: every record of the scene, flat ones first -- n_flat of them, cuboid faces included)
                ScalarPrimPtr sp = (ScalarPrimPtr) (uintptr_t) P.prims;
                DPrim G;
#pragma unroll
                for (int k = 0; k < 12; ++k) G.m[k] = sp[i].m[k];
                const int ks = sp[i].kind_shade;
                G.type = PRIM_SPHERE; G.shade = ks >> 8;
                intersect_prim<4>(G, G.shade, o, d, tmin, h);
            }
        return h;
    }
    return trace_brute<FEAT>(P, o, d, tmin, tmax);
}

// ------------------------------------------------------------------ scene tables
// Shading / BSDF / emitter records are read per lane after a hit. Small scenes stage them in LDS
// once per launch (a dependent chain of 3-4 global gathers per step was ~1/3 of the wave's time
// at one wave per SIMD); large scenes read them from HBM/L2.
struct GlobalTables {
    const DShade *sh; const DBsdf *bs; const DEmitter *em;
    DEV DShade shade(int i) const { return load_global16(sh + i); }
    DEV DBsdf bsdf(int i) const { return load_global16(bs + i); }
    DEV DEmitter emitter(int i) const { return load_global16(em + i); }
    DEV float emitter_cdf_lo(int i) const { return load_global_f32(&em[i].cdf_lo); }
    DEV DShade emitter_shade(int, const DEmitter &E) const { return shade(E.prim); } // the shading record of emitter ei's shape
};
struct LdsTables {
    uint32_t shade_off, bsdf_off, emit_off; // float offsets into lds_x, multiples of 4
    // explicit field assignment from four 16 B LDS reads: going through a float* view of the struct
    // would park the record in scratch memory
    DEV DShade shade(int i) const {
        const float4 *q = reinterpret_cast<const float4 *>(&lds_x[shade_off + (uint32_t) i * 16u]);
        const float4 a = q[0], b = q[1], c = q[2], d = q[3];
        DShade s;
        s.origin[0] = a.x; s.origin[1] = a.y; s.origin[2] = a.z; s.eu[0] = a.w;
        s.eu[1] = b.x; s.eu[2] = b.y; s.ev[0] = b.z; s.ev[1] = b.w;
        s.ev[2] = c.x; s.n[0] = c.y; s.n[1] = c.z; s.n[2] = c.w;
        s.inv_len_eu = d.x; s.bsdf = __float_as_int(d.y); s.emitter = __float_as_int(d.z); s.inv_area = d.w;
        return s;
    }
    DEV DBsdf bsdf(int i) const {
        const float4 *q = reinterpret_cast<const float4 *>(&lds_x[bsdf_off + (uint32_t) i * 12u]);
        const float4 a = q[0], b = q[1], c = q[2];
        DBsdf r;
        r.type = __float_as_int(a.x); r.rgb[0] = a.y; r.rgb[1] = a.z; r.rgb[2] = a.w;
        r.p[0] = b.x; r.p[1] = b.y; r.p[2] = b.z; r.p[3] = b.w;
        r.p[4] = c.x; r.p[5] = c.y; r.p[6] = c.z; r.p[7] = c.w;
        return r;
    }
    DEV DEmitter emitter(int i) const {
        const float4 *q = reinterpret_cast<const float4 *>(&lds_x[emit_off + (uint32_t) i * 8u]);
        const float4 a = q[0], b = q[1];
        DEmitter e;
        e.radiance[0] = a.x; e.radiance[1] = a.y; e.radiance[2] = a.z; e.prim = __float_as_int(a.w);
        e.cdf_lo = b.x; e.cdf_hi = b.y; e.pad[0] = b.z; e.pad[1] = b.w;
        return e;
    }
    DEV float emitter_cdf_lo(int i) const { return lds_x[emit_off + (uint32_t) i * 8u + 4u]; }
    DEV DShade emitter_shade(int, const DEmitter &E) const { return shade(E.prim); }
};
// BSDF and emitter records in LDS, shading records in device memory: for kernels whose own rows leave less LDS than the shading table
// needs (k_mutate_bdpt at two waves per SIMD: 19.25 KB of rows; the Cornell scene's 30 shading records are 1.9 KB, its four BSDFs and one
// emitter 224 bytes)
// `L.shade_off` holds one shading record per EMITTER (the record of its shape, in emitter order): what a light sample needs after the
// emitter is picked -- otherwise a dependent round trip to memory behind the pick.
struct MixedTables {
    const DShade *sh;
    LdsTables L;
    DEV DShade shade(int i) const { return load_global16(sh + i); }
    DEV DBsdf bsdf(int i) const { return L.bsdf(i); }
    DEV DEmitter emitter(int i) const { return L.emitter(i); }
    DEV float emitter_cdf_lo(int i) const { return L.emitter_cdf_lo(i); }
    DEV DShade emitter_shade(int ei, const DEmitter &) const { return L.shade(ei); }
};
// The same with a fallback: scenes whose BSDF / emitter tables do not fit beside a kernel's rows read them from device memory
// (wave-uniform `lds`). The BVH builds of k_mutate_v5: a path step's chain of dependent gathers -- shading record -> BSDF, light pick
// -> emitter -> its shape's record -- shrinks to the shading record alone.
struct HybridTables {
    const DShade *sh; const DBsdf *bs; const DEmitter *em;
    LdsTables L;
    bool lds;
    DEV DShade shade(int i) const { return load_global16(sh + i); }
    DEV DBsdf bsdf(int i) const { if (lds) return L.bsdf(i); return load_global16(bs + i); }
    DEV DEmitter emitter(int i) const { if (lds) return L.emitter(i); return load_global16(em + i); }
    DEV float emitter_cdf_lo(int i) const { if (lds) return L.emitter_cdf_lo(i); return load_global_f32(&em[i].cdf_lo); }
    DEV DShade emitter_shade(int ei, const DEmitter &E) const { if (lds) return L.shade(ei); return load_global16(sh + E.prim); }
};
DEV uint32_t small_tables_floats(const DParams &P) { return (uint32_t) P.n_emitters * 16u + (uint32_t) P.n_bsdfs * 12u + (uint32_t) P.n_emitters * 8u; }
// emitter shape records at T.shade_off, BSDFs at T.bsdf_off, emitters at T.emit_off
DEV void stage_bsdfs_emitters(const DParams &P, const LdsTables &T, uint32_t lane) {
    const float *src = reinterpret_cast<const float *>(P.bsdfs);
    for (uint32_t i = lane; i < (uint32_t) P.n_bsdfs * 12u; i += 64u) lds_x[T.bsdf_off + i] = src[i];
    src = reinterpret_cast<const float *>(P.emitters);
    for (uint32_t i = lane; i < (uint32_t) P.n_emitters * 8u; i += 64u) lds_x[T.emit_off + i] = src[i];
    for (uint32_t i = lane; i < (uint32_t) P.n_emitters * 16u; i += 64u) {
        const int prim = P.emitters[i >> 4].prim;
        lds_x[T.shade_off + i] = reinterpret_cast<const float *>(P.shade + prim)[i & 15u];
    }
}
// cooperative copy of the three tables into LDS by one wave
DEV void stage_tables(const DParams &P, const LdsTables &T, uint32_t lane) {
    const float *src = reinterpret_cast<const float *>(P.shade);
    for (uint32_t i = lane; i < (uint32_t) P.n_shade * 16u; i += 64u) lds_x[T.shade_off + i] = src[i];
    src = reinterpret_cast<const float *>(P.bsdfs);
    for (uint32_t i = lane; i < (uint32_t) P.n_bsdfs * 12u; i += 64u) lds_x[T.bsdf_off + i] = src[i];
    src = reinterpret_cast<const float *>(P.emitters);
    for (uint32_t i = lane; i < (uint32_t) P.n_emitters * 8u; i += 64u) lds_x[T.emit_off + i] = src[i];
}

// ------------------------------------------------------------------ estimator state machine
// MIPathTracer::Li + the sampleSplats prologue as a resumable machine. One call of path_step
// consumes the result of the ray query issued by the previous call and issues the next one.
// All PSS components a step needs are drawn at ONE site (the `next` loop below): the sampler
// code (Philox + transition kernels) is the bulk of the instruction footprint, so it is
// instantiated once per kernel instead of once per consumer.
enum { PH_DONE = 0, PH_BEGIN = 1, PH_CLOSEST = 2, PH_SHADOW = 3, PH_IDLE = 4, PH_FLUSH = 5 };

struct PathState {
    f3 o, d;          // ray to trace next (o doubles as the current surface point)
    float tmin, tmax;
    f3 thr, Li;
    f3 n, s, wi;      // shading frame (t = n x s) and local incident direction at the current vertex
    f3 nee;           // NEE contribution pending on the shadow ray
    f3 bweight;       // BSDF sample weight pending on the bounce ray
    float bpdf, beta_eta, eta;
    float bx, by;     // BSDF sample components, drawn together with the NEE ones
    float px, py;     // film position of this sample
    int phase, depth, bsdf;
    uint32_t k;       // next PSS dimension
    uint32_t nrays;
    bool non_specular, direct_on, has_bounce, bdelta, refn_zero;
    bool shadow_pending; // dual-lane mode: ps.nee waits for the partner lane's occlusion result
};

// shadow ray handed to the partner lane (dual-lane mode)
struct ShadowRay {
    f3 o, d;
    float tmin, tmax;
    bool valid;
};

DEV float ray_eps_closest(f3 o) { // skdtree.cpp:125-129
    return EPSILON_F * fmaxf(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)), EPSILON_F);
}
DEV float ray_eps_shadow(f3 o) { // skdtree.cpp:213-218
    return EPSILON_F * fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
}

// Sphere as an area light seen from `ref` (sphere.cpp:286-385): uniform sampling of the cone it subtends when the
// reference point is outside, uniform area sampling otherwise. Outputs direction, distance, surface normal, solid-angle pdf.
DEV void sphere_sample_direct(f3 c, float radius, float inv_area, f3 ref, float sx, float sy, f3 &d, float &dist, f3 &n, float &pdf) {
    const f3 rc = c - ref;
    const float refDist2 = dot3(rc, rc);
    const float invRefDist = rsqrtf(refDist2);
    const float sinAlpha = radius * invRefDist;
    if (sinAlpha < 1.f - EPSILON_F) {
        const float cosAlpha = sqrtf(fmaxf(0.f, 1.f - sinAlpha * sinAlpha));
        const float cosTheta = (1.f - sx) + sx * cosAlpha, sinTheta = sqrtf(fmaxf(0.f, 1.f - cosTheta * cosTheta));
        const f3 fn = rc * invRefDist;
        f3 fs, ft; // Frame(n): coordinateSystem (util.cpp:606-616)
        if (fabsf(fn.x) > fabsf(fn.y)) { float inv = rsqrtf(fn.x * fn.x + fn.z * fn.z); ft = mk3(fn.z * inv, 0.f, -fn.x * inv); }
        else { float inv = rsqrtf(fn.y * fn.y + fn.z * fn.z); ft = mk3(0.f, fn.z * inv, -fn.y * inv); }
        fs = cross3(ft, fn);
        d = fma3(fs, cos_rev(sy) * sinTheta, fma3(ft, sin_rev(sy) * sinTheta, fn * cosTheta));
        pdf = 0.15915494309189535f / (1.f - cosAlpha);
        const float projDist = dot3(rc, d);
        const float baseT = refDist2 / projDist;
        const f3 qc = c - fma3(d, baseT, ref);
        const float queryDist2 = dot3(qc, qc), queryProjDist = dot3(qc, d);
        // solveQuadratic(1, -2 queryProjDist, queryDist2 - r^2), util.cpp:447-485
        const float B = -2.f * queryProjDist, C = queryDist2 - radius * radius;
        const float discrim = B * B - 4.f * C;
        float nearT = queryProjDist;
        if (discrim >= 0.f) {
            const float sq = sqrtf(discrim);
            const float temp = B < 0.f ? -0.5f * (B - sq) : -0.5f * (B + sq);
            nearT = fminf(temp, C / temp);
        }
        dist = baseT + nearT;
        n = normalize3(d * nearT - qc);
    } else {
        const float z = 1.f - 2.f * sy, r = sqrtf(fmaxf(0.f, 1.f - z * z));
        n = mk3(r * cos_rev(sx), r * sin_rev(sx), z);
        const f3 dv = fma3(n, radius, c) - ref;
        const float dist2 = dot3(dv, dv);
        dist = sqrtf(dist2);
        d = dv * (1.f / dist);
        pdf = inv_area * dist2 / fabsf(dot3(d, n));
    }
}
DEV float sphere_pdf_direct(f3 c, float radius, float inv_area, f3 ref, float dist, float cos_light) { // sphere.cpp:357-385
    const f3 rc = c - ref;
    const float sinAlpha = radius * rsqrtf(dot3(rc, rc));
    if (sinAlpha < 1.f - EPSILON_F) return 0.15915494309189535f / (1.f - sqrtf(fmaxf(0.f, 1.f - sinAlpha * sinAlpha)));
    return inv_area * dist * dist / cos_light;
}

// warp.cpp:81-102 + :43-52 (angles expressed in revolutions for v_sin/v_cos)
DEV f3 square_to_cosine_hemisphere(float sx, float sy) {
    float r1 = 2.f * sx - 1.f, r2 = 2.f * sy - 1.f;
    // selects instead of the three-way branch: num / den = r2 / r1 or r1 / r2, whichever has the larger denominator
    const bool first = r1 * r1 > r2 * r2;
    const float num = first ? r2 : r1, den = first ? r1 : r2;
    const float q = den != 0.f ? 0.125f * (num / den) : 0.f; // den == 0 only for r1 == r2 == 0: r = 0, angle irrelevant
    const float r = den, rev = first ? q : 0.25f - q;
    float px = r * cos_rev(rev), py = r * sin_rev(rev);
    float z = sqrtf(fmaxf(0.f, 1.f - px * px - py * py));
    if (z == 0.f) z = 1e-10f;
    return mk3(px, py, z);
}

// util.cpp:659-689
DEV float fresnel_dielectric_ext(float cosThetaI_, float &cosThetaT_, float eta) {
    if (eta == 1.f) { cosThetaT_ = -cosThetaI_; return 0.f; }
    float scale = (cosThetaI_ > 0.f) ? 1.f / eta : eta;
    float cosThetaTSqr = 1.f - (1.f - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.f) { cosThetaT_ = 0.f; return 1.f; }
    float cosThetaI = fabsf(cosThetaI_), cosThetaT = sqrtf(cosThetaTSqr);
    float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0.f) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}

DEV void path_init(const DParams &P, PathState &ps) {
    ps.phase = PH_BEGIN;
    ps.k = 0u;
    ps.nrays = 0u;
    ps.thr = mk3(1.f, 1.f, 1.f);
    ps.Li = mk3(0.f, 0.f, 0.f);
    ps.eta = 1.f;
    ps.depth = 1;
    ps.non_specular = false;
    ps.direct_on = P.exclude_direct == 0; // pathsampler.cpp:558-561
    ps.has_bounce = false;
    ps.refn_zero = false;
    ps.shadow_pending = false;
    ps.px = ps.py = 0.f;
}

// First step of every path: film position from the first two PSS components (pathsampler.cpp:538-543) and the
// camera ray (sampleRayDifferential, perspective.cpp:271-286). Leaves the path in PH_CLOSEST.
DEV void path_begin(const DParams &P, PathState &ps, float v0, float v1) {
    ps.k = 2u;
    ps.px = v0 * (float) P.width;
    ps.py = v1 * (float) P.height;
    f3 nearP = mk3((1.f - 2.f * v0) * P.tan_half_fov * P.near_clip, (1.f - 2.f * v1) * P.tan_half_fov * P.inv_aspect * P.near_clip,
                   P.near_clip);
    f3 dl = normalize3(nearP);
    float invZ = 1.f / dl.z;
    ps.tmin = P.near_clip * invZ;
    ps.tmax = P.far_clip * invZ;
    ps.o = mk3(P.cam[3], P.cam[7], P.cam[11]);
    ps.d = mk3(fmaf(P.cam[0], dl.x, fmaf(P.cam[1], dl.y, P.cam[2] * dl.z)), fmaf(P.cam[4], dl.x, fmaf(P.cam[5], dl.y, P.cam[6] * dl.z)),
               fmaf(P.cam[8], dl.x, fmaf(P.cam[9], dl.y, P.cam[10] * dl.z)));
    ps.phase = PH_CLOSEST;
    ps.nrays = 1u;
}

// Consume the result of the ray query issued for `ps` (none in PH_BEGIN) and either issue the
// next ray (PH_CLOSEST / PH_SHADOW) or finish the path (PH_DONE, radiance in ps.Li).
// DUAL = false: one lane per chain, shadow rays take a step of their own (PH_SHADOW).
// DUAL = true : two lanes per chain. This lane traces the camera/bounce rays, its partner traces
//   the shadow ray of the SAME vertex concurrently; `shadow_clear` is the partner's result for the
//   ray handed over in the previous step (`sr`), so a bounce costs one step instead of two. The
//   order of the radiance additions is the same in both modes (NEE of vertex i, then MIS of i+1).
// FEAT: scene features compiled in (bit 0 rough conductor, bit 1 dielectric, bit 2 spheres, bit 3 BVH); kernels for
// plain diffuse polygon scenes (the Cornell configs) carry none of the other code or its registers.
// HAS_BEGIN = false: the caller starts every path with path_begin itself (k_mutate_v4: in its bookkeeping branch), the
// step never sees PH_BEGIN and carries none of its code.
template <bool DUAL, int FEAT, class SamplerT, class TablesT, bool HAS_BEGIN = true>
DEV void path_step(const DParams &P, const TablesT &T, PathState &ps, SamplerT &smp, const Hit &hit, bool shadow_clear,
                   ShadowRay &sr) {
    // (row samplers: the five components a step can draw are requested HERE, all together and ahead of the shading record's gather -- one
    // round trip beside it instead of `need` of them behind it; what the step does not draw is not used)
    float pre0 = 0.f, pre1 = 0.f, pre2 = 0.f, pre3 = 0.f, pre4 = 0.f;
    if constexpr (draws_batched<SamplerT>::value) {
        const uint32_t kmax = (uint32_t) P.eff_dim - 1u, k = ps.k;
        pre0 = smp.next(min(k, kmax)); pre1 = smp.next(min(k + 1u, kmax)); pre2 = smp.next(min(k + 2u, kmax));
        pre3 = smp.next(min(k + 3u, kmax)); pre4 = smp.next(min(k + 4u, kmax));
    }
    // ---------------- part 1: digest the ray query, decide which PSS components are needed
    bool want_rr = false, want_nee = false;
    int need = 0;
    float rr_q = 1.f;
    f3 p = ps.o, n = ps.n, s = ps.s;
    DBsdf B;
    if (DUAL) {
        sr.valid = false;
        if (ps.shadow_pending) {
            if (shadow_clear) ps.Li = ps.Li + ps.nee;
            ps.shadow_pending = false;
        }
        if (ps.phase == PH_FLUSH) { ps.phase = PH_DONE; return; }
    }
    if (HAS_BEGIN && ps.phase == PH_BEGIN) {
        need = 2; // film position, pathsampler.cpp:538-543
    } else if (!DUAL && ps.phase == PH_SHADOW) {
        if (hit.prim < 0) ps.Li = ps.Li + ps.nee; // unoccluded
        B = T.bsdf(ps.bsdf);
    } else {
        if (hit.prim < 0) { ps.phase = PH_DONE; return; } // no environment emitter
        const DShade S = T.shade(hit.prim);
        const int ptype = S.bsdf >> 24; // primitive kind rides in the top byte
        // surface point + shading frame (skdtree.h:340-429, rectangle.cpp:155-168, sphere.cpp:207-255)
        if (!(FEAT & 4) || ptype != PRIM_SPHERE) {
            p = fma3(ld3(S.eu), hit.u, fma3(ld3(S.ev), hit.v, ld3(S.origin)));
            n = ld3(S.n);
            s = ld3(S.eu) * S.inv_len_eu;
        } else {
            f3 c = ld3(S.origin);
            f3 local = normalize3(fma3(ps.d, hit.t, ps.o) - c);
            p = fma3(local, S.eu[0], c);
            n = local;
            float zrad2 = local.x * local.x + local.y * local.y;
            float inv = rsqrtf(zrad2);
            s = zrad2 > 0.f ? mk3(-local.y * inv, local.x * inv, 0.f) : mk3(1.f, 0.f, 0.f);
        }
        if (ps.has_bounce) {
            ps.thr = ps.thr * ps.bweight;
            ps.eta *= ps.beta_eta;
            // emitter hit by the BSDF-sampled ray: MIS against direct sampling (path.cpp:269-285)
            if (S.emitter >= 0 && ps.direct_on && ps.non_specular) {
                const DEmitter E = T.emitter(S.emitter);
                float dn = dot3(ps.d, n);
                if (dn < 0.f) { // AreaLight::eval: dot(n, -d) > 0
                    float lumPdf = 0.f;
                    if (!ps.bdelta) { // pdfEmitterDirect: refN belongs to the PREVIOUS vertex (ps.n) or is 0
                        float dr = ps.refn_zero ? 0.f : dot3(ps.d, ps.n);
                        if (dr >= 0.f) {
                            lumPdf = S.inv_area * hit.t * hit.t / fabsf(dn);
                            if ((FEAT & 4) && ptype == PRIM_SPHERE) lumPdf = sphere_pdf_direct(ld3(S.origin), S.eu[0], S.inv_area, ps.o, hit.t, fabsf(dn));
                            lumPdf *= E.cdf_hi - E.cdf_lo;
                        }
                    }
                    float a = ps.bpdf * ps.bpdf, b = lumPdf * lumPdf;
                    ps.Li = fma3(ps.thr * ld3(E.radiance), a / (a + b), ps.Li);
                }
            }
            ps.direct_on = true;                   // rRec.type = ERadianceNoEmission
            want_rr = ps.depth++ >= P.rr_depth;    // russian roulette, path.cpp:297-307
        }
        // `depth >= maxDepth` ends the path whatever the roulette draw says, so test it first
        if (ps.depth >= P.max_depth && P.max_depth > 0) {
            if (want_rr) ps.k++;
            ps.phase = PH_DONE;
            return;
        }
        if (want_rr) rr_q = fminf(max3(ps.thr) * ps.eta * ps.eta, 0.95f);
        // adopt the new vertex
        f3 t = cross3(n, s);
        f3 md = -ps.d;
        ps.wi = mk3(dot3(md, s), dot3(md, t), dot3(md, n));
        ps.o = p; ps.n = n; ps.s = s;
        ps.bsdf = S.bsdf & 0xffffff;
        B = T.bsdf(ps.bsdf);
        ps.refn_zero = (FEAT & 2) && B.type == 1; // transmissive / two-sided: DirectSamplingRecord(its) zeroes refN
        want_nee = ps.direct_on && (B.type == 0 || ((FEAT & 1) && B.type == 2));
        need = (want_rr ? 1 : 0) + (want_nee ? 2 : 0) + 2;
    }

    // ---------------- part 2: the one place PSS components are drawn
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    if constexpr (draws_batched<SamplerT>::value) {
        v0 = pre0; v1 = pre1; v2 = pre2; v3 = pre3; v4 = pre4;
    } else {
#pragma nounroll
        for (int j = 0; j < need; ++j) {
            float v = smp.next(ps.k + (uint32_t) j);
            if (j == 0) v0 = v; else if (j == 1) v1 = v; else if (j == 2) v2 = v; else if (j == 3) v3 = v; else v4 = v;
        }
    }

    // ---------------- part 3: use them
    if (HAS_BEGIN && ps.phase == PH_BEGIN) {
        path_begin(P, ps, v0, v1);
        return;
    }
    if (ps.phase == PH_CLOSEST) {
        if (want_rr) {
            ps.k++;
            if (v0 >= rr_q) { ps.phase = PH_DONE; return; }
            ps.thr = ps.thr * (1.f / rr_q);
        }
        float sx = want_rr ? v1 : v0, sy = want_rr ? v2 : v1;
        const int bpos = (want_rr ? 1 : 0) + (want_nee ? 2 : 0);
        ps.bx = bpos == 0 ? v0 : (bpos == 1 ? v1 : (bpos == 2 ? v2 : v3));
        ps.by = bpos == 0 ? v1 : (bpos == 1 ? v2 : (bpos == 2 ? v3 : v4));
        // direct illumination sampling (path.cpp:187-218, scene.cpp:879-904)
        if (want_nee) {
            ps.k += 2u;
            f3 t = cross3(n, s);
            int ei = 0; // DiscreteDistribution::sample (lower_bound semantics)
            for (int i = 1; i < P.n_emitters; ++i)
                if (T.emitter_cdf_lo(i) < sx) ei = i;
            const DEmitter E = T.emitter(ei);
            float emPdf = E.cdf_hi - E.cdf_lo;
            sx = (sx - E.cdf_lo) / emPdf; // sampleReuse
            const DShade L = T.emitter_shade(ei, E);
            f3 lp;
            if ((L.bsdf >> 24) == PRIM_RECTANGLE) { // rectangle.cpp:210-216
                lp = fma3(ld3(L.eu), sx, fma3(ld3(L.ev), sy, ld3(L.origin))); // origin = corner (-1,-1), eu/ev = full edges
            } else { // single triangle: squareToUniformTriangle
                float a = sqrtf(fmaxf(0.f, 1.f - sx));
                lp = fma3(ld3(L.eu), 1.f - a, fma3(ld3(L.ev), a * sy, ld3(L.origin)));
            }
            f3 ln = ld3(L.n);
            f3 dv = lp - p;
            float dist2 = dot3(dv, dv), dist = sqrtf(dist2);
            f3 dd = dv * (1.f / dist);
            float dln = dot3(dd, ln);
            float pdf = dln != 0.f ? L.inv_area * dist2 / fabsf(dln) : 0.f; // Shape::sampleDirect
            if ((FEAT & 4) && (L.bsdf >> 24) == PRIM_SPHERE) // sphere light: cone sampling, sphere.cpp:286-355
                sphere_sample_direct(ld3(L.origin), L.eu[0], L.inv_area, p, sx, sy, dd, dist, ln, pdf), dln = dot3(dd, ln);
            float dr = ps.refn_zero ? 0.f : dot3(dd, n);
            if (dr >= 0.f && dln < 0.f && pdf != 0.f) { // AreaLight::sampleDirect
                f3 wo = mk3(dot3(dd, s), dot3(dd, t), dot3(dd, n));
                if (ps.wi.z > 0.f && wo.z > 0.f) {
                    f3 bsdfVal;
                    float bsdfPdf;
                    if (!(FEAT & 1) || B.type == 0) { // diffuse eval / pdf (diffuse.cpp:110-127)
                        bsdfVal = ld3(B.rgb) * (INV_PI_F * wo.z);
                        bsdfPdf = INV_PI_F * wo.z;
                    } else { // rough conductor (roughconductor.cpp:258-323)
                        const DRoughConductor rc{DMicrofacet{B.p[7] != 0.f, fmaxf(B.p[0], 1e-4f)}, mk3(B.p[1], B.p[2], B.p[3]),
                                                 mk3(B.p[4], B.p[5], B.p[6]), ld3(B.rgb)};
                        bsdfVal = rc.eval(ps.wi, wo);
                        bsdfPdf = rc.pdf(ps.wi, wo);
                    }
                    float lpdf = pdf * emPdf;
                    float a = lpdf * lpdf, b = bsdfPdf * bsdfPdf;
                    f3 value = ld3(E.radiance) * (1.f / lpdf);
                    f3 c = ps.thr * value * bsdfVal * (a / (a + b));
                    if (!is_zero3(c)) {
                        ps.nee = c;
                        ps.nrays++;
                        if (DUAL) { // partner lane traces it while this lane traces the bounce ray
                            sr.o = p; sr.d = dd;
                            sr.tmin = ray_eps_shadow(p);
                            sr.tmax = dist * (1.f - SHADOW_EPSILON_F);
                            sr.valid = true;
                            ps.shadow_pending = true;
                        } else {
                            ps.d = dd;
                            ps.tmin = ray_eps_shadow(p);
                            ps.tmax = dist * (1.f - SHADOW_EPSILON_F);
                            ps.phase = PH_SHADOW;
                            return;
                        }
                    }
                }
            }
        }
    }
    // ---------------- BSDF sampling (path.cpp:224-245), reached from PH_CLOSEST and PH_SHADOW
    ps.k += 2u;
    f3 wo;
    if (B.type == 0) { // diffuse.cpp:139-149
        if (!(ps.wi.z > 0.f)) { ps.phase = (DUAL && ps.shadow_pending) ? PH_FLUSH : PH_DONE; return; }
        wo = square_to_cosine_hemisphere(ps.bx, ps.by);
        ps.bpdf = INV_PI_F * wo.z;
        ps.bweight = ld3(B.rgb);
        ps.beta_eta = 1.f;
        ps.bdelta = false;
    } else if ((FEAT & 2) && B.type == 1) { // dielectric.cpp:270-306
        float eta = B.p[0], invEta = B.p[1];
        float cosThetaT;
        float F = fresnel_dielectric_ext(ps.wi.z, cosThetaT, eta);
        ps.bdelta = true;
        if (ps.bx <= F) {
            wo = mk3(-ps.wi.x, -ps.wi.y, ps.wi.z);
            ps.bpdf = F;
            ps.bweight = mk3(1.f, 1.f, 1.f);
            ps.beta_eta = 1.f;
        } else {
            float scale = -(cosThetaT < 0.f ? invEta : eta);
            wo = mk3(scale * ps.wi.x, scale * ps.wi.y, cosThetaT);
            ps.beta_eta = cosThetaT < 0.f ? eta : invEta;
            ps.bpdf = 1.f - F;
            float factor = cosThetaT < 0.f ? invEta : eta;
            ps.bweight = mk3(factor * factor, factor * factor, factor * factor);
        }
    } else if ((FEAT & 1) && B.type == 2) { // roughconductor.cpp:371-409
        const DRoughConductor rc{DMicrofacet{B.p[7] != 0.f, fmaxf(B.p[0], 1e-4f)}, mk3(B.p[1], B.p[2], B.p[3]),
                                 mk3(B.p[4], B.p[5], B.p[6]), ld3(B.rgb)};
        ps.bpdf = 0.f;
        ps.bweight = rc.sample(ps.wi, ps.bx, ps.by, wo, ps.bpdf);
        ps.beta_eta = 1.f;
        ps.bdelta = false;
    } else {
        ps.phase = (DUAL && ps.shadow_pending) ? PH_FLUSH : PH_DONE;
        return;
    }
    if (is_zero3(ps.bweight)) { ps.phase = (DUAL && ps.shadow_pending) ? PH_FLUSH : PH_DONE; return; }
    ps.non_specular = ps.non_specular || !ps.bdelta;
    f3 t = cross3(ps.n, ps.s);
    ps.d = fma3(ps.s, wo.x, fma3(t, wo.y, ps.n * wo.z));
    ps.tmin = ray_eps_closest(ps.o);
    ps.tmax = INFINITY;
    ps.has_bounce = true;
    ps.phase = PH_CLOSEST;
    ps.nrays++;
}

// PSSMLTSampler (src/integrators/pssmlt/pssmlt_sampler.cpp:93-168, pssmlt_sampler.h:113-143) as a pure function of the
// addressed stream: at the first primarySample of a mutation the reference rewrites the whole vector in order --
// components that already exist are mutated (Kelemen: one draw, toroidal wrap; Gaussian: two draws, modulo 1) or
// redrawn (large step: one draw), components that do not exist yet (beyond what the seed path consumed; only possible
// in a chain's first mutation) are appended as fresh uniforms (one draw) and survive a rejection.
struct PssmltSampler {
    uint32_t key0, key1, chain, major;
    bool large, kelemen;
    float sigma;
    uint32_t lane;
    uint32_t n_exist; // components that exist before this mutation
    u4 b1;
    uint32_t b1_idx;
    DEV void reset_caches() { b1_idx = 0xffffffffu; }
    DEV float u_s1(uint32_t idx) {
        uint32_t blk = idx >> 2;
        if (blk != b1_idx) { b1 = philox4x32_10(key0, key1, blk, major, chain, TAG_S1); b1_idx = blk; }
        return pick4(b1, idx & 3u);
    }
    DEV float x(uint32_t k) const { return lds_x[k * 64u + lane]; }
    DEV float next(uint32_t k) {
        const uint32_t per = (large || kelemen) ? 1u : 2u; // draws taken by an existing component
        if (k >= n_exist) return u_s1(per * n_exist + (k - n_exist));
        if (large) return u_s1(k);
        float value = x(k);
        if (kelemen) {
            float xi = u_s1(k);
            const bool add = xi < 0.5f;
            xi = add ? 2.f * xi : 2.f * (xi - 0.5f);
            const float dv = KELEMEN_S2 * fast_exp2(xi * LOG2_S1_OVER_S2);
            if (add) { value += dv; if (value > 1.f) value -= 1.f; }
            else { value -= dv; if (value < 0.f) value += 1.f; }
            return value;
        }
        const float v = value + gaussian_sample(u_s1(2u * k), u_s1(2u * k + 1u), sigma);
        return v - floorf(v); // math::modulo(v, 1)
    }
};

// Run one full PSS evaluation (one wave-divergent loop; every step issues at most one ray query).
template <class SamplerT> DEV DSplat eval_path(const DParams &P, SamplerT &smp, uint32_t &nrays, uint32_t &ndims) {
    PathState ps;
    smp.reset_caches();
    path_init(P, ps);
    Hit h{-1, 0.f, 0.f, 0.f};
    const GlobalTables T{P.shade, P.bsdfs, P.emitters};
    ShadowRay sr_unused;
    for (;;) {
        if (ps.phase != PH_BEGIN) h = trace(P, ps.o, ps.d, ps.tmin, ps.tmax, ps.phase == PH_SHADOW);
        path_step<false, 15>(P, T, ps, smp, h, false, sr_unused);
        if (ps.phase == PH_DONE) break;
    }
    DSplat out;
    out.px = ps.px; out.py = ps.py;
    out.r = ps.Li.x; out.g = ps.Li.y; out.b = ps.Li.z;
    out.lum = luminance3(ps.Li);
    nrays = ps.nrays;
    ndims = ps.k;
    return out;
}
