// technique=mmlt on the device: the multiplexed estimator of PathSampler::sampleSplats (EMMLT branch) and the
// three-segment PSS sampler of its chains. One chain per lane; the two random walks run through ONE step site
// (one ray query, one BSDF sampling site), the MIS sweep keeps its per-vertex densities in LDS rows.
//
// Reference behaviour restated here (paths relative to the reference checkout):
//   sampleSplats, EMMLT          src/libbidir/pathsampler.cpp:84-320
//   Path::randomWalk             src/libbidir/path.cpp:500-535
//   PathVertex::sampleNext       src/libbidir/vertex.cpp:37-350   (no media, rrStart never reached)
//   PathEdge::sampleNext         src/libbidir/edge.cpp:27-84
//   PathVertex::eval / evalPdf   src/libbidir/vertex.cpp:958-1205
//   PathVertex::cast             src/libbidir/vertex.cpp:1384-1404
//   pathConnectAndCollapse       src/libbidir/edge.cpp:558-690    (no null interactions)
//   PathEdge::evalCached         src/libbidir/edge.cpp:221-271    (EGeneralizedGeometricTerm)
//   Path::miWeight               src/libbidir/path.cpp:763-1028   (sampleDirect = false)
//   perspective sensor           src/sensors/perspective.cpp:191-245,299-372
//   area emitter                 src/emitters/area.cpp:96-150, src/librender/scene.cpp:1066-1087
//   sampler triple               src/integrators/drmlt/drmlt_proc.cpp:84-141, drmlt_sampler.cpp:112-177,313-394
#pragma once
#include "device_path.h"

enum { SEG_SENSOR = 0, SEG_EMITTER = 1, SEG_DIRECT = 2 };

// (depth + 2) * 3 rounded up to even (pssmlt_utils.h:58-63)
__host__ __device__ inline int mmlt_max_dim(int depth) {
    int d = (depth + 2) * 3;
    return d + (d & 1);
}

// ------------------------------------------------------------------ PSS sampler, three segments
// State rows in LDS: sensor [0, S), emitter [S, S + E), direct S + E (only the dimensions a path of the
// configured maxDepth can consume: S = 2 (maxDepth + 1), E = 2 maxDepth). Draws of a (tag, mutation) stream:
// sensor from index 0, emitter from 2 Dmax, direct from 4 Dmax, Dmax = mmlt_max_dim(maxDepth) -- the same
// addressing as the oracle's MMLTSamplers, so chains are comparable mutation by mutation.
struct MSampler {
    uint32_t key0, key1, chain, major;
    int mode, type;
    bool large;
    float sigma2;
    uint32_t lane;
    const float *x_dir;      // bdpt chains: the direct sampler's state stays in memory (column of this chain, row stride x_dir_n); else NULL
    uint32_t x_dir_n;
    const float *arr;        // SM_ARRAY: [sensor S | emitter E | direct]
    uint32_t S, E;
    uint32_t base_e, base_d; // draw bases of the emitter / direct segments
    bool emitter_ident2;     // fixEmitterPath and the current path is not pure light tracing (drmlt_proc.cpp:566-573)
    bool direct_ident;       // mmlt: the direct sampler's stages are identities (drmlt_proc.cpp:133-135); bdpt: an ordinary third sampler
    // active segment
    int seg;
    uint32_t x_off, draw_base;
    uint32_t boot_k;         // SM_BOOT: one replayable stream serves all three samplers in call order
    // caches
    u4 b1, b2;
    uint32_t b1_idx, b2_idx;
    uint32_t pair_base;
    float pair_y0, pair_y1, pair_z0, pair_z1;
    bool pair_has_z;

    DEV void reset_caches() { b1_idx = b2_idx = 0xffffffffu; pair_base = 0xffffffffu; boot_k = 0u; }
    DEV void select(int s) {
        seg = s;
        x_off = s == SEG_SENSOR ? 0u : (s == SEG_EMITTER ? S : S + E);
        draw_base = s == SEG_SENSOR ? 0u : (s == SEG_EMITTER ? base_e : base_d);
        pair_base = 0xffffffffu;
    }
    DEV float u_boot(uint32_t k) {
        uint32_t blk = k >> 2;
        if (blk != b1_idx) { b1 = philox4x32_10(key0, key1, blk, major, chain, TAG_BOOT); b1_idx = blk; }
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
    DEV float x(uint32_t k) const {
        if (x_dir && seg == SEG_DIRECT) return x_dir[(size_t) k * x_dir_n];
        return lds_x[(x_off + k) * 64u + lane];
    }
    DEV bool ident2() const { return (seg == SEG_DIRECT && direct_ident) || (seg == SEG_EMITTER && emitter_ident2); }

    DEV float y_raw(uint32_t k) {
        if (large) return u_s1(draw_base + k);             // the uniform branch comes first (drmlt_sampler.cpp:319-321)
        if (seg == SEG_DIRECT && direct_ident) return x(k); // identity stages (setStagesToIdentity)
        if (type != 2) return x(k) + kelemen_sample(u_s1(draw_base + k), KELEMEN_S2);
        ensure_pair(k & ~1u, false);
        return (k & 1u) ? pair_y1 : pair_y0;
    }
    DEV float z_raw(uint32_t k) {
        // second stage of a rejected LARGE step (timidAfterLarge; technique=bdpt only -- mmlt refuses it): fillSpace takes its uniform
        // branch again, before it looks at the kernel (drmlt_sampler.cpp:319-321 behind a debug-only assertion: DESIGN deviation 2)
        if (large) return u_s2(draw_base + k);
        if (ident2()) return x(k);
        if (type != 2) return x(k) + gaussian_sample(u_s2(draw_base + 2u * k), u_s2(draw_base + 2u * k + 1u), sigma2);
        ensure_pair(k & ~1u, true);
        return (k & 1u) ? pair_z1 : pair_z0;
    }
    DEV void ensure_pair(uint32_t k0, bool need_z) {
        if (pair_base != k0) {
            pair_base = k0;
            float x0 = x(k0), x1 = x(k0 + 1u);
            float d = kelemen_sample(u_s1(draw_base + k0), KELEMEN_S2 * ORBITAL_SCALE);
            float a = u_s1(draw_base + k0 + 1u);
            pair_y0 = fmaf(d, cos_rev(a), x0);
            pair_y1 = fmaf(d, sin_rev(a), x1);
            pair_has_z = false;
        }
        if (need_z && !pair_has_z) {
            pair_has_z = true;
            float x0 = x(k0), x1 = x(k0 + 1u);
            float xi = u_s2(draw_base + (k0 >> 1));
            float sign = 1.f;
            if (xi < 0.5f) { xi *= 2.f; } else { sign = -1.f; xi = 2.f * (xi - 0.5f); }
            float V = cos_rev(xi);
            float A = fminf(1.f, fmaxf(-1.f, (V + WC_DISPERSION) / (1.f + WC_DISPERSION * V)));
            float ct = A, st = sign * sqrtf(fmaxf(0.f, 1.f - A * A));
            float dx0 = x0 - pair_y0, dx1 = x1 - pair_y1;
            pair_z0 = pair_y0 + (ct * dx0 - st * dx1);
            pair_z1 = pair_y1 + (st * dx0 + ct * dx1);
        }
    }
    // primarySample(k) of the active segment
    DEV float next(uint32_t k) {
        switch (mode) {
            case SM_BOOT: return u_boot(boot_k++);
            case SM_ARRAY: return arr[x_off + k];
            case SM_STAGE1: return wrap01(y_raw(k));
            case SM_STAGE2: return wrap01(z_raw(k));
            default: { // Green reverse: y* = z - (y - x)
                float du = y_raw(k) - x(k);
                return wrap01(z_raw(k) - du);
            }
        }
    }
};

// ------------------------------------------------------------------ BSDF helpers (local frame)
DEV DRoughConductor make_rc(const DBsdf &B) {
    return DRoughConductor{DMicrofacet{B.p[7] != 0.f, fmaxf(B.p[0], 1e-4f)}, mk3(B.p[1], B.p[2], B.p[3]), mk3(B.p[4], B.p[5], B.p[6]),
                           ld3(B.rgb)};
}
// f * |cos wo| under the solid-angle measure (delta BSDFs: 0)
DEV f3 bsdf_eval_sa(const DBsdf &B, f3 wi, f3 wo) {
    if (B.type == 0) {
        if (!(wi.z > 0.f && wo.z > 0.f)) return mk3(0.f, 0.f, 0.f);
        return ld3(B.rgb) * (INV_PI_F * wo.z);
    }
    if (B.type == 2) return make_rc(B).eval(wi, wo);
    return mk3(0.f, 0.f, 0.f);
}
DEV float bsdf_pdf_sa(const DBsdf &B, f3 wi, f3 wo) {
    if (B.type == 0) return (wi.z > 0.f && wo.z > 0.f) ? INV_PI_F * wo.z : 0.f;
    if (B.type == 2) return make_rc(B).pdf(wi, wo);
    return 0.f;
}
// dielectric.cpp:255-276: discrete-measure pdf of wo given wi
DEV float dielectric_pdf_delta(const DBsdf &B, f3 wi, f3 wo) {
    float cosThetaT;
    float F = fresnel_dielectric_ext(wi.z, cosThetaT, B.p[0]);
    if (wi.z * wo.z >= 0.f) {
        if (fabsf(dot3(mk3(-wi.x, -wi.y, wi.z), wo) - 1.f) > 1e-3f) return 0.f; // DeltaEpsilon
        return F;
    }
    float scale = -(cosThetaT < 0.f ? B.p[1] : B.p[0]);
    if (fabsf(dot3(mk3(scale * wi.x, scale * wi.y, cosThetaT), wo) - 1.f) > 1e-3f) return 0.f;
    return 1.f - F;
}

// ------------------------------------------------------------------ sensor helpers
DEV f3 cam_pos(const DParams &P) { return mk3(P.cam[3], P.cam[7], P.cam[11]); }
DEV f3 cam_dir(const DParams &P) { return mk3(P.cam[2], P.cam[6], P.cam[10]); }
DEV f3 to_camera(const DParams &P, f3 d) { // inverse rotation
    return mk3(fmaf(P.cam[0], d.x, fmaf(P.cam[4], d.y, P.cam[8] * d.z)), fmaf(P.cam[1], d.x, fmaf(P.cam[5], d.y, P.cam[9] * d.z)),
               fmaf(P.cam[2], d.x, fmaf(P.cam[6], d.y, P.cam[10] * d.z)));
}
DEV f3 cam_to_world(const DParams &P, f3 d) {
    return mk3(fmaf(P.cam[0], d.x, fmaf(P.cam[1], d.y, P.cam[2] * d.z)), fmaf(P.cam[4], d.x, fmaf(P.cam[5], d.y, P.cam[6] * d.z)),
               fmaf(P.cam[8], d.x, fmaf(P.cam[9], d.y, P.cam[10] * d.z)));
}
DEV float cam_normalization(const DParams &P) { // 1 / area of the image rectangle at z = 1
    return 1.f / (4.f * P.tan_half_fov * P.tan_half_fov * P.inv_aspect);
}
DEV float cam_importance(const DParams &P, f3 dl) { // perspective.cpp:191-245
    if (!(dl.z > 0.f)) return 0.f;
    float inv = 1.f / dl.z;
    float px = dl.x * inv, py = dl.y * inv;
    float hx = P.tan_half_fov, hy = P.tan_half_fov * P.inv_aspect;
    if (px < -hx || px > hx || py < -hy || py > hy) return 0.f;
    return cam_normalization(P) * inv * inv * inv;
}
DEV bool cam_sample_position(const DParams &P, f3 dWorld, float &sx, float &sy) { // getSamplePosition, :355-372
    f3 l = to_camera(P, dWorld);
    if (!(l.z > 0.f)) return false;
    float u = 0.5f - 0.5f * (l.x / l.z) / P.tan_half_fov;
    float v = 0.5f - 0.5f * (l.y / l.z) / (P.tan_half_fov * P.inv_aspect);
    if (u < 0.f || u > 1.f || v < 0.f || v > 1.f) return false;
    sx = u * (float) P.width; sy = v * (float) P.height;
    return true;
}
// util.cpp:606-616 + Frame(n)
DEV void frame_from_normal(f3 n, f3 &s, f3 &t) {
    if (fabsf(n.x) > fabsf(n.y)) {
        float inv = rsqrtf(n.x * n.x + n.z * n.z);
        t = mk3(n.z * inv, 0.f, -n.x * inv);
    } else {
        float inv = rsqrtf(n.y * n.y + n.z * n.z);
        t = mk3(0.f, n.z * inv, -n.y * inv);
    }
    s = cross3(t, n);
}

// ------------------------------------------------------------------ path vertices
enum { BK_SUPER_S = 0, BK_SUPER_E = 1, BK_END_S = 2, BK_END_E = 3, BK_SURF = 4 };

struct BVert {
    f3 p, n, s;      // position, normal (geometric = shading for every supported primitive), frame axis s (t = n x s)
    f3 wi;           // local direction towards the predecessor (surface vertices)
    float e_len2;    // squared length of the edge to the predecessor (0: supernode edge)
    float e_cos;     // |cos| of that edge at the predecessor
    int kind, bsdf, emitter, shade;
    bool degenerate;
};

struct MmltResult {
    DSplat splat;
    uint32_t nrays;
    int s, t;
    uint32_t n_sensor, n_emitter, n_direct; // PSS components consumed per sampler
};

DEV f3 to_local(const BVert &v, f3 d) { return mk3(dot3(d, v.s), dot3(d, cross3(v.n, v.s)), dot3(d, v.n)); }
DEV f3 to_world(const BVert &v, f3 l) { return fma3(v.s, l.x, fma3(cross3(v.n, v.s), l.y, v.n * l.z)); }

// PathVertex::eval towards world direction `wo` (unit): divided by |cos| as the reference does
DEV f3 vert_eval(const DParams &P, const DBsdf &B, const BVert &v, f3 wo, bool importance) {
    if (v.kind == BK_END_E) { // area.cpp:132-140 / vertex.cpp:984-1001
        float dp = dot3(wo, v.n);
        float r = dp < 0.f ? 0.f : INV_PI_F * dp;
        if (dp != 0.f) r /= fabsf(dp);
        return mk3(r, r, r);
    }
    if (v.kind == BK_END_S) { // vertex.cpp:1003-1022
        float r = cam_importance(P, to_camera(P, wo));
        float dp = fabsf(dot3(v.n, wo));
        if (dp != 0.f) r /= dp;
        return mk3(r, r, r);
    }
    f3 wol = to_local(v, wo);
    // light-leak test of vertex.cpp:1043-1049: geometric and shading normal coincide here
    if (v.wi.z == 0.f || wol.z == 0.f) return mk3(0.f, 0.f, 0.f);
    f3 r = bsdf_eval_sa(B, v.wi, wol);
    (void) importance; // the adjoint correction |cos wi * (wo.ng)| / |cos wo * (wi.ng)| is exactly 1 for ng == ns
    return r * (1.f / fabsf(wol.z));
}
// solid-angle density of leaving `v` towards `wo` having arrived from direction `wi` (both local)
DEV float vert_pdf_sa(const DParams &P, const DBsdf &B, const BVert &v, f3 wi, f3 wo_local, f3 wo_world) {
    if (v.kind == BK_END_E) { float dp = dot3(wo_world, v.n); return dp < 0.f ? 0.f : INV_PI_F * dp; }
    if (v.kind == BK_END_S) return cam_importance(P, to_camera(P, wo_world));
    if (wi.z == 0.f || wo_local.z == 0.f) return 0.f;
    return bsdf_pdf_sa(B, wi, wo_local);
}

// ------------------------------------------------------------------ the estimator
// LDS rows used for the MIS sweep, per lane: pImp[NV], pRad[NV], gInv[NV] from row `mis_row`, NV = maxDepth + 3.
// FEAT: as for trace() -- 15 with BVH traversal (its LDS stack, its registers), 7 without
template <int FEAT = 15, class TablesT>
DEV void eval_mmlt(const DParams &P, const TablesT &T, MSampler &smp, int depth, uint32_t mis_row, MmltResult &R) {
    const uint32_t lane = smp.lane;
    const uint32_t NV = (uint32_t) P.max_depth + 3u;
    auto pImp = [&](int j) -> float & { return lds_x[(mis_row + (uint32_t) j) * 64u + lane]; };
    auto pRad = [&](int j) -> float & { return lds_x[(mis_row + NV + (uint32_t) j) * 64u + lane]; };
    auto gInv = [&](int j) -> float & { return lds_x[(mis_row + 2u * NV + (uint32_t) j) * 64u + lane]; };

    R.splat.lum = R.splat.px = R.splat.py = R.splat.r = R.splat.g = R.splat.b = 0.f;
    R.nrays = 0u; R.n_sensor = R.n_emitter = 0u; R.n_direct = 1u;
    smp.reset_caches();
    smp.select(SEG_DIRECT);
    const float decision = smp.next(0u);
    int nStrats, s, t;
    if (P.light_image) { nStrats = depth + 1; s = min((int) ((float) nStrats * decision), nStrats - 1); t = nStrats - s; }
    else { nStrats = depth; s = min((int) ((float) nStrats * decision), nStrats - 1); t = 1 + (nStrats - s); }
    R.s = s; R.t = t;
    if (depth == 1) return;
    const int k = s + t + 1;

    BVert cur, vt;
    cur.kind = BK_SUPER_S; cur.p = cur.n = cur.s = cur.wi = mk3(0.f, 0.f, 0.f);
    cur.e_len2 = 0.f; cur.e_cos = 0.f; cur.bsdf = 0; cur.emitter = -1; cur.shade = 0; cur.degenerate = true;
    vt = cur;
    f3 thr = mk3(1.f, 1.f, 1.f);
    uint32_t conn = 1u; // bit j: vertex j is connectable. Emitter supernode: area emitters only
    float film_x = 0.f, film_y = 0.f;
    int pos = k;
    uint32_t kdim = 0u;
    smp.select(SEG_SENSOR);

    // A walk that fails ends there, but the other walk is still made (pathsampler.cpp:140-160 checks both
    // lengths afterwards): the PSS components and rays it consumes are part of the chain's bookkeeping.
    bool failed = false;
#define WALK_FAIL { failed = true; if (sensor) { step = t - 1; continue; } break; }
#pragma nounroll
    for (int step = 0; step < s + t; ++step) {
        const bool sensor = step < t;
        if (step == t) { // sensor walk complete: park its end vertex, start at the emitter supernode
            vt = cur;
            cur.kind = BK_SUPER_E; cur.e_len2 = 0.f; cur.degenerate = false;
            pos = 0; kdim = 0u;
            smp.select(SEG_EMITTER);
        }
        const float u0 = smp.next(kdim), u1 = smp.next(kdim + 1u);
        kdim += 2u;
        if (sensor) R.n_sensor = kdim; else R.n_emitter = kdim;
        const int npos = sensor ? pos - 1 : pos + 1;

        if (cur.kind == BK_SUPER_S) { // perspective.cpp:299-307: pinhole position, discrete
            pRad(k - 1) = 1.f;
            cur.kind = BK_END_S; cur.p = cam_pos(P); cur.n = cam_dir(P); cur.degenerate = false; cur.e_len2 = 0.f;
            pos = npos;
            continue;
        }
        if (cur.kind == BK_SUPER_E) { // Scene::sampleEmitterPosition, scene.cpp:1066-1079
            float sx = u0;
            int ei = 0;
            for (int i = 1; i < P.n_emitters; ++i)
                if (T.emitter_cdf_lo(i) < sx) ei = i;
            const DEmitter E = T.emitter(ei);
            const float emPdf = E.cdf_hi - E.cdf_lo;
            sx = (sx - E.cdf_lo) / emPdf;
            const DShade L = T.emitter_shade(ei, E);
            f3 lp;
            if ((L.bsdf >> 24) == PRIM_RECTANGLE) lp = fma3(ld3(L.eu), sx, fma3(ld3(L.ev), u1, ld3(L.origin)));
            else if ((L.bsdf >> 24) == PRIM_SPHERE) { // sphere.cpp:257-268: uniform on the sphere
                const float z = 1.f - 2.f * u1, r = sqrtf(fmaxf(0.f, 1.f - z * z));
                lp = mk3(r * cos_rev(sx), r * sin_rev(sx), z); // unit normal for now, scaled below
            } else { float a = sqrtf(fmaxf(0.f, 1.f - sx)); lp = fma3(ld3(L.eu), 1.f - a, fma3(ld3(L.ev), a * u1, ld3(L.origin))); }
            pImp(1) = L.inv_area * emPdf;
            thr = thr * (ld3(E.radiance) * (PI_F / (L.inv_area * emPdf))); // m_power / emitter pdf (area.cpp:96-101)
            const bool sph = (L.bsdf >> 24) == PRIM_SPHERE;
            cur.kind = BK_END_E; cur.n = sph ? lp : ld3(L.n); cur.p = sph ? fma3(lp, L.eu[0], ld3(L.origin)) : lp; cur.emitter = ei; cur.shade = E.prim; cur.degenerate = false; cur.e_len2 = 0.f;
            pos = npos;
            continue;
        }

        // ---- sample a direction at `cur`
        f3 d;
        float pdf_fwd, pdf_rev = 1.f;
        bool delta = false;
        if (cur.kind == BK_END_E) { // area.cpp:117-130
            f3 fs, ft;
            frame_from_normal(cur.n, fs, ft);
            f3 l = square_to_cosine_hemisphere(u0, u1);
            d = fma3(fs, l.x, fma3(ft, l.y, cur.n * l.z));
            pdf_fwd = INV_PI_F * l.z;
            conn |= 1u << pos;
        } else if (cur.kind == BK_END_S) { // perspective.cpp:317-343
            f3 nearP = mk3((1.f - 2.f * u0) * P.tan_half_fov * P.near_clip, (1.f - 2.f * u1) * P.tan_half_fov * P.inv_aspect * P.near_clip,
                           P.near_clip);
            f3 dl = normalize3(nearP);
            d = cam_to_world(P, dl);
            pdf_fwd = cam_normalization(P) / (dl.z * dl.z * dl.z);
            conn |= 1u << pos;
        } else {
            const DBsdf B = T.bsdf(cur.bsdf);
            f3 wo;
            f3 w;
            if (B.type == 0) {
                if (!(cur.wi.z > 0.f)) WALK_FAIL;
                wo = square_to_cosine_hemisphere(u0, u1);
                pdf_fwd = INV_PI_F * wo.z;
                w = ld3(B.rgb);
            } else if (B.type == 1) { // dielectric.cpp:270-306; radiance scaling only in ERadiance mode
                float cosThetaT;
                float F = fresnel_dielectric_ext(cur.wi.z, cosThetaT, B.p[0]);
                delta = true;
                if (u0 <= F) {
                    wo = mk3(-cur.wi.x, -cur.wi.y, cur.wi.z);
                    pdf_fwd = F;
                    w = mk3(1.f, 1.f, 1.f);
                } else {
                    float scale = -(cosThetaT < 0.f ? B.p[1] : B.p[0]);
                    wo = mk3(scale * cur.wi.x, scale * cur.wi.y, cosThetaT);
                    pdf_fwd = 1.f - F;
                    float factor = sensor ? (cosThetaT < 0.f ? B.p[1] : B.p[0]) : 1.f;
                    w = mk3(factor * factor, factor * factor, factor * factor);
                }
            } else {
                pdf_fwd = 0.f;
                w = make_rc(B).sample(cur.wi, u0, u1, wo, pdf_fwd);
            }
            if (is_zero3(w)) WALK_FAIL;
            if (cur.wi.z == 0.f || wo.z == 0.f) WALK_FAIL; // vertex.cpp:206-211 with ng == ns
            pdf_rev = delta ? dielectric_pdf_delta(B, wo, cur.wi) : bsdf_pdf_sa(B, wo, cur.wi);
            if (!(pdf_rev > 2.93873587705571876e-39f)) WALK_FAIL; // RCPOVERFLOW, :236-239
            thr = thr * w;
            if (!cur.degenerate && !delta) conn |= 1u << pos;
            d = fma3(cur.s, wo.x, fma3(cross3(cur.n, cur.s), wo.y, cur.n * wo.z));
        }

        // ---- PathEdge::sampleNext: next surface along the ray
        const Hit h = trace<FEAT>(P, cur.p, d, ray_eps_closest(cur.p), INFINITY, false);
        R.nrays++;
        if (h.prim < 0) WALK_FAIL;
        const DShade Sh = T.shade(h.prim);
        BVert nv;
        if ((Sh.bsdf >> 24) != PRIM_SPHERE) {
            nv.p = fma3(ld3(Sh.eu), h.u, fma3(ld3(Sh.ev), h.v, ld3(Sh.origin)));
            nv.n = ld3(Sh.n);
            nv.s = ld3(Sh.eu) * Sh.inv_len_eu;
        } else {
            f3 c = ld3(Sh.origin);
            f3 local = normalize3(fma3(d, h.t, cur.p) - c);
            nv.p = fma3(local, Sh.eu[0], c);
            nv.n = local;
            float zrad2 = local.x * local.x + local.y * local.y;
            float inv = rsqrtf(zrad2);
            nv.s = zrad2 > 0.f ? mk3(-local.y * inv, local.x * inv, 0.f) : mk3(1.f, 0.f, 0.f);
        }
        if (h.t == 0.f) WALK_FAIL;
        const float len2 = h.t * h.t;
        const float cosNew = fabsf(dot3(d, nv.n)), cosCur = fabsf(dot3(d, cur.n));
        // solid angle -> area (vertex.cpp:332-347)
        float fwd = pdf_fwd, rev = pdf_rev;
        if (!delta) {
            fwd = pdf_fwd * cosNew / len2;
            if (cur.e_len2 != 0.f) rev = pdf_rev * cur.e_cos / cur.e_len2;
        }
        if (sensor) { pRad(npos) = fwd; if (cur.kind == BK_SURF || cur.kind == BK_END_S) pImp(pos + 1) = rev; }
        else { pImp(npos) = fwd; if (cur.kind == BK_SURF || cur.kind == BK_END_E) pRad(pos - 1) = rev; }
        gInv(sensor ? npos : pos) = len2 / (cosNew * cosCur);
        if (cur.kind == BK_END_S) { film_x = u0 * (float) P.width; film_y = u1 * (float) P.height; }

        nv.kind = BK_SURF;
        nv.bsdf = Sh.bsdf & 0xffffff;
        nv.emitter = Sh.emitter;
        nv.shade = h.prim;
        {
            const int bt = T.bsdf(nv.bsdf).type;
            nv.degenerate = !(bt == 0 || bt == 2 || Sh.emitter >= 0); // edge.cpp:66-69
        }
        nv.wi = to_local(nv, -d);
        nv.e_len2 = len2;
        nv.e_cos = cosCur;
        cur = nv;
        pos = npos;
    }

#undef WALK_FAIL
    if (failed) return;

    // ---- both walks complete
    BVert vs;
    if (s == 0) { vt = cur; vs.kind = BK_SUPER_E; vs.degenerate = false; vs.p = vs.n = vs.s = vs.wi = mk3(0.f, 0.f, 0.f); vs.e_len2 = vs.e_cos = 0.f; vs.bsdf = 0; vs.emitter = -1; vs.shade = 0; }
    else vs = cur;
    if (s >= 1 && !vs.degenerate) conn |= 1u << s;
    if (!vt.degenerate) conn |= 1u << (s + 1);
    // "Check if subpaths are connectable", pathsampler.cpp:161-173: emitter vertices 2 .. s and sensor vertices 2 .. t, i.e.
    // positions 2 .. k - 2 -- and position 1 when s = 0, where the sensor subpath's LAST vertex sits (the emitter it has hit: a
    // camera path that reaches a light over specular vertices only, E S* L, is kept by that vertex)
    {
        uint32_t inner = 0u;
        if (k - 2 >= 2) inner = ((1u << (k - 1)) - 1u) & ~3u;
        if (s == 0) inner |= 2u;
        if ((conn & inner) == 0u) return;
    }

    f3 value;
    float geo = 1.f;
    if (s == 0) {
        // PathVertex::cast(EEmitterSample): the sensor path must end on an emitter
        if (vt.kind != BK_SURF || vt.emitter < 0) return;
        const DEmitter E = T.emitter(vt.emitter);
        const f3 wo = to_world(vt, vt.wi); // unit direction towards vtPred
        float dp = dot3(wo, vt.n);
        float r = dp < 0.f ? 0.f : INV_PI_F * dp;
        if (dp != 0.f) r /= fabsf(dp);
        value = thr * (ld3(E.radiance) * (PI_F * r)); // evalPosition = radiance * pi (area.cpp:103-105)
        if (is_zero3(value)) return;
        // densities at the connection (emitter supernode -> vt -> vtPred)
        pImp(1) = T.shade(vt.shade).inv_area * (E.cdf_hi - E.cdf_lo);
        {
            float pd = dp < 0.f ? 0.f : INV_PI_F * dp; // cosine lobe towards vtPred
            pImp(2) = pd * vt.e_cos / vt.e_len2;
        }
        pRad(0) = 0.f; // evalPdf(..) * connectionEdge.pdf[ERadiance] = 0, never read (i >= s + 1)
    } else {
        if (vs.degenerate || vt.degenerate) return;
        f3 dc = vt.p - vs.p; // vs -> vt
        const float len2 = dot3(dc, dc);
        const float len = sqrtf(len2);
        if (len == 0.f) return;
        dc = dc * (1.f / len);
        const DBsdf Bs = T.bsdf(vs.bsdf), Bt = T.bsdf(vt.bsdf);
        value = thr * vert_eval(P, Bs, vs, dc, true) * vert_eval(P, Bt, vt, -dc, false);
        if (is_zero3(value)) return;
        // mutual visibility, ray from vt towards vs (edge.cpp:575-600)
        const Hit h = trace<FEAT>(P, vt.p, -dc, ray_eps_closest(vt.p), len * (1.f - SHADOW_EPSILON_F), true);
        R.nrays++;
        if (h.prim >= 0) return;
        const float cs = fabsf(dot3(vs.n, dc)), ct = fabsf(dot3(vt.n, dc));
        geo = cs * ct / len2;
        // densities at the connection
        const f3 wos = to_local(vs, dc), wot = to_local(vt, -dc);
        pImp(s + 1) = vert_pdf_sa(P, Bs, vs, vs.wi, wos, dc) * ct / len2;
        pRad(s) = vert_pdf_sa(P, Bt, vt, vt.wi, wot, -dc) * cs / len2;
        // reverse densities towards the predecessors
        if (vt.kind == BK_END_S) pImp(s + 2) = 1.f;
        else pImp(s + 2) = bsdf_pdf_sa(Bt, wot, vt.wi) * ((wot.z == 0.f || vt.wi.z == 0.f) ? 0.f : 1.f) * vt.e_cos / vt.e_len2;
        if (vs.kind == BK_END_E) pRad(s - 1) = 1.f;
        else pRad(s - 1) = bsdf_pdf_sa(Bs, wos, vs.wi) * ((wos.z == 0.f || vs.wi.z == 0.f) ? 0.f : 1.f) * vs.e_cos / vs.e_len2;
    }
    if (P.exclude_direct && depth <= 2) return; // pathsampler.cpp:274-280

    // ---- Path::miWeight
    pImp(0) = 1.f;
    pRad(k) = 1.f;
    for (int i = 1; i <= k - 3; ++i) { // densities next to specular chains: area -> projected solid angle
        if (i == s || !((conn >> i) & 1u) || ((conn >> (i + 1)) & 1u)) continue;
        pImp(i + 1) *= gInv(i);
    }
    for (int i = k - 1; i >= 3; --i) {
        if (i - 1 == s || !((conn >> i) & 1u) || ((conn >> (i - 1)) & 1u)) continue;
        pRad(i - 1) *= gInv(i - 1);
    }
    // The reference sweeps in fp64 (path.cpp:985-1024) so that products of many densities neither overflow nor vanish.
    // Each density RATIO is formed in fp32 here (one v_rcp instead of a software fp64 division per vertex; a ratio of
    // two fp32 densities fits fp32 with room to spare), the running products and their squares stay in fp64.
    double weight = 1.0, pdf = 1.0;
    for (int i = s + 1; i < k; ++i) {
        double next = pdf * (double) (pImp(i) / pRad(i));
        int tPrime = k - i - 1;
        if (((conn >> i) & 1u) && ((conn >> (i + 1)) & 1u) && (P.light_image || tPrime > 1)) weight += next * next;
        pdf = next;
    }
    pdf = 1.0;
    for (int i = s - 1; i >= 0; --i) {
        double next = pdf * (double) (pRad(i + 1) / pImp(i + 1));
        int tPrime = k - i - 1;
        if (((conn >> i) & 1u) && ((conn >> (i + 1)) & 1u) && (P.light_image || tPrime > 1)) weight += next * next;
        pdf = next;
    }
    const float miw = 1.f / (float) weight;
    value = value * (geo * miw * (float) nStrats);

    // ---- splat position
    float sx = 0.f, sy = 0.f;
    if (t >= 2) { sx = film_x; sy = film_y; }
    else if (!cam_sample_position(P, vs.p - vt.p, sx, sy)) return;
    R.splat.px = sx; R.splat.py = sy;
    R.splat.r = value.x; R.splat.g = value.y; R.splat.b = value.z;
    R.splat.lum = luminance3(value);
}
