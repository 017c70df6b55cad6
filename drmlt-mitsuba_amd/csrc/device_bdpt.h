// technique=bdpt on the device: PathSampler::sampleSplats, EBidirectional branch (src/libbidir/pathsampler.cpp:321-527),
// with directSampling = true (P.bd_Dd != 0: the s = 1 / t = 1 strategies of :424-452 and the sampleDirect terms of
// Path::miWeight, path.cpp:799-824,936-1012) or false. Both random walks (with russian roulette from rrDepth) are stored, then
// every (s, t) pair is connected and weighted with Path::miWeight; the result is a splat LIST: the sensor-side pixel
// accumulates all t >= 2 strategies, every t = 1 strategy adds a light-image splat.
//
// Storage per chain:
//   global workspace `bd_verts`  [chain][NVS] records of BR_FLOATS floats (80 B): every stored vertex, contiguous -- written by
//                                the chain's lane during the walks, read by WHICHEVER lane connects a pair of them
//                                behind them [NVS][chain] fp64: the part of Path::miWeight's sum that lies beyond each vertex
//   LDS rows from `mis_row`      2 x NVS rows           fwd / rev densities (area measure) -- what the MIS weight reads for every pair
//   global lists `bd_lists`      [slot][BL_ROWS][n]     splat lists of the current state and the two proposals
// NVS = ME + MS vertex slots: emitter vertices 1..ME (ME = maxDepth), sensor vertices 1..MS (MS = maxDepth + 1);
// the supernodes are implicit.
//
// Two phases per evaluation:
//   walks        lane = chain: both random walks, vertex records to the workspace, densities to the LDS rows;
//   connections  lane = CELL: the (s, t) pairs of all the wave's chains are laid out side by side -- chain after chain, each chain's
//                cells in the reference's order (s and t descending, pathsampler.cpp:373-380) -- and handed out 64 at a time, whole
//                chains per round. A cell's lane gathers the two vertex records, connects, traces the visibility ray and weights
//                with Path::miWeight; a chain's contributions are summed by a segmented scan whose tree depends on nothing but
//                the chain's own cell count (so a chain's result does not depend on its neighbours in the wave), its light-image
//                splats numbered by ballot + prefix count. The lockstep alternative -- the wave walks one (s, t) grid, every lane
//                taking the cells its own subpaths reach -- ran 44 grid iterations per evaluation with 17 of 64 lanes in each.
#pragma once
#include "device_bidir.h"

enum { BR_P = 0, BR_N = 3, BR_S = 6, BR_WI = 9, BR_LEN2 = 12, BR_COS = 13, BR_GINV = 14, BR_IDS = 15, BR_THR = 16, BR_SHADE = 19, BR_FLOATS = 20 };
#define BDPT_MAX_BSDFS 4096      // BR_IDS: kind | bsdf << 4 | (emitter + 1) << 16 (drmlt_create refuses scenes beyond these for technique=bdpt)
#define BDPT_MAX_EMITTERS 65534
enum { BL_LUM = 0, BL_META = 1, BL_MAIN = 2, BL_MORE = 7 }; // rows of a splat list; 5 rows (px, py, r, g, b) per splat
enum { MF_FWD = 0, MF_REV = 1, MF_GROUPS = 2 }; // LDS row groups. The edge factor len^2 / |cos cos| (read by the rare specular-chain correction only) sits in
// the vertex record, the two flag bits per vertex in a 64-bit register.
#define BF_CONN 1u
#define BF_DEGEN 2u

__host__ __device__ inline int bdpt_list_rows(int max_depth) { return BL_MORE + 5 * max_depth; }
__host__ __device__ inline int bdpt_dims_sensor(int max_depth, int rr_depth) {
    int rr = max_depth + 1 - (rr_depth > 0 ? rr_depth : 0);
    int d = 2 * (max_depth + 1) + (rr > 0 ? rr : 0);
    return d + (d & 1);
}
__host__ __device__ inline int bdpt_dims_emitter(int max_depth, int rr_depth) {
    int rr = max_depth + 1 - (rr_depth > 0 ? rr_depth : 0);
    int d = 2 * max_depth + (rr > 1 ? rr - 1 : 0);
    return d + (d & 1);
}
// components of the direct sampler with directSampling = true: two per s = 1 (t = 2 .. maxDepth) and t = 1 (s = 2 .. maxDepth)
// connection -- the size at which no chain can overrun it (the reference's maxDepth, pssmlt_utils.h:75, is too small)
__host__ __device__ inline int bdpt_dims_direct(int max_depth) { return 2 * (2 * max_depth - 1); }
__host__ __device__ inline int bdpt_max_dim(int max_depth, int rr_depth) { // pssmlt_utils.h:69-75
    int d = (max_depth + 2) * (2 + (rr_depth < max_depth ? 1 : 0));
    return d + (d & 1);
}
// LDS floats of an evaluation beside the sampler rows: the two density row groups and the 64 segment heads of a connection round
__host__ __device__ inline int bdpt_eval_lds_floats(int max_depth) { return (2 * (2 * max_depth + 1) + 1) * 64; }

struct BdptResult {
    float lum;
    uint32_t nrays, n_sensor, n_emitter, n_direct;
    int n_more;
    bool has_main;
};

struct BdptStore {
    float *verts;     // P.bd_verts
    uint32_t NVS, n;  // n: columns of the arrays behind the records (P.n_chains_alloc)
    // behind the records: the MIS sums of the subpaths beyond each vertex (below), [slot][chain], fp64
    DEV double *tails(uint32_t chain) const { return reinterpret_cast<double *>(verts + (size_t) n * NVS * BR_FLOATS) + chain; }
    // ... and behind those the emitter samples of the s = 1 strategies, rows 2t and 2t + 1 for sensor vertex t, [row][chain]
    DEV float *direct_samples(uint32_t chain) const { return verts + (size_t) n * NVS * (BR_FLOATS + 2) + chain; }
    DEV float *rec(uint32_t chain, int slot) const { return verts + ((size_t) chain * NVS + (uint32_t) slot) * BR_FLOATS; }
    DEV float f(uint32_t chain, int slot, int field) const { return rec(chain, slot)[field]; }
    DEV f3 pos(uint32_t chain, int slot) const { const float *r = rec(chain, slot); return mk3(r[BR_P], r[BR_P + 1], r[BR_P + 2]); }
    DEV f3 nrm(uint32_t chain, int slot) const { const float *r = rec(chain, slot); return mk3(r[BR_N], r[BR_N + 1], r[BR_N + 2]); }
    DEV void put(uint32_t chain, int slot, const BVert &v, f3 thr, float ginv) const {
        float4 *r = reinterpret_cast<float4 *>(rec(chain, slot));
        r[0] = make_float4(v.p.x, v.p.y, v.p.z, v.n.x);
        r[1] = make_float4(v.n.y, v.n.z, v.s.x, v.s.y);
        r[2] = make_float4(v.s.z, v.wi.x, v.wi.y, v.wi.z);
        r[3] = make_float4(v.e_len2, v.e_cos, ginv, __int_as_float(v.kind | (v.bsdf << 4) | ((v.emitter + 1) << 16)));
        r[4] = make_float4(thr.x, thr.y, thr.z, __int_as_float(v.shade));
    }
    DEV void get(uint32_t chain, int slot, BVert &v, f3 &thr, float &ginv) const {
        const float4 *r = reinterpret_cast<const float4 *>(rec(chain, slot));
        const float4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4];
        v.p = mk3(a.x, a.y, a.z); v.n = mk3(a.w, b.x, b.y); v.s = mk3(b.z, b.w, c.x); v.wi = mk3(c.y, c.z, c.w);
        v.e_len2 = d.x; v.e_cos = d.y; ginv = d.z;
        const int ids = __float_as_int(d.w);
        v.kind = ids & 15; v.bsdf = (ids >> 4) & 4095; v.emitter = (int) ((uint32_t) ids >> 16) - 1;
        thr = mk3(e.x, e.y, e.z);
        v.shade = __float_as_int(e.w);
    }
};

// Scene::pdfEmitterDirect under the AREA measure -- what PathVertex::evalPdfDirect(sample, EImportance, EArea) asks for
// in Path::miWeight (vertex.cpp:1355-1382, scene.cpp:1057-1060, area.cpp:180-189, shape.cpp:118-127, sphere.cpp:356-385):
// the density with which direct sampling from `ref` produces the emitter point (sp, sn)
template <class TablesT>
DEV float emitter_direct_pdf_area(const TablesT &T, f3 ref_p, f3 ref_n, bool refn_zero, f3 sp, f3 sn, int emitter) {
    const DEmitter E = T.emitter(emitter);
    const DShade L = T.emitter_shade(emitter, E);
    f3 d = sp - ref_p;
    const float dist2 = dot3(d, d);
    d = d * rsqrtf(dist2);
    float pdf = 0.f;
    const float dr = refn_zero ? 0.f : dot3(d, ref_n), dl = dot3(d, sn);
    if (dr >= 0.f && dl < 0.f) {
        pdf = L.inv_area;
        if ((L.bsdf >> 24) == PRIM_SPHERE) {
            const f3 rc = ld3(L.origin) - ref_p;
            const float sinAlpha = L.eu[0] * rsqrtf(dot3(rc, rc));
            if (sinAlpha < 1.f - EPSILON_F) pdf = 0.15915494309189535f / (1.f - sqrtf(fmaxf(0.f, 1.f - sinAlpha * sinAlpha))) * fabsf(dl) / dist2;
        }
    }
    return pdf * (E.cdf_hi - E.cdf_lo);
}

// One evaluation per lane whose `active` is set; EVERY lane of the wave must make the call (the connection phase hands the cells
// of all chains out to all lanes). `chain`: workspace column; `list`: this lane's column of the target splat list (row r at
// list[r * n]). FEAT: as for trace() -- 15 with BVH traversal (and its 6 KB LDS stack), 7 without.
DEV unsigned long long shfl_u64(unsigned long long v, uint32_t src) {
    const uint32_t lo = (uint32_t) __shfl((int) (uint32_t) v, (int) src, 64), hi = (uint32_t) __shfl((int) (uint32_t) (v >> 32), (int) src, 64);
    return ((unsigned long long) hi << 32) | lo;
}
// cells of row s of a chain whose sensor subpath has nS vertices: t = maxT .. maxT - count + 1 (pathsampler.cpp:373-380;
// t = 0 needs a sensor that can be hit: a pinhole cannot, vertex.cpp:1405-1413)
DEV int bdpt_row_cells(const DParams &P, int s, int nS, int &maxT) {
    int minT = max(2 - s, P.light_image ? 0 : 2);
    if (minT < 1) minT = 1;
    maxT = min(min(P.max_depth + 1, P.max_depth + 1 - s), nS - 1);
    return max(0, maxT - minT + 1);
}

template <int FEAT = 15, class TablesT>
DEV void eval_bdpt(const DParams &P, const TablesT &T, MSampler &smp, bool active, uint32_t chain, uint32_t mis_row, float *list, BdptResult &R) {
    const uint32_t lane = smp.lane, n = P.n_chains_alloc;
    const int ME = P.max_depth, MS = P.max_depth + 1;
    const uint32_t NVS = (uint32_t) (ME + MS);
    const BdptStore W{P.bd_verts, NVS, n};
    auto mis = [&](int group, int slot) -> float & { return lds_x[(mis_row + (uint32_t) group * NVS + (uint32_t) slot) * 64u + lane]; };
    auto lrow = [&](int r) -> float & { return list[(size_t) r * n]; };
    // two flags per vertex slot (BF_CONN, BF_DEGEN), ONE BIT PER SLOT IN EACH OF TWO 64-bit words (round 4: one word of bit
    // pairs held 2 maxDepth + 1 <= 32 slots, i.e. maxDepth <= 15)
    static_assert(2 * BDPT_MAX_DEPTH + 1 <= 64, "one flag bit per vertex slot in a 64-bit register");
    unsigned long long connbits = 0ull, degenbits = 0ull;
    auto set_flags = [&](int slot, unsigned v) {
        const unsigned long long bit = 1ull << slot;
        connbits = (v & BF_CONN) ? connbits | bit : connbits & ~bit;
        degenbits = (v & BF_DEGEN) ? degenbits | bit : degenbits & ~bit;
    };

    R.lum = 0.f; R.nrays = 0u; R.n_sensor = R.n_emitter = R.n_direct = 0u; R.n_more = 0; R.has_main = false;
    smp.reset_caches();
    const bool stamps = (P.debug & 128) != 0; // diagnostic: per-wave cycles of the walks / the connections -> stats[16], [17]
    const unsigned long long st0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;

    // ------------------------------------------------------------ the two random walks (emitter first, :331-340)
    int nE = 1, nS = 1;            // vertices of each subpath, supernode included
    uint32_t emit_bits = 0u;       // bit t: sensor vertex t lies on an emitter (the s = 0 strategies that exist at all)
    float em0_fwd = 0.f;           // density of the emitter sample (area x emitter choice)
    float film_x = 0.f, film_y = 0.f;
    if (active) {
        BVert cur;
        cur.kind = BK_SUPER_E; cur.p = cur.n = cur.s = cur.wi = mk3(0.f, 0.f, 0.f);
        cur.e_len2 = cur.e_cos = 0.f; cur.bsdf = 0; cur.emitter = -1; cur.shade = 0; cur.degenerate = false;
        f3 thr = mk3(1.f, 1.f, 1.f);   // cumulative weight[mode] * rrWeight: importanceWeights / radianceWeights
        f3 thr_rr = thr;               // the walk's roulette throughput (tracks eta^2 as well, vertex.cpp:263-265)
        uint32_t kdim = 0u;
        smp.select(SEG_EMITTER);
#pragma nounroll
        for (int step = 0; step < ME + MS; ++step) {
            const bool emitter = step < ME;
            if (step == ME) {
                cur.kind = BK_SUPER_S; cur.degenerate = true; cur.e_len2 = 0.f;
                thr = thr_rr = mk3(1.f, 1.f, 1.f);
                kdim = 0u;
                smp.select(SEG_SENSOR);
            }
            const int i = emitter ? step : step - ME;          // this step samples from vertex i of its walk
            const int base = emitter ? 0 : ME;                 // slot of vertex v (>= 1) is base + v - 1
#define WALK_FAIL { if (emitter) { step = ME - 1; continue; } break; }
            const float u0 = smp.next(kdim), u1 = smp.next(kdim + 1u);
            kdim += 2u;
            if (emitter) R.n_emitter = kdim; else R.n_sensor = kdim;

            if (cur.kind == BK_SUPER_S) {
                cur.kind = BK_END_S; cur.p = cam_pos(P); cur.n = cam_dir(P); cur.degenerate = false; cur.e_len2 = 0.f; cur.e_cos = 0.f;
                cur.s = cur.wi = mk3(0.f, 0.f, 0.f);
                W.put(chain, base, cur, thr, 0.f);
                set_flags(base, 0u);
                nS = 2;
                continue;
            }
            if (cur.kind == BK_SUPER_E) {
                float sx = u0;
                int ei = 0;
                for (int q = 1; q < P.n_emitters; ++q)
                    if (T.emitter_cdf_lo(q) < sx) ei = q;
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
                em0_fwd = L.inv_area * emPdf;
                thr = thr * (ld3(E.radiance) * (PI_F / (L.inv_area * emPdf)));
                // the roulette throughput starts at the emitter sample: sampleNext returns before updating it (vertex.cpp:50-72)
                const bool sph = (L.bsdf >> 24) == PRIM_SPHERE;
                cur.kind = BK_END_E; cur.n = sph ? lp : ld3(L.n); cur.p = sph ? fma3(lp, L.eu[0], ld3(L.origin)) : lp; cur.emitter = ei; cur.shade = E.prim; cur.degenerate = false;
                cur.e_len2 = 0.f; cur.e_cos = 0.f; cur.s = cur.wi = mk3(0.f, 0.f, 0.f);
                W.put(chain, base, cur, thr, 0.f);
                set_flags(base, 0u);
                nE = 2;
                continue;
            }

            // ---- sample a direction at `cur` (vertex i >= 1)
            f3 d;
            f3 w = mk3(1.f, 1.f, 1.f);
            float pdf_fwd, pdf_rev = 1.f, eta2 = 1.f;
            bool delta = false;
            if (cur.kind == BK_END_E) {
                f3 fs, ft;
                frame_from_normal(cur.n, fs, ft);
                f3 l = square_to_cosine_hemisphere(u0, u1);
                d = fma3(fs, l.x, fma3(ft, l.y, cur.n * l.z));
                pdf_fwd = INV_PI_F * l.z;
            } else if (cur.kind == BK_END_S) {
                f3 nearP = mk3((1.f - 2.f * u0) * P.tan_half_fov * P.near_clip, (1.f - 2.f * u1) * P.tan_half_fov * P.inv_aspect * P.near_clip,
                               P.near_clip);
                f3 dl = normalize3(nearP);
                d = cam_to_world(P, dl);
                pdf_fwd = cam_normalization(P) / (dl.z * dl.z * dl.z);
                film_x = u0 * (float) P.width; film_y = u1 * (float) P.height;
            } else {
                const DBsdf B = T.bsdf(cur.bsdf);
                f3 wo;
                if (B.type == 0) {
                    if (!(cur.wi.z > 0.f)) WALK_FAIL;
                    wo = square_to_cosine_hemisphere(u0, u1);
                    pdf_fwd = INV_PI_F * wo.z;
                    w = ld3(B.rgb);
                } else if (B.type == 1) {
                    float cosThetaT;
                    float F = fresnel_dielectric_ext(cur.wi.z, cosThetaT, B.p[0]);
                    delta = true;
                    if (u0 <= F) {
                        wo = mk3(-cur.wi.x, -cur.wi.y, cur.wi.z);
                        pdf_fwd = F;
                    } else {
                        float scale = -(cosThetaT < 0.f ? B.p[1] : B.p[0]);
                        wo = mk3(scale * cur.wi.x, scale * cur.wi.y, cosThetaT);
                        pdf_fwd = 1.f - F;
                        float factor = emitter ? 1.f : (cosThetaT < 0.f ? B.p[1] : B.p[0]);
                        w = mk3(factor * factor, factor * factor, factor * factor);
                        const float e = cosThetaT < 0.f ? B.p[0] : B.p[1]; // bRec.eta
                        if (!emitter) eta2 = e * e;
                    }
                } else {
                    pdf_fwd = 0.f;
                    w = make_rc(B).sample(cur.wi, u0, u1, wo, pdf_fwd);
                }
                if (is_zero3(w)) WALK_FAIL;
                if (cur.wi.z == 0.f || wo.z == 0.f) WALK_FAIL;
                pdf_rev = delta ? dielectric_pdf_delta(B, wo, cur.wi) : bsdf_pdf_sa(B, wo, cur.wi);
                if (!(pdf_rev > 2.93873587705571876e-39f)) WALK_FAIL;
                d = fma3(cur.s, wo.x, fma3(cross3(cur.n, cur.s), wo.y, cur.n * wo.z));
            }
            // russian roulette of the random walk (path.cpp:515-518, vertex.cpp:310-324)
            thr_rr = thr_rr * (w * eta2);
            float rrw = 1.f;
            if (P.rr_depth != -1 && i >= P.rr_depth) {
                const float q = fminf(max3(thr_rr), 0.95f);
                const float ur = smp.next(kdim);
                kdim += 1u;
                if (emitter) R.n_emitter = kdim; else R.n_sensor = kdim;
                if (ur > q) WALK_FAIL;
                rrw = 1.f / q;
                thr_rr = thr_rr * rrw;
            }

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
            float fwd = pdf_fwd, rev = pdf_rev;
            if (!delta) {
                fwd = pdf_fwd * cosNew / len2;
                if (cur.e_len2 != 0.f) rev = pdf_rev * cur.e_cos / cur.e_len2;
            }
            // vertex i is now complete: densities and connectability
            const int cslot = base + i - 1;
            mis(MF_FWD, cslot) = fwd;
            mis(MF_REV, cslot) = rev;
            set_flags(cslot, ((!cur.degenerate && !delta) ? BF_CONN : 0u) | (cur.degenerate ? BF_DEGEN : 0u));
            thr = thr * (w * rrw);

            nv.kind = BK_SURF;
            nv.bsdf = Sh.bsdf & 0xffffff;
            nv.emitter = Sh.emitter;
            nv.shade = h.prim;
            {
                const int bt = T.bsdf(nv.bsdf).type;
                nv.degenerate = !(bt == 0 || bt == 2 || Sh.emitter >= 0);
            }
            nv.wi = to_local(nv, -d);
            nv.e_len2 = len2;
            nv.e_cos = cosCur;
            W.put(chain, base + i, nv, thr, len2 / (cosNew * cosCur)); // vertex i + 1, with the factor of edge (i, i + 1)
            set_flags(base + i, nv.degenerate ? BF_DEGEN : 0u);
            if (emitter) nE = i + 2; else { nS = i + 2; if (nv.emitter >= 0) emit_bits |= 1u << (i + 1); }
            cur = nv;
        }
#undef WALK_FAIL
    }

    const unsigned long long st1 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    // ------------------------------------------------------------ the splat list
    R.has_main = nS > 2; // "if (m_sensorSubpath.vertexCount() > 2)", :357-361
    const bool direct = P.bd_Dd != 0;
    auto refn_zero = [&](uint32_t ch, int slot) { return T.bsdf((__float_as_int(W.f(ch, slot, BR_IDS)) >> 4) & 4095).type == 1; }; // records.inl:160-164
    float re_walk = 0.f;     // ratioEmitterDirect of the emitter walk's own vertices 1 and 2 (every s >= 2 strategy)
    if (active && direct && nE >= 3)
        re_walk = emitter_direct_pdf_area(T, W.pos(chain, 1), W.nrm(chain, 1), refn_zero(chain, 1), W.pos(chain, 0), W.nrm(chain, 0),
                                          (int) ((uint32_t) __float_as_int(W.f(chain, 0, BR_IDS)) >> 16) - 1) / em0_fwd;
    // Path::miWeight (path.cpp:763-1028) sums, over every other strategy i of a path, the squared ratio of its density to the
    // connection's; the ratios are running products along the path, p_{i+1} = p_i x pdfImp[i] / pdfRad[i] towards the sensor and
    // the inverse towards the emitter. Two positions on either side of the connection edge involve the connection itself; beyond
    // them every factor belongs to ONE subpath, so their part of the sum is  p^2 x tail  with a tail that is the same for every
    // connection made at that vertex:  tail[m] = r_m^2 (c_m + tail[m - 1])  (r: the ratio at vertex m, c: whether that strategy
    // counts -- both endpoints connectable, not the two-vertex sensor path unless there is a light image, the emitter's
    // direct-sampling ratio at position 1). One pass over each subpath here; a connection then costs four ratios and two tails
    // instead of a sweep over the whole path.
    if (active) {
        double *const tails = W.tails(chain);
        auto fl = [&](int slot) -> bool { return ((connbits >> slot) & 1ull) != 0ull; };
        const float re_e = (direct && fl(1)) ? re_walk : 0.f; // the sampleDirect ratio of the s >= 4 strategies, at position 1
        double acc = 0.0;
        for (int m = 1; m <= nE - 3; ++m) { // emitter vertex m = position m: pdfRad[m] / pdfImp[m] (path.cpp:868-898 for the specular neighbours)
            const bool cm = fl(m - 1), cm1 = m == 1 ? true : fl(m - 2), cp1 = fl(m);
            float num = mis(MF_REV, m);
            if (m >= 2 && cp1 && !cm) num *= W.f(chain, m, BR_GINV);
            float den = m == 1 ? em0_fwd : mis(MF_FWD, m - 2);
            if (m >= 2 && cm1 && !cm) den *= W.f(chain, m - 1, BR_GINV);
            const double r = (double) (num / den);
            double c = (cm1 && cm) ? 1.0 : 0.0; // strategy i = m - 1
            if (direct && m == 2) c *= (double) re_e * (double) re_e;
            acc = r * r * (c + acc);
            tails[(size_t) (m - 1) * n] = acc;
        }
        acc = 0.0;
        for (int m = 1; m <= nS - 3; ++m) { // sensor vertex m = position k - m: pdfImp / pdfRad
            const bool cw = fl(ME + m - 1), cwm = m == 1 ? false : fl(ME + m - 2), cwp = fl(ME + m);
            float num = mis(MF_REV, ME + m);
            if (m >= 2 && cwp && !cw) num *= W.f(chain, ME + m, BR_GINV);
            float den = m == 1 ? 1.f : mis(MF_FWD, ME + m - 2);
            if (m >= 2 && cwm && !cw) den *= W.f(chain, ME + m - 1, BR_GINV);
            const double r = (double) (num / den);
            const double c = (cw && cwm && (P.light_image || m - 1 > 1)) ? 1.0 : 0.0;
            acc = r * r * (c + acc);
            tails[(size_t) (ME + m - 1) * n] = acc;
        }
    }
    // The direct sampler's components (directSampling = true, :424-452): two per t = 1, s > 1 connection and then two per s = 1,
    // t > 1 connection whose vertex can be connected, in the reference's cell order. Which cells those are follows from the
    // subpaths alone, so the chain draws its s = 1 emitter samples HERE, in one uniform loop, and the cells pick them up (drawn
    // in a cell, the sampler -- every mode and kernel of it -- ran for a few lanes in nearly every round).
    uint32_t kd_total = 0u;
    if (active && direct && nE >= 2) {
        auto degen = [&](int slot) -> bool { return ((degenbits >> slot) & 1ull) != 0ull; };
        int mt;
        for (int s = nE - 1; s >= 2; --s) // t = 1 is the last cell of a row that reaches it (light image)
            if (P.light_image && bdpt_row_cells(P, s, nS, mt) > 0 && !degen(s - 1)) kd_total += 2u;
        const int rc = bdpt_row_cells(P, 1, nS, mt);
        float *const ds = W.direct_samples(chain);
        smp.select(SEG_DIRECT);
        for (int t = mt; t > mt - rc && t >= 2; --t) {
            if (degen(ME + t - 1)) continue;
            smp.boot_k = R.n_emitter + R.n_sensor + kd_total; // a replayed stream: emitter walk, sensor walk, then the direct components in order
            ds[(size_t) (2 * t) * n] = smp.next(kd_total);
            ds[(size_t) (2 * t + 1) * n] = smp.next(kd_total + 1u);
            kd_total += 2u;
        }
    }
    uint32_t cells = 0u;     // the (s, t) pairs this chain's subpaths reach
    if (active)
    {   // row s = 0 (the sensor subpath ends on an emitter, :381-395) has a cell only where it does: most of its vertices do not
        int mt;
        const int rc = bdpt_row_cells(P, 0, nS, mt);
        emit_bits &= rc > 0 ? ((2u << mt) - 1u) & ~((1u << (mt - rc + 1)) - 1u) : 0u;
        cells = (uint32_t) __popc(emit_bits);
        for (int s = 1; s <= nE - 1; ++s) cells += (uint32_t) bdpt_row_cells(P, s, nS, mt);
    }
    float total_lum = 0.f;
    f3 main_v = mk3(0.f, 0.f, 0.f);
    int n_more = 0;
    int *const head = reinterpret_cast<int *>(&lds_x[(mis_row + 2u * NVS) * 64u]);
    const unsigned long long lanes_below = (1ull << lane) - 1ull;

    // Rounds: chains c0 .. c1 - 1 are the next whole chains whose cells fit the 64 lanes; cell i of the round goes to lane i.
    // (Only beyond maxDepth 9 can one chain have more than 64 cells: it then gets rounds of its own, 64 of its cells at a time, and
    // joins the others with what is left -- the order of its sums still depends on its own cell count alone.)
    uint32_t done = 0u;      // cells of this chain that earlier rounds have dealt with
    for (uint32_t c0 = 0u; c0 < 64u;) {
        const uint32_t rem = lane >= c0 ? cells - done : 0u;
        uint32_t pre = rem; // inclusive prefix sum over the lanes
#pragma unroll
        for (uint32_t d = 1u; d < 64u; d <<= 1) {
            const uint32_t v = (uint32_t) __shfl_up((int) pre, d, 64);
            if (lane >= d) pre += v;
        }
        const uint32_t nfit = (uint32_t) __popcll(__ballot(lane >= c0 && pre <= 64u));
        const uint32_t c1 = c0 + nfit;
        const bool alone = nfit == 0u; // chain c0 by itself, 64 of its cells
        const uint32_t total = alone ? 64u : (uint32_t) __shfl((int) pre, (int) ((c1 - 1u) & 63u), 64);
        const uint32_t cnt = alone ? (lane == c0 ? 64u : 0u) : (lane < c1 ? rem : 0u);
        const bool owner = cnt != 0u; // this lane's chain is in the round
        const uint32_t start = alone ? 0u : pre - rem;
        const uint32_t progress = done | ((uint32_t) n_more << 8); // what the chain's cells continue from
        if (owner) done += cnt;
        c0 = c1;
        if (total == 0u) continue;

        // ---- lane -> cell: segment heads through LDS, the chain's data by cross-lane reads
        __builtin_amdgcn_wave_barrier();
        head[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if (owner) head[start] = (int) lane + 1;
        __builtin_amdgcn_wave_barrier();
        const unsigned long long H = __ballot(head[lane] != 0);
        const unsigned long long upto = ~0ull >> (63u - lane); // bits 0 .. lane
        const uint32_t hp = 63u - (uint32_t) __builtin_clzll((H & upto) | 1ull);
        const unsigned long long above = H & ~upto;
        const uint32_t seg_end = above ? (uint32_t) __builtin_ctzll(above) : total;
        const unsigned long long segmask = (seg_end >= 64u ? ~0ull : ((1ull << seg_end) - 1ull)) & ~((1ull << hp) - 1ull);
        const bool mine = lane < total;
        const uint32_t c = (uint32_t) max(head[hp] - 1, 0); // the lane of the chain this cell belongs to
        const uint32_t jj = lane - hp;                      // index of the cell within its chain
        const uint32_t wc = (uint32_t) __shfl((int) chain, (int) c, 64);
        const uint32_t nEnS = (uint32_t) __shfl((int) ((uint32_t) nE | ((uint32_t) nS << 8)), (int) c, 64);
        const int nEc = (int) (nEnS & 255u), nSc = (int) (nEnS >> 8);
        const uint32_t emit_c = (uint32_t) __shfl((int) emit_bits, (int) c, 64);
        const unsigned long long fbc = shfl_u64(connbits, c), fbd = shfl_u64(degenbits, c);
        const float em0_c = __shfl(em0_fwd, (int) c, 64), re_walk_c = __shfl(re_walk, (int) c, 64);
        float *const list_c = reinterpret_cast<float *>(shfl_u64((unsigned long long) list, c));
        const uint32_t prog_c = (uint32_t) __shfl((int) progress, (int) c, 64);
        auto flags = [&](int slot) -> unsigned { return ((unsigned) (fbc >> slot) & 1u) | (((unsigned) (fbd >> slot) & 1u) << 1); };
        auto misc = [&](int group, int slot) -> float { return lds_x[(mis_row + (uint32_t) group * NVS + (uint32_t) slot) * 64u + c]; };

        int s = 0, t = 0;
        {
            int j = (int) (jj + (prog_c & 255u));
            bool found = false;
            for (int ss = nEc - 1; ss >= 1; --ss) {
                int mt;
                const int rc = bdpt_row_cells(P, ss, nSc, mt);
                if (j < rc) { s = ss; t = mt - j; found = true; break; }
                j -= rc;
            }
            if (!found) { // row s = 0: the j-th emitter hit of the sensor subpath, t descending
                uint32_t eb = emit_c;
                for (; j > 0 && eb; --j) eb &= ~(0x80000000u >> __builtin_clz(eb));
                t = eb ? 31 - __builtin_clz(eb) : 2;
            }
        }
        f3 value = mk3(0.f, 0.f, 0.f);
        float light_x = 0.f, light_y = 0.f;
        bool produced = false, traced = false;
        if (mine) do {
            BVert vs, vt;
            f3 thr_s = mk3(1.f, 1.f, 1.f), thr_t;
            vs.p = vs.n = vs.s = vs.wi = mk3(0.f, 0.f, 0.f); vs.e_len2 = vs.e_cos = 0.f; vs.kind = BK_SURF; vs.bsdf = 0; vs.emitter = -1; vs.shade = 0;
            vs.degenerate = false;
            float ginv_s = 0.f, ginv_t;
            if (s >= 1) { W.get(wc, s - 1, vs, thr_s, ginv_s); vs.degenerate = (flags(s - 1) & BF_DEGEN) != 0u; }
            W.get(wc, ME + t - 1, vt, thr_t, ginv_t);
            // the subpaths' tails for the weight at the end: asked for now, so that they arrive behind the connection's arithmetic
            const double *const tails = W.tails(wc);
            const double tail_t = t >= 3 ? tails[(size_t) (ME + t - 3) * n] : 0.0, tail_s = s >= 3 ? tails[(size_t) (s - 3) * n] : 0.0;
            vt.degenerate = (flags(ME + t - 1) & BF_DEGEN) != 0u;
            const int k = s + t + 1, depth = s + t - 1;
            float geo = 1.f;
            float pc_i1, pc_i2, pc_r0 = 0.f, pc_r1 = 0.f;
            float re_s1 = 0.f;
            float em0 = em0_c;   // pImp[1]; the s = 1 direct strategy swaps in its own emitter sample
            if (s == 0) {
                if (vt.kind != BK_SURF || vt.emitter < 0) break;
                const DEmitter E = T.emitter(vt.emitter);
                const f3 wo = to_world(vt, vt.wi);
                const float dp = dot3(wo, vt.n);
                float r = dp < 0.f ? 0.f : INV_PI_F * dp;
                if (dp != 0.f) r /= fabsf(dp);
                value = thr_t * (ld3(E.radiance) * (PI_F * r));
                if (is_zero3(value)) break;
                pc_i1 = T.shade(vt.shade).inv_area * (E.cdf_hi - E.cdf_lo);
                pc_i2 = (dp < 0.f ? 0.f : INV_PI_F * dp) * vt.e_cos / vt.e_len2;
            } else {
                if (direct && s == 1 && t > 1) {
                    // s = 1, t > 1: the emitter vertex is drawn by direct sampling from vt (:424-437, vertex.cpp:1285-1346,
                    // scene.cpp:879-904 without the visibility test) and replaces the walk's vertex 1, in miWeight too
                    // (:486-503). It goes through the connection code below as a vertex `vs` whose weight makes that code's
                    // value radiance / (pdf_direct) * f * cos -- cells at s = 1 and cells at other s share one code path.
                    if (vt.degenerate) break;
                    const float *const ds = W.direct_samples(wc); // drawn by the chain after its walks
                    float sx = ds[(size_t) (2 * t) * n];
                    const float sy = ds[(size_t) (2 * t + 1) * n];
                    int ei = 0;
                    for (int q = 1; q < P.n_emitters; ++q)
                        if (T.emitter_cdf_lo(q) < sx) ei = q;
                    const DEmitter E = T.emitter(ei);
                    const float emPdf = E.cdf_hi - E.cdf_lo;
                    sx = (sx - E.cdf_lo) / emPdf;
                    const DShade L = T.emitter_shade(ei, E);
                    f3 ln = ld3(L.n), dd;
                    float dist, pdf;
                    if ((L.bsdf >> 24) == PRIM_SPHERE) {
                        sphere_sample_direct(ld3(L.origin), L.eu[0], L.inv_area, vt.p, sx, sy, dd, dist, ln, pdf);
                        vs.p = fma3(dd, dist, vt.p);
                    } else {
                        if ((L.bsdf >> 24) == PRIM_RECTANGLE) vs.p = fma3(ld3(L.eu), sx, fma3(ld3(L.ev), sy, ld3(L.origin)));
                        else { const float a = sqrtf(fmaxf(0.f, 1.f - sx)); vs.p = fma3(ld3(L.eu), 1.f - a, fma3(ld3(L.ev), a * sy, ld3(L.origin))); }
                        const f3 dv = vs.p - vt.p;
                        const float dist2 = dot3(dv, dv);
                        dist = sqrtf(dist2);
                        dd = dv * (1.f / dist);
                        const float cc = dot3(dd, ln);
                        pdf = cc != 0.f ? L.inv_area * dist2 / fabsf(cc) : 0.f;
                    }
                    const float dln = dot3(dd, ln);
                    const float dr = T.bsdf(vt.bsdf).type == 1 ? 0.f : dot3(dd, vt.n);
                    if (!(dr >= 0.f && dln < 0.f && pdf != 0.f)) break; // AreaLight::sampleDirect, area.cpp:164-178
                    if (!(dist > 0.f)) break;
                    vs.kind = BK_END_E; vs.n = ln; vs.s = vs.wi = mk3(0.f, 0.f, 0.f); vs.e_len2 = vs.e_cos = 0.f;
                    vs.bsdf = 0; vs.emitter = ei; vs.shade = E.prim; vs.degenerate = false;
                    // the connection computes thr_s * (1 / pi) * f * cos_s cos_t / len^2: make that radiance / pdf * f * cos_t
                    thr_s = ld3(E.radiance) * (PI_F * dist * dist / (pdf * emPdf * fabsf(dln)));
                    em0 = L.inv_area * emPdf;
                }
                // t = 1 with direct sampling: a pinhole's sampleDirect returns the point the sensor subpath's vertex 1 already is
                // (perspective.cpp:386-420) and the same value term by term; what remains is that it consumes two components
                // (counted by the chain, above)
                if (vs.degenerate || vt.degenerate) break;
                f3 dc = vt.p - vs.p;
                const float len2 = dot3(dc, dc);
                const float len = sqrtf(len2);
                if (len == 0.f) break;
                dc = dc * (1.f / len);
                const DBsdf Bs = T.bsdf(vs.bsdf), Bt = T.bsdf(vt.bsdf);
                value = thr_s * thr_t * vert_eval(P, Bs, vs, dc, true) * vert_eval(P, Bt, vt, -dc, false);
                if (is_zero3(value)) break;
                const Hit h = trace<FEAT>(P, vt.p, -dc, ray_eps_closest(vt.p), len * (1.f - SHADOW_EPSILON_F), true);
                traced = true;
                if (h.prim >= 0) { value = mk3(0.f, 0.f, 0.f); break; }
                const float cs = fabsf(dot3(vs.n, dc)), ct = fabsf(dot3(vt.n, dc));
                geo = cs * ct / len2;
                const f3 wos = to_local(vs, dc), wot = to_local(vt, -dc);
                pc_i1 = vert_pdf_sa(P, Bs, vs, vs.wi, wos, dc) * ct / len2;
                pc_r0 = vert_pdf_sa(P, Bt, vt, vt.wi, wot, -dc) * cs / len2;
                if (vt.kind == BK_END_S) pc_i2 = 1.f;
                else pc_i2 = bsdf_pdf_sa(Bt, wot, vt.wi) * ((wot.z == 0.f || vt.wi.z == 0.f) ? 0.f : 1.f) * vt.e_cos / vt.e_len2;
                if (vs.kind == BK_END_E) pc_r1 = 1.f;
                else pc_r1 = bsdf_pdf_sa(Bs, wos, vs.wi) * ((wos.z == 0.f || vs.wi.z == 0.f) ? 0.f : 1.f) * vs.e_cos / vs.e_len2;
                if (direct && s == 1 && t > 1) re_s1 = emitter_direct_pdf_area(T, vt.p, vt.n, Bt.type == 1, vs.p, vs.n, vs.emitter) / em0;
            }
            if (P.exclude_direct && depth <= 2) { value = mk3(0.f, 0.f, 0.f); break; }

            // ---- Path::miWeight over positions 0..k (emitter vertex j at j, sensor vertex j at k - j): the two positions on either
            // side of the connection edge explicitly, the rest of either subpath through its tail (above)
            auto cflag = [&](int slot) -> bool { return (flags(slot) & BF_CONN) != 0u; };
            // sampleDirect terms (path.cpp:799-824,936-965): the emitter's direct-sampling density relative to its area density
            // at position 1. The sensor's ratio is 1: a pinhole's direct density is discrete, its position density 1.
            const bool sd = direct && k > 3;
            const bool c2 = t >= 2 && cflag(ME + t - 2); // connectable(s + 2): sensor vertex t - 1
            float re = 0.f;
            double initial = 1.0;
            if (sd) {
                if (s == 1) { re = re_s1; initial = 1.0 / (double) re; }
                else if (s == 0) { if (c2) re = emitter_direct_pdf_area(T, W.pos(wc, ME + t - 2), W.nrm(wc, ME + t - 2), refn_zero(wc, ME + t - 2), vt.p, vt.n, vt.emitter) / pc_i1; }
                else if (s == 2 || cflag(1)) re = re_walk_c;
            }
            double weight = 1.0;
            { // towards the sensor: strategies i = s + 1, s + 2, then the tail of sensor vertex t - 2 (ratio in fp32, product in fp64, see device_bidir.h)
                const float den1 = t == 1 ? 1.f : misc(MF_FWD, ME + t - 2);
                double pdf = initial * (double) (pc_i1 / den1);
                const double v1 = (sd && s == 0) ? pdf * (double) re : pdf;
                if (c2 && (P.light_image || t - 1 > 1)) weight += v1 * v1;
                if (t >= 2) {
                    const bool c3 = t >= 3 && cflag(ME + t - 3); // connectable(s + 3): sensor vertex t - 2
                    float num = pc_i2;
                    if (t >= 3 && !c2) num *= ginv_t;
                    float den2 = t == 2 ? 1.f : misc(MF_FWD, ME + t - 3);
                    if (t >= 3 && c3 && !c2) den2 *= W.f(wc, ME + t - 2, BR_GINV);
                    pdf *= (double) (num / den2);
                    if (c2 && c3 && (P.light_image || t - 2 > 1)) weight += pdf * pdf;
                    weight += pdf * pdf * tail_t;
                }
            }
            if (s >= 1) { // towards the emitter: strategies i = s - 1, s - 2, then the tail of emitter vertex s - 2
                const bool cm1 = s == 1 ? true : cflag(s - 2); // connectable(s - 1)
                const float den1 = s == 1 ? em0 : misc(MF_FWD, s - 2);
                double pdf = initial * (double) (pc_r0 / den1);
                const double v1 = (sd && s == 2) ? pdf * (double) re : pdf;
                if (cm1) weight += v1 * v1;
                if (s >= 2) {
                    const bool cm2 = s == 2 ? true : cflag(s - 3); // connectable(s - 2)
                    float num = pc_r1;
                    if (s >= 3 && !cm1) num *= ginv_s;
                    float den2 = s == 2 ? em0 : misc(MF_FWD, s - 3);
                    if (s >= 3 && cm2 && !cm1) den2 *= W.f(wc, s - 2, BR_GINV);
                    pdf *= (double) (num / den2);
                    const double v2 = (sd && s == 3) ? pdf * (double) re : pdf;
                    if (cm2 && cm1) weight += v2 * v2;
                    weight += pdf * pdf * tail_s;
                }
            }
            value = value * (geo / (float) weight);
            if (t == 1 && !cam_sample_position(P, vs.p - vt.p, light_x, light_y)) { value = mk3(0.f, 0.f, 0.f); break; } // light image: its own splat (:514-516)
            produced = true;
        } while (0);
        if (!produced) value = mk3(0.f, 0.f, 0.f);

        // ---- a chain's results: light-image splats numbered in cell order, everything else summed over its segment
        const bool is_light = produced && t == 1;
        const unsigned long long LB = __ballot(is_light), TB = __ballot(traced);
        if (is_light) {
            const int idx = (int) ((prog_c >> 8) & 255u) + (int) __popcll(LB & segmask & lanes_below);
            if (idx < P.max_depth) {
                float *l = list_c + (size_t) (BL_MORE + 5 * idx) * n;
                l[0] = light_x; l[n] = light_y; l[2 * (size_t) n] = value.x; l[3 * (size_t) n] = value.y; l[4 * (size_t) n] = value.z;
            }
        }
        float sv0 = is_light ? 0.f : value.x, sv1 = is_light ? 0.f : value.y, sv2 = is_light ? 0.f : value.z, sv3 = luminance3(value);
#pragma unroll
        for (uint32_t d = 1u; d < 64u; d <<= 1) { // (a chain has at most 45 cells; the tree of a segment's last element depends on its length alone)
            const float u0 = __shfl_up(sv0, d, 64), u1 = __shfl_up(sv1, d, 64), u2 = __shfl_up(sv2, d, 64), u3 = __shfl_up(sv3, d, 64);
            if (jj >= d) { sv0 += u0; sv1 += u1; sv2 += u2; sv3 += u3; }
        }
        const int last = (int) ((start + cnt - 1u) & 63u);
        const float t0 = __shfl(sv0, last, 64), t1 = __shfl(sv1, last, 64), t2 = __shfl(sv2, last, 64), t3 = __shfl(sv3, last, 64);
        if (owner) {
            const unsigned long long my = (start + cnt >= 64u ? ~0ull : ((1ull << (start + cnt)) - 1ull)) & ~((1ull << start) - 1ull);
            main_v = main_v + mk3(t0, t1, t2);
            total_lum += t3;
            n_more = min(n_more + (int) __popcll(LB & my), P.max_depth);
            R.nrays += (uint32_t) __popcll(TB & my);
        }
    }
    if (stamps) {
        const unsigned long long st2 = __builtin_amdgcn_s_memtime();
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) { // first active lane
            atomicAdd(P.stats + 16, st1 - st0); atomicAdd(P.stats + 17, st2 - st1); atomicAdd(P.stats + 19, 1ull);
        }
    }
    if (active) {
        lrow(BL_LUM) = total_lum;
        lrow(BL_META) = __int_as_float((R.has_main ? 1 : 0) | (n_more << 1));
        lrow(BL_MAIN) = R.has_main ? film_x : 0.f; lrow(BL_MAIN + 1) = R.has_main ? film_y : 0.f;
        lrow(BL_MAIN + 2) = main_v.x; lrow(BL_MAIN + 3) = main_v.y; lrow(BL_MAIN + 4) = main_v.z;
    }
    R.lum = total_lum;
    R.n_more = n_more;
    R.n_direct = kd_total;
}
