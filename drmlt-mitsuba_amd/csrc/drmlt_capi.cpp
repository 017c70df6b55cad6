// C-ABI of the MI355X-native DRMLT hot path (include/drmlt_abi.h): context management,
// scene flattening, bootstrap/seeding host logic and kernel orchestration. Everything that
// touches path evaluation runs in the HIP kernels of kernels.hip; there is no CPU fallback.
#include "../../include/drmlt_abi.h"
#include "box_merge.h"
#include "bvh_build.h"
#include "device_types.h"
#include "drmlt_ctx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// launchers defined in kernels.hip
void launch_bootstrap(const DParams &P, uint32_t n, float *lum_out, hipStream_t st);
void launch_init_chains(const DParams &P, const uint32_t *seed_index, const float *seed_lum, hipStream_t st);
void launch_mutate(const DParams &P, uint32_t n_mut, uint32_t mut_base, hipStream_t st);
void launch_mutate_pssmlt(const DParams &P, uint32_t n_mut, uint32_t mut_base, hipStream_t st);
void launch_eval_paths(const DParams &P, const float *u, uint32_t n, uint32_t dim, float *out8, hipStream_t st);
// technique=mmlt (kernels_mmlt.hip)
void launch_bootstrap_mmlt(const DParams &P, uint32_t n, float *lum_out, hipStream_t st);
void launch_init_chains_mmlt(const DParams &P, const uint32_t *seed_index, const float *seed_lum, hipStream_t st);
void launch_mutate_mmlt(const DParams &P, uint32_t n_mut, uint32_t mut_base, hipStream_t st);
void launch_eval_paths_mmlt(const DParams &P, const float *u, uint32_t n, uint32_t dim, float *out8, hipStream_t st);
void launch_regroup(const uint32_t *work, const int32_t *depth_or_null, uint32_t n, uint32_t n_mut, uint32_t md, uint32_t *order, uint32_t padded, uint32_t *scratch, hipStream_t st);
size_t regroup_scratch_words(uint32_t n, uint32_t md);
// technique=bdpt (kernels_bdpt.hip)
void launch_bootstrap_bdpt(const DParams &P, uint32_t n, float *lum_out, hipStream_t st);
void launch_init_chains_bdpt(const DParams &P, const uint32_t *seed_index, const float *seed_lum, hipStream_t st);
void launch_mutate_bdpt(const DParams &P, uint32_t n_mut, uint32_t mut_base, hipStream_t st);
void launch_eval_lists_bdpt(const DParams &P, const float *u, uint32_t n, uint32_t dim, float *out, uint32_t stride, hipStream_t st);
void launch_render_pt(const DParams &P, uint64_t n_samples, uint32_t stream, float scale, hipStream_t st);

namespace {

constexpr uint32_t TAG_SEEDSEL = 1;

// host Philox4x32-10 (seed selection draws; same function as device_math.h)
void philox_host(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t) 0xD2511F53u * c0, p1 = (uint64_t) 0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t) (p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t) p1, n2 = (uint32_t) (p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t) p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

} // namespace

drmlt_ctx::~drmlt_ctx() {
    if (comm) drmlt_comm_release(comm);
    if (own_stream && stream) (void) hipStreamDestroy(stream);
}

namespace {

bool invert3x4(const double *m, double *o) {
    double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (det == 0 || !std::isfinite(det)) return false;
    double id = 1.0 / det;
    o[0] = (e * i - f * h) * id; o[1] = (c * h - b * i) * id; o[2] = (b * f - c * e) * id;
    o[4] = (f * g - d * i) * id; o[5] = (a * i - c * g) * id; o[6] = (c * d - a * f) * id;
    o[8] = (d * h - e * g) * id; o[9] = (b * g - a * h) * id; o[10] = (a * e - b * d) * id;
    for (int r = 0; r < 3; ++r) o[r * 4 + 3] = -(o[r * 4] * m[3] + o[r * 4 + 1] * m[7] + o[r * 4 + 2] * m[11]);
    return true;
}

// Flatten the scene into intersection + shading records. Returns "" or an error.
std::string build_scene(drmlt_ctx *ctx, const drmlt_scene &s, std::vector<DBsdf> &bsdfs, std::vector<DEmitter> &emitters,
                        std::vector<PrimBounds> &bounds, std::vector<QuadGeo> &geo) {
    if (s.n_shapes <= 0) return "scene has no shapes";
    if (s.n_emitters <= 0) return "scene has no emitters";
    for (int i = 0; i < s.n_bsdfs; ++i) {
        const drmlt_bsdf &in = s.bsdfs[i];
        DBsdf b{};
        b.type = in.type;
        for (int k = 0; k < 3; ++k) b.rgb[k] = in.rgb[k];
        if (in.type == DRMLT_BSDF_DIFFUSE) {
        } else if (in.type == DRMLT_BSDF_DIELECTRIC) {
            if (!(in.p[0] > 0.f) || !(in.p[1] > 0.f)) return "dielectric: IORs must be positive";
            b.p[0] = in.p[0] / in.p[1];
            b.p[1] = 1.f / b.p[0];
        } else if (in.type == DRMLT_BSDF_ROUGHCONDUCTOR) {
            if (!(in.p[0] > 0.f)) return "roughconductor: alpha must be positive";
            for (int k = 0; k < 8; ++k) b.p[k] = in.p[k];
        } else {
            return "unsupported BSDF type " + std::to_string(in.type) + " (supported: diffuse, dielectric, roughconductor)";
        }
        bsdfs.push_back(b);
    }
    ctx->prims.clear(); ctx->shade.clear();
    for (int i = 0; i < s.n_shapes; ++i) {
        const drmlt_shape &in = s.shapes[i];
        if (in.bsdf < 0 || in.bsdf >= s.n_bsdfs) return "shape references an invalid bsdf";
        if (in.emitter >= s.n_emitters) return "shape references an invalid emitter";
        DPrim g{};
        DShade sh{};
        PrimBounds pb;
        QuadGeo qg{};
        qg.usable = false;
        sh.bsdf = in.bsdf; // | kind << 24, set below
        sh.emitter = in.emitter < 0 ? -1 : in.emitter;
        double m[12], inv[12];
        if (in.type == DRMLT_SHAPE_TRIANGLE) {
            double p0[3], e1[3], e2[3], n[3];
            for (int k = 0; k < 3; ++k) { p0[k] = in.data[k]; e1[k] = (double) in.data[3 + k] - p0[k]; e2[k] = (double) in.data[6 + k] - p0[k]; }
            n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
            double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (!(len > 0)) return "degenerate triangle";
            for (int k = 0; k < 3; ++k) n[k] /= len;
            for (int r = 0; r < 3; ++r) { m[r * 4] = e1[r]; m[r * 4 + 1] = e2[r]; m[r * 4 + 2] = n[r]; m[r * 4 + 3] = p0[r]; }
            if (!invert3x4(m, inv)) return "degenerate triangle";
            g.type = PRIM_TRIANGLE;
            for (int k = 0; k < 3; ++k) { sh.origin[k] = (float) p0[k]; sh.eu[k] = (float) e1[k]; sh.ev[k] = (float) e2[k]; sh.n[k] = (float) n[k]; }
            sh.inv_len_eu = (float) (1.0 / std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]));
            sh.inv_area = (float) (1.0 / (0.5 * len));
            for (int k = 0; k < 3; ++k) {
                double a = p0[k], b = p0[k] + e1[k], c = p0[k] + e2[k];
                pb.lo[k] = (float) std::min(a, std::min(b, c)); pb.hi[k] = (float) std::max(a, std::max(b, c));
            }
        } else if (in.type == DRMLT_SHAPE_RECTANGLE) {
            for (int k = 0; k < 12; ++k) m[k] = in.data[k];
            if (!invert3x4(m, inv)) return "rectangle: singular toWorld";
            double eu[3] = {m[0], m[4], m[8]}, ev[3] = {m[1], m[5], m[9]};
            double lu = std::sqrt(eu[0] * eu[0] + eu[1] * eu[1] + eu[2] * eu[2]), lv = std::sqrt(ev[0] * ev[0] + ev[1] * ev[1] + ev[2] * ev[2]);
            double sdot = (eu[0] * ev[0] + eu[1] * ev[1] + eu[2] * ev[2]) / (lu * lv);
            if (std::fabs(sdot) > 1e-4) return "Error: 'toWorld' transformation contains shear!"; // rectangle.cpp:107-108
            // normal: objectToWorld(Normal(0,0,1)) = third row of the inverse, normalised
            double n[3] = {inv[8], inv[9], inv[10]};
            double ln = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            g.type = PRIM_RECTANGLE;
            // parametrise the rectangle on [0,1]^2 from its (-1,-1) corner: u' = (u + 1) / 2 (same test as a merged
            // triangle pair). Shading record: origin = that corner, eu / ev = the full edge vectors.
            for (int c = 0; c < 4; ++c) { inv[c] = 0.5 * inv[c] + (c == 3 ? 0.5 : 0.0); inv[4 + c] = 0.5 * inv[4 + c] + (c == 3 ? 0.5 : 0.0); }
            for (int k = 0; k < 3; ++k) {
                sh.origin[k] = (float) (m[k * 4 + 3] - eu[k] - ev[k]); sh.eu[k] = (float) (2.0 * eu[k]); sh.ev[k] = (float) (2.0 * ev[k]);
                sh.n[k] = (float) (n[k] / ln);
            }
            sh.inv_len_eu = (float) (1.0 / (2.0 * lu));
            sh.inv_area = (float) (1.0 / (4.0 * lu * lv)); // |dpdu| |dpdv| with dpdu = 2 eu
            for (int k = 0; k < 3; ++k) { qg.a[k] = m[k * 4 + 3] - eu[k] - ev[k]; qg.e1[k] = 2.0 * eu[k]; qg.e2[k] = 2.0 * ev[k]; }
            qg.usable = true; // the record's (u, v) run over [0, 1]^2 from that corner
            for (int k = 0; k < 3; ++k) {
                double c = m[k * 4 + 3], ext = std::fabs(eu[k]) + std::fabs(ev[k]);
                pb.lo[k] = (float) (c - ext); pb.hi[k] = (float) (c + ext);
            }
        } else if (in.type == DRMLT_SHAPE_SPHERE) {
            double r = in.data[3];
            if (!(r > 0)) return "sphere: radius must be positive";
            for (int k = 0; k < 12; ++k) inv[k] = 0;
            for (int k = 0; k < 3; ++k) { inv[k * 4 + k] = 1.0 / r; inv[k * 4 + 3] = -(double) in.data[k] / r; }
            g.type = PRIM_SPHERE;
            for (int k = 0; k < 3; ++k) { sh.origin[k] = in.data[k]; pb.lo[k] = (float) (in.data[k] - r); pb.hi[k] = (float) (in.data[k] + r); }
            sh.eu[0] = (float) r;
            sh.inv_area = (float) (1.0 / (4.0 * M_PI * r * r));
        } else {
            return "unknown shape type " + std::to_string(in.type);
        }
        for (int k = 0; k < 12; ++k) g.m[k] = (float) inv[k];
        sh.bsdf |= g.type << 24;
        g.shade = (int32_t) ctx->shade.size();
        ctx->prims.push_back(g);
        ctx->shade.push_back(sh);
        bounds.push_back(pb);
        geo.push_back(qg);
    }
    // ---- merge triangle pairs (a,b,c),(a,c,d) that form a parallelogram into one intersection record.
    // Exact: the hit is attributed to the sub-triangle it falls in, with that triangle's barycentrics and
    // shading record, so every path is the one two separate triangles would give -- at half the tests.
    if (!getenv("DRMLT_NO_QUAD_MERGE")) {
        std::vector<DPrim> merged;
        std::vector<PrimBounds> mb;
        std::vector<QuadGeo> mg;
        for (size_t i = 0; i < ctx->prims.size(); ++i) {
            bool did = false;
            if (i + 1 < ctx->prims.size() && s.shapes[i].type == DRMLT_SHAPE_TRIANGLE && s.shapes[i + 1].type == DRMLT_SHAPE_TRIANGLE &&
                s.shapes[i].bsdf == s.shapes[i + 1].bsdf && s.shapes[i].emitter < 0 && s.shapes[i + 1].emitter < 0) {
                const float *A = s.shapes[i].data, *B = s.shapes[i + 1].data; // A: a,b,c   B: a',c',d
                bool shared = true;
                for (int k = 0; k < 3; ++k) shared = shared && A[k] == B[k] && A[6 + k] == B[3 + k];
                double a[3], b[3], c[3], d[3], e1[3], e2[3], n[3], err = 0, scale = 0;
                for (int k = 0; k < 3; ++k) {
                    a[k] = A[k]; b[k] = A[3 + k]; c[k] = A[6 + k]; d[k] = B[6 + k];
                    err = std::max(err, std::fabs(d[k] - (a[k] + c[k] - b[k])));
                    scale = std::max(scale, std::max(std::fabs(c[k] - a[k]), std::fabs(b[k] - a[k])));
                    e1[k] = b[k] - a[k]; e2[k] = d[k] - a[k];
                }
                if (shared && err <= 1e-6 * scale) {
                    n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
                    double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                    double m[12], inv[12];
                    for (int r = 0; r < 3; ++r) { m[r * 4] = e1[r]; m[r * 4 + 1] = e2[r]; m[r * 4 + 2] = n[r] / len; m[r * 4 + 3] = a[r]; }
                    if (len > 0 && invert3x4(m, inv)) {
                        DPrim g{};
                        for (int k = 0; k < 12; ++k) g.m[k] = (float) inv[k];
                        g.type = PRIM_QUAD2;
                        g.shade = (int32_t) i; // records i (a,b,c) and i+1 (a,c,d)
                        PrimBounds pb = bounds[i];
                        for (int k = 0; k < 3; ++k) { pb.lo[k] = std::min(pb.lo[k], bounds[i + 1].lo[k]); pb.hi[k] = std::max(pb.hi[k], bounds[i + 1].hi[k]); }
                        QuadGeo qg{};
                        for (int k = 0; k < 3; ++k) { qg.a[k] = a[k]; qg.e1[k] = e1[k]; qg.e2[k] = e2[k]; }
                        qg.usable = true;
                        merged.push_back(g); mb.push_back(pb); mg.push_back(qg);
                        ++i;
                        did = true;
                    }
                }
            }
            if (!did) { merged.push_back(ctx->prims[i]); mb.push_back(bounds[i]); mg.push_back(geo[i]); }
        }
        ctx->prims.swap(merged);
        bounds.swap(mb);
        geo.swap(mg);
    }
    // emitters + DiscreteDistribution over sampling weights (scene.cpp m_emitterPDF, pmf.h:109-121)
    double total = 0;
    for (int i = 0; i < s.n_emitters; ++i) {
        const drmlt_emitter &e = s.emitters[i];
        if (e.type != DRMLT_EMITTER_AREA) return "unsupported emitter type";
        if (e.shape < 0 || e.shape >= s.n_shapes || s.shapes[e.shape].emitter != i) return "emitter/shape link mismatch";
        if (!(e.sampling_weight >= 0)) return "negative emitter sampling weight";
        total += e.sampling_weight;
    }
    if (!(total > 0)) return "emitter sampling weights sum to zero";
    float cdf = 0.f, norm = 1.0f / (float) total;
    std::vector<float> raw(s.n_emitters + 1, 0.f);
    for (int i = 0; i < s.n_emitters; ++i) { cdf += s.emitters[i].sampling_weight; raw[i + 1] = cdf; }
    for (int i = 1; i <= s.n_emitters; ++i) raw[i] *= norm;
    raw[s.n_emitters] = 1.f;
    for (int i = 0; i < s.n_emitters; ++i) {
        DEmitter e{};
        for (int k = 0; k < 3; ++k) e.radiance[k] = s.emitters[i].radiance[k];
        e.prim = s.emitters[i].shape;
        e.cdf_lo = raw[i]; e.cdf_hi = raw[i + 1];
        emitters.push_back(e);
    }
    return "";
}

// ReconstructionFilter::configure (rfilter.cpp:37-55) for box.cpp / gaussian.cpp
void build_filter(int type, float param, float lut[32], float &radius, float &scale) {
    const int res = 31;
    bool gauss = type == DRMLT_FILTER_GAUSSIAN;
    float stddev = param;
    radius = gauss ? 4.f * stddev : param + 1e-5f;
    float sum = 0.f;
    for (int i = 0; i < res; ++i) {
        float x = (radius * i) / res, v;
        if (!gauss) v = std::fabs(x) <= radius ? 1.f : 0.f;
        else {
            float alpha = -1.f / (2.f * stddev * stddev);
            v = std::max(0.f, std::exp(alpha * x * x) - std::exp(alpha * radius * radius));
        }
        lut[i] = v;
        sum += v;
    }
    lut[res] = 0.f;
    scale = res / radius;
    sum *= 2.f * radius / res;
    float normalization = 1.f / sum;
    for (int i = 0; i < res; ++i) lut[i] *= normalization;
}

// Overflow area of the traversal stacks (deep trees only): [ovf_entries][lanes of the launch] ints. Grown on demand before a
// launch whose grid has more lanes than any before it; `P` is the parameter block the launch will use.
static hipError_t ensure_overflow(drmlt_ctx *ctx, DParams &P, size_t lanes) {
    if (ctx->ovf_entries == 0) { P.bvh_overflow = nullptr; P.bvh_ovf_lanes = 0; return hipSuccess; }
    lanes = (lanes + 63) / 64 * 64;
    if (lanes > ctx->ovf_lanes) {
        hipError_t e = hipStreamSynchronize(ctx->stream); // a launch in flight may still be using the old area
        if (e != hipSuccess) return e;
        e = ctx->d_ovf.alloc((size_t) ctx->ovf_entries * lanes * sizeof(int32_t));
        if (e != hipSuccess) { ctx->ovf_lanes = 0; return e; }
        ctx->ovf_lanes = lanes;
    }
    ctx->P.bvh_overflow = P.bvh_overflow = ctx->d_ovf.as<int32_t>();
    ctx->P.bvh_ovf_lanes = P.bvh_ovf_lanes = (uint32_t) ctx->ovf_lanes;
    return hipSuccess;
}

int find_max_dim_path(int maxDepth, int rrDepth) { // pssmlt_utils.h:62-68 (no media, no rough dielectric)
    int maxDim = (maxDepth + 2) * (4 + (rrDepth < maxDepth ? 1 : 0));
    if (maxDim % 2 == 1) ++maxDim;
    return maxDim;
}
// dimensions MIPathTracer::Li can actually consume: 2 (film) + 4 per scattering event at depth
// 1..maxDepth-1 + one roulette draw per event at depth >= rrDepth; rounded up to a full pair
int effective_dim_path(int maxDepth, int rrDepth) {
    int events = maxDepth - 1;
    int rr = std::max(0, maxDepth - std::max(rrDepth, 1));
    int d = 2 + 4 * events + rr;
    if (d % 2 == 1) ++d;
    return d;
}

} // namespace

extern "C" {

uint32_t drmlt_abi_version(void) { return DRMLT_ABI_VERSION; }

const char *drmlt_last_error(drmlt_ctx *ctx) { return ctx ? ctx->error.c_str() : "null context"; }

drmlt_ctx *drmlt_create(const drmlt_config *cfg, const drmlt_scene *scene, int device, char *err, size_t errlen) {
    auto bail = [&](drmlt_ctx *c, const std::string &msg) -> drmlt_ctx * {
        if (err && errlen) snprintf(err, errlen, "%s", msg.c_str());
        delete c;
        return nullptr;
    };
    if (!cfg || !scene) return bail(nullptr, "null config or scene");
    if (cfg->struct_size != sizeof(drmlt_config) || scene->struct_size != sizeof(drmlt_scene))
        return bail(nullptr, "struct_size mismatch (ABI version skew)");
    // ---- parameter checks of the DRMLT ctor / PathSampler ctor (drmlt.cpp:193-349, pathsampler.cpp:57-71)
    if (cfg->algo != DRMLT_ALGO_DRMLT && cfg->algo != DRMLT_ALGO_PSSMLT) return bail(nullptr, "Unknown algorithm");
    if (cfg->algo == DRMLT_ALGO_PSSMLT && cfg->technique != DRMLT_TECH_PATH)
        return bail(nullptr, "algo=pssmlt runs over technique=path on the device (BASELINE config 1)");
    if (cfg->technique != DRMLT_TECH_PATH && cfg->technique != DRMLT_TECH_BDPT && cfg->technique != DRMLT_TECH_MMLT)
        return bail(nullptr, "Unknown technique type");
    if (cfg->type < DRMLT_TYPE_GREEN || cfg->type > DRMLT_TYPE_ORBITAL) return bail(nullptr, "Unknown implementation type");
    if (cfg->technique == DRMLT_TECH_MMLT && cfg->max_depth == -1) return bail(nullptr, "Impossible to use MMLT with no max depth");
    if (cfg->fix_emitter_path && cfg->technique != DRMLT_TECH_MMLT) return bail(nullptr, "Impossible to use fixEmitterPath without MMLT");
    if (cfg->scale_second > 1.0f) return bail(nullptr, "scaleSecond is bigger than the first stage");
    if (cfg->seed_rule != DRMLT_SEED_TARGET && cfg->seed_rule != DRMLT_SEED_REFERENCE) return bail(nullptr, "Unknown seeding rule (firstStageSeeding: target | reference)");
    if (cfg->work_units_rule != DRMLT_WORK_UNITS_DEVICE && cfg->work_units_rule != DRMLT_WORK_UNITS_REFERENCE) return bail(nullptr, "Unknown work-unit rule (workUnitsRule: device | reference)");
    const bool mmlt = cfg->technique == DRMLT_TECH_MMLT, bdpt = cfg->technique == DRMLT_TECH_BDPT;
    // (timidAfterLarge under bdpt: the reference's code path with its debug assertions compiled out -- a rejected large step's second
    // stage draws uniforms again for all three samplers, drmlt_sampler.cpp:319-321 -- as for technique=path, DESIGN deviation 2)
    // a wave's sampler and density rows (device_bdpt.h) take 60.7 KB of LDS at maxDepth 24 and pass the 64 KB of a workgroup at 26
    if (bdpt && cfg->max_depth > BDPT_MAX_DEPTH) return bail(nullptr, "technique=bdpt: maxDepth above 24 is not supported on the device");
    // ... and a vertex record packs its bsdf and emitter numbers into one word (device_bdpt.h: BR_IDS)
    if (bdpt && (scene->n_bsdfs > 4096 || scene->n_emitters > 65534)) return bail(nullptr, "technique=bdpt: more than 4096 bsdfs or 65534 emitters are not supported on the device");
    if (cfg->max_depth <= 0) return bail(nullptr, "technique=path needs a finite maxDepth (pssmlt_utils.h:63)");
    if (mmlt && cfg->max_depth > 24) return bail(nullptr, "technique=mmlt: maxDepth above 24 is not supported on the device");
    // a rejected large step re-draws the strategy; its second stage would read an emitter state that may be
    // empty (drmlt_sampler.cpp:189-191 with an unused emitter sampler): undefined in the reference, refused here
    if (mmlt && cfg->timid_after_large) return bail(nullptr, "timidAfterLarge is not defined for technique=mmlt");
    if (cfg->sample_count <= 0) return bail(nullptr, "sample_count must be positive");
    if (!(cfg->p_large >= 0.f && cfg->p_large <= 1.f)) return bail(nullptr, "pLarge must be in [0,1]");
    const drmlt_camera &cam = scene->camera;
    if (cam.width <= 0 || cam.height <= 0) return bail(nullptr, "film size must be positive");
    if (cam.filter != DRMLT_FILTER_BOX && cam.filter != DRMLT_FILTER_GAUSSIAN) return bail(nullptr, "unsupported reconstruction filter");
    if (cfg->acceptance_map && !(cam.filter == DRMLT_FILTER_BOX && cam.filter_param + 1e-5f - 0.500010f <= 1e-6f))
        return bail(nullptr, "Box filter required for acceptance map!"); // drmlt_proc.cpp:76-79

    drmlt_ctx *ctx = new drmlt_ctx();
    ctx->cfg = *cfg;
    ctx->device = device;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bail(ctx, "no HIP device available (the DRMLT kernels have no CPU fallback)");
    if (device < 0 || device >= ndev) return bail(ctx, "invalid device index");
    if (hipSetDevice(device) != hipSuccess) return bail(ctx, "hipSetDevice failed");
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return bail(ctx, "hipStreamCreate failed");
    ctx->own_stream = true;
    if (const char *e = getenv("DRMLT_SLICE")) ctx->slice = std::max(1, std::min(32768, atoi(e)));

    std::vector<DBsdf> bsdfs;
    std::vector<DEmitter> emitters;
    std::vector<PrimBounds> bounds;
    std::vector<QuadGeo> geo; // world-space parallelograms of the flat records (box_merge.h)
    std::string e = build_scene(ctx, *scene, bsdfs, emitters, bounds, geo);
    if (!e.empty()) return bail(ctx, e);

    // ---- acceleration structure: brute force over wave-uniform records for tiny scenes, BVH otherwise
    DParams &P = ctx->P;
    int bvh_threshold = 48;
    if (const char *t = getenv("DRMLT_BVH_THRESHOLD")) bvh_threshold = atoi(t);
    std::vector<DBvhNode> nodes;   // binary SAH tree (host only)
    std::vector<DBvh4Node> nodes4; // what the kernels traverse
    P.use_bvh = (int) ctx->prims.size() > bvh_threshold ? 1 : 0;
    if (P.use_bvh) {
        std::vector<int> order;
        // SAH splits wherever they lead (the builder's recursion bound, 64 levels, is far from what a surface-area tree
        // needs): a traversal stack that outgrows its LDS column spills to memory (device_path.h: trav_run)
        int max_depth = 64;
        if (const char *t = getenv("DRMLT_BVH_MAX_DEPTH")) max_depth = std::min(max_depth, atoi(t)); // tests: exercise the depth-bounded splits
        const int median_splits = build_bvh(bounds, nodes, order, max_depth);
        int leaf_shift = 0;
        const int depth4 = build_bvh4(nodes, nodes4, &leaf_shift);
        P.bvh_leaf_shift = leaf_shift;
        // the traversal addresses node and primitive records by 32-bit byte offsets into buffer resources of 2 GiB (trav_run)
        if (nodes4.size() * sizeof(DBvh4Node) >= (1ull << 31) || ctx->prims.size() * sizeof(DPrim) >= (1ull << 31))
            return bail(ctx, "scene too large: the BVH node and primitive arrays must stay below 2 GiB each");
        // 16-bit stack entries when every node index and leaf reference fits (k_mutate_v4: 3 KB of LDS instead of 6)
        P.bvh_stack16 = (nodes4.size() < 32768 && ((order.size() << leaf_shift) | 7u) < 32768 && !getenv("DRMLT_BVH_STACK32")) ? 1 : 0;
        // a 4-wide node pushes at most 3 entries, so a node at level l is entered with at most 3 (l - 1) on the stack and
        // 3 * depth4 bound it: up to BVH_STACK that is the LDS column (the branch-free pushes use its spare rows); deeper
        // trees get an overflow area in memory, sized per launch (ensure_overflow)
        ctx->bvh_depth = depth4;
        // (k_mutate_v4 keeps only 11 entries of a 32-bit stack in LDS: those scenes always have the area)
        ctx->ovf_entries = (3 * depth4 > BVH_STACK || !P.bvh_stack16) ? (3 * depth4 + 3 + BVH_SPILL - 1) / BVH_SPILL * BVH_SPILL : 0;
        if (getenv("DRMLT_VERBOSE")) fprintf(stderr, "[drmlt] BVH: %zu primitives, %zu binary / %zu 4-wide nodes, 4-wide depth %d (stack %d in LDS + %d in memory), %d median splits, %d-bit stack entries\n", order.size(), nodes.size(), nodes4.size(), depth4, BVH_STACK, ctx->ovf_entries, median_splits, P.bvh_stack16 ? 16 : 32);
        // intersection records go into leaf order; shading records stay where the emitters expect them
        std::vector<DPrim> np(order.size());
        for (size_t i = 0; i < order.size(); ++i) np[i] = ctx->prims[order[i]];
        ctx->prims.swap(np);
    }

    if (!P.use_bvh) { // brute-force order: flat records first, spheres last (trace(): flat loop, then the sphere loop)
        std::vector<size_t> perm(ctx->prims.size());
        for (size_t i = 0; i < perm.size(); ++i) perm[i] = i;
        std::stable_partition(perm.begin(), perm.end(), [&](size_t i) { return ctx->prims[i].type != PRIM_SPHERE; });
        std::vector<DPrim> np(perm.size());
        std::vector<QuadGeo> ng(perm.size());
        for (size_t i = 0; i < perm.size(); ++i) { np[i] = ctx->prims[perm[i]]; ng[i] = geo[perm[i]]; }
        ctx->prims.swap(np);
        geo.swap(ng);
    }
    for (DPrim &g : ctx->prims) g.kind_shade = g.type | (g.shade << 8);
    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> bool {
        if (b.alloc(std::max<size_t>(bytes, 64)) != hipSuccess) return false;
        return hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    float lut[32], radius, scale;
    build_filter(cam.filter, cam.filter_param, lut, radius, scale);
    bool ok = up(ctx->d_prims, ctx->prims.data(), ctx->prims.size() * sizeof(DPrim)) &&
              up(ctx->d_shade, ctx->shade.data(), ctx->shade.size() * sizeof(DShade)) &&
              up(ctx->d_bsdfs, bsdfs.data(), bsdfs.size() * sizeof(DBsdf)) &&
              up(ctx->d_emitters, emitters.data(), emitters.size() * sizeof(DEmitter)) &&
              up(ctx->d_lut, lut, sizeof lut);
    if (ok && P.use_bvh) ok = up(ctx->d_bvh, nodes4.data(), nodes4.size() * sizeof(DBvh4Node));
    // flat-primitive fast path of the brute-force loop: interleaved records + two sentinels no ray can hit
    // (ld.z = 0, lo.z = 1: t = -inf fails t >= tmin)
    P.prims_flat = nullptr; P.has_plain_tri = 0; P.n_flat = 0; P.n_flat_rec = 0; P.prims_box = nullptr; P.n_box = 0;
    const bool flat_loop = !P.use_bvh && !getenv("DRMLT_NO_FLAT_LOOP");
    for (const DPrim &g : ctx->prims) if (g.type != PRIM_SPHERE) P.n_flat++;
    if (ok && flat_loop) {
        // Faces that bound a parallelepiped -- a `cube`'s six merged triangle pairs, the walls of a room -- become ONE cuboid
        // record (box_merge.h; device_path.h: test_box): config 2's 18 records -> 1 + 3 cuboids. DRMLT_NO_BOX_MERGE: the
        // separate faces (the tests compare the two).
        std::vector<char> in_box((size_t) P.n_flat, 0);
        std::vector<DPrimBox> boxes;
        if (!getenv("DRMLT_NO_BOX_MERGE")) {
            std::vector<QuadGeo> fg(geo.begin(), geo.begin() + P.n_flat);
            for (size_t i = 0; i < fg.size(); ++i) // a record's shading index must fit the face half-word
                if (ctx->prims[i].shade >= 1024 || (ctx->prims[i].type != PRIM_RECTANGLE && ctx->prims[i].type != PRIM_QUAD2)) fg[i].usable = false;
            for (const BoxGeo &bg : find_boxes(fg)) {
                double m[12], inv[12];
                for (int r = 0; r < 3; ++r) { m[r * 4] = bg.E[0][r]; m[r * 4 + 1] = bg.E[1][r]; m[r * 4 + 2] = bg.E[2][r]; m[r * 4 + 3] = bg.a[r]; }
                if (!invert3x4(m, inv)) continue;
                DPrimBox b{};
                for (int c = 0; c < 4; ++c) { b.c[2 * c] = (float) inv[c]; b.c[2 * c + 1] = (float) inv[4 + c]; b.rz[c] = (float) inv[8 + c]; }
                for (int f = 0; f < 6; ++f) {
                    if (bg.face[f] < 0) continue;
                    const DPrim &g = ctx->prims[(size_t) bg.face[f]];
                    const uint32_t half = 1u | ((uint32_t) bg.code[f] << 1) | ((uint32_t) g.type << 4) | ((uint32_t) g.shade << 6);
                    b.fw[f >> 1] |= half << ((f & 1) ? 16 : 0);
                    in_box[(size_t) bg.face[f]] = 1;
                }
                boxes.push_back(b);
            }
            if (getenv("DRMLT_VERBOSE")) fprintf(stderr, "[drmlt] brute-force loop: %d flat records, %zu of them as the faces of %zu cuboids\n", P.n_flat,
                                                 (size_t) std::count(in_box.begin(), in_box.end(), 1), boxes.size());
        }
        std::vector<DPrimFlat> flat;
        for (size_t i = 0; i < (size_t) P.n_flat; ++i) {
            if (in_box[i]) continue;
            const DPrim &g = ctx->prims[i];
            if (g.type == PRIM_TRIANGLE) P.has_plain_tri = 1;
            DPrimFlat f{};
            for (int c = 0; c < 4; ++c) { f.c[2 * c] = g.m[c]; f.c[2 * c + 1] = g.m[4 + c]; f.rz[c] = g.m[8 + c]; }
            f.kind_shade = g.kind_shade;
            flat.push_back(f);
        }
        P.n_flat_rec = (int) flat.size();
        for (int k = 0; k < 2; ++k) { DPrimFlat f{}; f.rz[3] = 1.f; f.kind_shade = PRIM_RECTANGLE; flat.push_back(f); }
        ok = up(ctx->d_prims_flat, flat.data(), flat.size() * sizeof(DPrimFlat));
        if (ok) P.prims_flat = ctx->d_prims_flat.as<DPrimFlat>();
        if (ok && !boxes.empty()) {
            P.n_box = (int) boxes.size();
            boxes.push_back(DPrimBox{}); // sentinel for the read-ahead (never tested)
            ok = up(ctx->d_prims_box, boxes.data(), boxes.size() * sizeof(DPrimBox));
            if (ok) P.prims_box = ctx->d_prims_box.as<DPrimBox>();
        }
    } else {
        for (const DPrim &g : ctx->prims) if (g.type == PRIM_TRIANGLE) P.has_plain_tri = 1;
    }
    if (!ok) return bail(ctx, "device allocation/upload of the scene failed");

    // ---- derived quantities of DRMLT::render (drmlt.cpp:434-476)
    const uint64_t budget = (uint64_t) cam.width * cam.height * (uint64_t) cfg->sample_count;
    int work_units = cfg->work_units;
    if (work_units <= 0) {
        // "derived" (workUnits = -1, the default). The reference sizes work units for its CPU scheduler -- 200 000 (path) or
        // 100 000 (mmlt, bdpt) mutations each, drmlt.cpp:434-444: a few hundred chains for a whole image. A device wants the
        // count that fills it: 196 608 or 131 072 chains for the path technique's pool kernel (64 per wave, three or two waves per SIMD:
        // below), 131 072 for bdpt's one-chain-per-lane kernel, 262 144 or -- long renders -- 1 048 576 for mmlt's (two rounds, so that shallow waves make room for the next), but never chains shorter than 64 mutations. An explicit workUnits is
        // taken as given; drmlt_config.work_units_rule = DRMLT_WORK_UNITS_REFERENCE (adaptor: workUnitsRule=reference) restores the reference's formula.
        if (cfg->work_units_rule == DRMLT_WORK_UNITS_REFERENCE) {
            const uint64_t per_unit = (mmlt || bdpt) ? 100000 : 200000;
            work_units = (int) std::max<uint64_t>(1, (budget + per_unit - 1) / per_unit);
        } else {
            // (k_mutate_v5, the ray-pool kernel, carries 64 chains per wave: 131 072 fill the device; it is the path technique's kernel
            // for all three types -- flat scenes included: 2.15e9 at 131 072 chains against k_mutate_v4's 1.79e9 at 65 536)
            const bool pool_kernel = !mmlt && !bdpt && cfg->algo != DRMLT_ALGO_PSSMLT && !getenv("DRMLT_KERNEL");
            // (with its proposal rows in device memory the pool kernel runs a THIRD wave per SIMD, kernels.hip: ROWS_MEM -- 196 608 chains:
            // traversed scenes, whose node fetches the extra wave covers, + 4 % (2000 triangles) ... + 17 % (50 000, 1 000 000), flat
            // scenes + 20 % -- measured with a step's components read together)
            const bool small_tables = ctx->shade.size() * 64 + bsdfs.size() * 48 + emitters.size() * 32 <= 16384; // (= P.tables_in_lds, below)
            const bool three_waves = pool_kernel && (P.use_bvh || small_tables);
            // mmlt: MANY rounds of waves, run in depth order -- the kernel holds two 64-chain waves per SIMD (131 072 chains), and the more
            // waves queue behind them the less of a launch is its tail: 262 144 chains 2.56e9 mutations/s on BASELINE's config 5, 524 288
            // 2.79e9, 1 048 576 2.91e9, 2 097 152 2.94e9. (bdpt, whose workspace is 2 KB per chain, loses with more than fill the device.)
            // The price is paid before the first mutation: 50 x maxDepth bootstrap samples per chain (drmlt.cpp:456-473) are 3e8 samples for a
            // million chains at maxDepth 6 -- 2.3 s of seeding against 0.6 s for 262 144 chains (the resampling table is built on the host).
            // A render gets the million chains when the 14 % are worth more than that: from 2^35 mutations (12 s of kernel time) up.
            const uint64_t fill = mmlt ? (budget >= (1ull << 35) ? 1048576 : 262144) : (three_waves ? 196608 : ((bdpt || pool_kernel) ? 131072 : 65536));
            work_units = (int) std::min<uint64_t>(fill, std::max<uint64_t>(64, budget / 64 / 64 * 64));
        }
    }
    ctx->cfg.work_units = work_units;
    ctx->n_chains = (uint32_t) work_units;

    P.prims = ctx->d_prims.as<DPrim>(); P.shade = ctx->d_shade.as<DShade>(); P.bsdfs = ctx->d_bsdfs.as<DBsdf>();
    P.emitters = ctx->d_emitters.as<DEmitter>(); P.bvh = ctx->d_bvh.as<DBvh4Node>(); P.filter_lut = ctx->d_lut.as<float>();
    P.n_prims = (int) ctx->prims.size(); P.n_shade = (int) ctx->shade.size(); P.n_emitters = (int) emitters.size(); P.n_bvh_nodes = (int) nodes4.size();
    P.n_bsdfs = (int) bsdfs.size();
    // tables ride in LDS when they are small (Cornell class); 16 KB cap keeps 4+ waves per CU
    P.tables_in_lds = (ctx->shade.size() * 64 + bsdfs.size() * 48 + emitters.size() * 32 <= 16384) ? 1 : 0;
    if (const char *t = getenv("DRMLT_TABLES_LDS")) P.tables_in_lds = atoi(t) ? P.tables_in_lds : 0;
    P.box_weight = cam.filter == DRMLT_FILTER_BOX ? lut[0] : 0.f;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) P.cam[r * 4 + c] = cam.to_world[r * 4 + c];
    P.tan_half_fov = (float) std::tan(0.5 * (double) cam.fov_x_deg * M_PI / 180.0);
    P.inv_aspect = (float) cam.height / (float) cam.width;
    P.near_clip = cam.near_clip; P.far_clip = cam.far_clip;
    P.width = cam.width; P.height = cam.height;
    P.filter_radius = radius; P.filter_scale = scale;
    P.type = cfg->type; P.max_depth = cfg->max_depth; P.rr_depth = cfg->rr_depth;
    P.exclude_direct = cfg->direct_samples >= 0 ? 1 : 0; // separateDirect, drmlt.cpp:242
    P.acceptance_map = cfg->acceptance_map; P.timid_after_large = cfg->timid_after_large; P.use_mixture = cfg->use_mixture;
    P.max_dim = find_max_dim_path(cfg->max_depth, cfg->rr_depth);
    P.eff_dim = std::min(P.max_dim, effective_dim_path(cfg->max_depth, cfg->rr_depth));
    P.p_large = cfg->p_large; P.sigma2 = cfg->scale_second * cfg->sigma;
    P.n_chains = ctx->n_chains;
    P.kelemen_weights = cfg->kelemen_style_weights; P.kelemen_mutation = cfg->kelemen_style_mutation;
    P.pss_sigma = cfg->sigma; P.luminance_b = 1.f;
    P.technique = cfg->technique; P.light_image = cfg->no_light_image ? 0 : 1; P.fix_emitter_path = cfg->fix_emitter_path;
    P.mmlt_S = P.mmlt_E = P.mmlt_dmax = P.bd_Dd = 0;
    if (mmlt) { // PSS layout of a chain: [sensor S | emitter E | direct] (device_bidir.h)
        P.mmlt_S = 2 * (cfg->max_depth + 1); P.mmlt_E = 2 * cfg->max_depth;
        P.mmlt_dmax = (cfg->max_depth + 2) * 3; P.mmlt_dmax += P.mmlt_dmax & 1; // pssmlt_utils.h:58-63
        P.max_dim = 2 * P.mmlt_dmax + 1;
        P.eff_dim = P.mmlt_S + P.mmlt_E + 1;
    }
    if (bdpt) { // [sensor S | emitter E | direct Dd]: what the two walks and the direct strategies can consume (device_bdpt.h)
        const int rr = cfg->max_depth + 1 - (cfg->rr_depth > 0 ? cfg->rr_depth : 0);
        P.mmlt_S = 2 * (cfg->max_depth + 1) + (rr > 0 ? rr : 0); P.mmlt_S += P.mmlt_S & 1;
        P.mmlt_E = 2 * cfg->max_depth + (rr > 1 ? rr - 1 : 0); P.mmlt_E += P.mmlt_E & 1;
        P.mmlt_dmax = (cfg->max_depth + 2) * (2 + (cfg->rr_depth < cfg->max_depth ? 1 : 0)); P.mmlt_dmax += P.mmlt_dmax & 1; // pssmlt_utils.h:69-75
        // directSampling=true (the reference's default): every s = 1 / t = 1 connection draws two components of the direct
        // sampler (pathsampler.cpp:424-452, vertex.cpp:1304-1305), a sample makes up to maxDepth + (maxDepth - 1) of them. The
        // reference sizes that sampler maxDepth (pssmlt_utils.h:75) and reads past it; here it holds what can be consumed.
        P.bd_Dd = cfg->no_direct_sampling ? 0 : 2 * (2 * cfg->max_depth - 1); // bdpt_dims_direct, device_bdpt.h
        P.max_dim = 2 * P.mmlt_dmax + P.bd_Dd;
        P.eff_dim = P.mmlt_S + P.mmlt_E + P.bd_Dd;
    }

    ctx->film_floats = (size_t) cam.width * cam.height * 3;
    const size_t film_bytes = film_alloc_floats(cam.width, cam.height) * sizeof(float); // zero rows behind the film: film_tiles.h
    ok = ctx->d_film.alloc(film_bytes) == hipSuccess && ctx->d_x.alloc((size_t) P.eff_dim * ctx->n_chains * sizeof(float)) == hipSuccess &&
         ctx->d_cur.alloc((size_t) 6 * ctx->n_chains * sizeof(float)) == hipSuccess && ctx->d_stats.alloc(32 * sizeof(unsigned long long)) == hipSuccess &&
         ctx->d_err.alloc(64) == hipSuccess && ctx->d_chain_i.alloc((size_t) 2 * ctx->n_chains * sizeof(int32_t)) == hipSuccess;
    if (!ok) return bail(ctx, "device allocation of chain state / film failed");
    if (hipMemset(ctx->d_chain_i.p, 0, (size_t) 2 * ctx->n_chains * sizeof(int32_t)) != hipSuccess) return bail(ctx, "hipMemset of the chain state failed");
    P.chain_depth = ctx->d_chain_i.as<int32_t>(); P.cur_t = P.chain_depth + ctx->n_chains;
    // run-ahead of k_mutate_v4 (drmlt_run): per-chain mutation counts + the launch's counter of waves short of the target
    if (ctx->d_done.alloc(((size_t) ctx->n_chains + 1) * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(ctx->d_done.p, 0, ((size_t) ctx->n_chains + 1) * sizeof(uint32_t)) != hipSuccess) return bail(ctx, "device allocation failed");
    P.chain_done = nullptr; P.waves_left = ctx->d_done.as<uint32_t>() + ctx->n_chains; P.run_limit = 0;
    P.importance = nullptr;
    P.bd_verts = nullptr; P.bd_lists = nullptr; P.n_chains_alloc = ctx->n_chains;
    if (bdpt) {
        const size_t nvs = (size_t) 2 * cfg->max_depth + 1, rows = (size_t) 7 + 5 * cfg->max_depth;
        if (ctx->d_bd_verts.alloc(((size_t) (20 + 2) * nvs + 2 * ((size_t) cfg->max_depth + 2)) * ctx->n_chains * sizeof(float)) /* records of BR_FLOATS, the fp64 MIS tails, the s = 1 emitter samples: device_bdpt.h */ != hipSuccess ||
            ctx->d_bd_lists.alloc((size_t) 3 * rows * ctx->n_chains * sizeof(float)) != hipSuccess)
            return bail(ctx, "device allocation of the bdpt workspace failed");
        if (hipMemset(ctx->d_bd_lists.p, 0, (size_t) 3 * rows * ctx->n_chains * sizeof(float)) != hipSuccess) return bail(ctx, "hipMemset of the bdpt workspace failed");
        P.bd_verts = ctx->d_bd_verts.as<float>(); P.bd_lists = ctx->d_bd_lists.as<float>();
    }
    if (hipMemset(ctx->d_film.p, 0, film_bytes) != hipSuccess || hipMemset(ctx->d_stats.p, 0, 32 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(ctx->d_err.p, 0, 64) != hipSuccess)
        return bail(ctx, "hipMemset of the film / counters failed");
    P.film = ctx->d_film.as<float>();
    P.x = ctx->d_x.as<float>();
    float *cur = ctx->d_cur.as<float>();
    P.cur_lum = cur; P.cur_px = cur + ctx->n_chains; P.cur_py = cur + 2 * (size_t) ctx->n_chains;
    P.cur_r = cur + 3 * (size_t) ctx->n_chains; P.cur_g = cur + 4 * (size_t) ctx->n_chains; P.cur_b = cur + 5 * (size_t) ctx->n_chains;
    P.stats = ctx->d_stats.as<unsigned long long>();
    P.error_flag = ctx->d_err.as<int32_t>();
    P.debug = 0;
    if (const char *d = getenv("DRMLT_DEBUG")) P.debug = atoi(d);
    // chain kernel of technique=path: 4 = k_mutate_v4 (lane pairs, 32 chains per wave), 5 = k_mutate_v5 (ray pool, 64 chains per wave:
    // the default where it applies, see below), 3 = k_mutate_v3 (the bit-equality cross-check)
    P.kernel_variant = 5;
    if (const char *k = getenv("DRMLT_KERNEL")) { int kv = atoi(k); P.kernel_variant = kv == 3 ? 3 : (kv == 5 ? 5 : 4); }
    P.features = 0;
    for (const DBsdf &b : bsdfs) P.features |= b.type == DRMLT_BSDF_ROUGHCONDUCTOR ? 1 : (b.type == DRMLT_BSDF_DIELECTRIC ? 2 : 0);
    for (const DPrim &g : ctx->prims) if (g.type == PRIM_SPHERE) P.features |= 4;
    if (P.use_bvh) P.features |= 8;
    if (getenv("DRMLT_FEAT_ALL")) P.features = 15;
    // the ray-pool kernel keeps ONE proposal row group in LDS: Green's reverse move and Mira's ratio, which need x, y and z together,
    // recompute what is not there from the state in device memory and the addressed stream.
    // On flat scenes it needs the chains to put two of its 64-chain waves on a SIMD: from 98 304 chains up it is the default
    // (Cornell: v5 2.15e9 at 131 072 chains, 1.11e9 at 65 536; v4 1.79e9 at 65 536, 1.55e9 at 131 072). BASELINE's config 2 fixes
    // 65 536 chains and therefore runs k_mutate_v4.
    const bool v5_forced = getenv("DRMLT_KERNEL") && atoi(getenv("DRMLT_KERNEL")) == 5;
    if (P.kernel_variant == 5 && !P.use_bvh && !v5_forced && ctx->n_chains < 98304u) P.kernel_variant = 4;
    // (v5 on the Cornell scene, 131 072 chains: batch 16 1.98e9, 24 2.08e9, 32 2.13e9, 48 1.84e9; on the soup: 8 5.05e8, 16 5.24e8, 32 5.06e8)
    // A scene whose nodes and records exceed the L2 caches (8 x 4 MB) is traversed against memory latency: the ray pool then wants
    // SHORT phases -- chains step and refill it as soon as a few rays are done (1 000 000 triangles, 2-step calls: yield x batch
    // 20 x 16 6.9e7, 12 x 8 7.5e7, 8 x 8 7.7e7, 4 x 8 7.7e7 mutations/s; 50 000 triangles, in the L2s: 2.82e8 / 2.75e8 / 2.61e8)
    const bool beyond_l2 = P.use_bvh && (size_t) P.n_bvh_nodes * sizeof(DBvh4Node) + ctx->prims.size() * sizeof(DPrim) > (size_t) 32 << 20;
    // (round 4, with the cuboid records -- a cheaper trace pass moves the balance towards larger batches: v4 on config 2 batch 8 1.99e9, 12 2.02e9,
    // 14 2.05e9, 16 2.04e9, 20 1.94e9; v5 on the same scene at 131 072 chains 24 2.33e9, 32 2.39e9, 40 2.42e9, 48 2.36e9)
    P.mh_batch = P.kernel_variant == 5 ? (P.use_bvh ? (beyond_l2 ? 8 : 16) : 40) : (P.kernel_variant == 4 ? (P.use_bvh ? (P.bvh_stack16 ? 6 : 4) : (P.features == 0 ? 14 : 8)) : 32); // v4 / v5: chains run free, the bookkeeping branch fires as soon as a few are parked
    if (const char *k = getenv("DRMLT_MH_BATCH")) P.mh_batch = std::max(1, std::min(64, atoi(k)));
    // k_mutate_v5 on traversed scenes with chains for more than two 64-chain waves per SIMD (from 163 840 per GPU): the proposal
    // rows move from LDS to device memory and the kernel is built for three waves per SIMD (kernels.hip: ROWS_MEM)
    P.rows = nullptr;
    {
        int cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        const bool can = P.kernel_variant == 5 && (P.use_bvh || P.tables_in_lds) && !mmlt && !bdpt && cfg->algo != DRMLT_ALGO_PSSMLT;
        bool rows_mem = can && (uint64_t) ctx->n_chains * 2u >= (uint64_t) cus * 4u * 64u * 5u;
        if (const char *e = getenv("DRMLT_ROWS_MEM")) rows_mem = atoi(e) != 0 && can;
        if (rows_mem) {
            if (ctx->d_rows.alloc((size_t) P.eff_dim * ctx->n_chains * sizeof(float)) != hipSuccess) return bail(ctx, "device allocation of the proposal rows failed");
            P.rows = ctx->d_rows.as<float>();
        }
    }
    // k_mutate_v5 on traversed scenes: room for the small tables beside the pool? LDS per wave without them: 20 480 B in the builds with
    // 32-bit stacks and rows in LDS (none), 19.5 KB with 16-bit stacks, 11.5 / 10.8 KB with the rows in device memory (twelve waves per CU: 13 KB)
    {
        const size_t small = ((size_t) P.n_bsdfs * 12 + (size_t) P.n_emitters * 24) * sizeof(float);
        const size_t room = P.rows ? (P.bvh_stack16 ? 1536 : 1024) : (P.bvh_stack16 ? 768 : 0);
        P.small_tables_lds = (P.use_bvh && P.kernel_variant == 5 && small <= room && !getenv("DRMLT_NO_SMALL_TABLES")) ? 1 : 0;
        P.pad_tables = 0;
    }
    P.bvh_overflow = nullptr; P.bvh_ovf_lanes = 0;
    P.exec_order = nullptr;
    // measured (5-launch calls) on the 2000-triangle soup: 16 3.69e8, 20 3.80e8, 24 3.84e8, 28 3.84e8 mutations/s; on 50 000 triangles (32-bit
    // stacks, longer traversals): 20 1.90e8, 24 1.86e8, 28 1.79e8; bookkeeping batch there 4 1.89e8, 6 1.86e8, 8 1.82e8
    // k_mutate_v5 (131 072 chains, 3-step calls) on the soup: yield x bookkeeping batch -- 12: 4.84 / 5.13 / 5.12e8 (batch 8 / 16 / 28),
    // 16: 5.08 / 5.37 / 5.20, 20: 5.25 / 5.44 / 5.10, 24: 5.31 / 5.40 / 4.81
    P.trace_yield = P.kernel_variant == 5 ? (beyond_l2 ? 8 : 20) : (P.bvh_stack16 ? 24 : 20);
    if (const char *k = getenv("DRMLT_TRACE_YIELD")) P.trace_yield = std::max(0, std::min(64, atoi(k)));
    P.boot_weighted = 0;
    P.pool_refill = 8; // soup, 131 072 chains: 1 5.38e8, 2 5.41e8, 4 5.43e8, 8 5.45e8, 16 5.37e8 mutations/s
    if (const char *k = getenv("DRMLT_POOL_REFILL")) P.pool_refill = std::max(1, std::min(64, atoi(k)));
    P.trace_vote = 10; // measured on the 2000-triangle soup: 16 (plain majority) 2.70e8, 10 2.78e8, 5 2.73e8 mutations/s
    if (const char *k = getenv("DRMLT_TRACE_VOTE")) P.trace_vote = std::max(1, std::min(1024, atoi(k)));
    if (hipDeviceSynchronize() != hipSuccess) return bail(ctx, "device synchronisation failed after setup");
    return ctx;
}

void drmlt_destroy(drmlt_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    delete ctx;
}

int drmlt_set_stream(drmlt_ctx *ctx, void *hip_stream) {
    if (!ctx) return DRMLT_E_INVALID;
    (void) hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream && ctx->stream) (void) hipStreamDestroy(ctx->stream);
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
    return DRMLT_OK;
}

// Bootstrap + seed selection + replay. `pool_chains` = 0: this context draws its own seeds from its own bootstrap stream
// (stream id = chain_offset; per-rank estimates of b are averaged by the caller, the reference's multi-threaded seeding,
// drmlt.cpp:531-546). `pool_chains` > 0: SURVEY 8(e)'s global pool -- the bootstrap stream 0 is sized for, and
// `pool_chains` seeds are drawn for, the whole job (every rank repeats this cheap step and gets the same list and the
// same b); this context then takes seeds [chain_offset, chain_offset + work_units) of the sorted list. A job split over
// several contexts runs exactly the chains one context with pool_chains work units would.
static int seed_impl(drmlt_ctx *ctx, uint64_t seed, uint32_t chain_offset, uint32_t pool_chains, double *b_out) {
    if (!ctx) return DRMLT_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DParams &P = ctx->P;
    const bool pool = pool_chains > 0;
    if (pool && (uint64_t) chain_offset + ctx->n_chains > pool_chains) return ctx->fail(DRMLT_E_INVALID, "seed pool of %u chains does not cover chains [%u, %u)", pool_chains, chain_offset, chain_offset + ctx->n_chains);
    const uint32_t n_select = pool ? pool_chains : ctx->n_chains;
    EventPair ev;
    HIP_TRY(ctx, ev.create());
    HIP_TRY(ctx, hipEventRecord(ev.a, ctx->stream));
    P.key0 = (uint32_t) seed; P.key1 = (uint32_t) (seed >> 32);
    P.chain_offset = chain_offset; P.boot_stream = pool ? 0u : chain_offset;
    ctx->chain_offset = chain_offset;
    // luminance sample floor: max(luminanceSamples, 10 * workUnits), drmlt.cpp:454-466
    // technique=mmlt: x50, and as many again per depth; b is scaled by maxDepth below (drmlt.cpp:456-473,
    // pathsampler.cpp:884-890,932-934). One bootstrap stream per GPU, as with nCores = 1 in the reference.
    const bool mmlt = ctx->cfg.technique == DRMLT_TECH_MMLT;
    uint64_t n64 = (uint64_t) std::max<int64_t>(ctx->cfg.luminance_samples, (int64_t) n_select * (mmlt ? 50 : 10));
    if (mmlt) n64 *= (uint64_t) ctx->cfg.max_depth;
    if (n64 > 0x7fffffffull) return ctx->fail(DRMLT_E_INVALID, "too many luminance samples");
    uint32_t n = (uint32_t) n64;
    // Two-stage MLT: the chains sample f / importance, so that is what their seeds are drawn from (each bootstrap sample's luminance under
    // the map, second half of the buffer). The reference draws them from f itself (pathsampler.cpp:903-905 takes the luminance BEFORE
    // SplatList::normalize(importanceMap)) -- chains then start outside their stationary distribution; over its work units of 1e5
    // mutations that start-up bias is nothing, over the device's short chains it is not (a map of contrast 100 on the Cornell box, 1024
    // mutations per chain: the bright half + 13 %, the dark half - 15 %; DESIGN section 5, deviation 19). b stays the mean of f.
    // drmlt_config.seed_rule = DRMLT_SEED_REFERENCE (adaptor: firstStageSeeding=reference) restores the reference's rule; the oracle
    // follows the same field, so the chain-tracking tests run under both.
    const bool weighted_seeds = P.importance != nullptr && ctx->cfg.seed_rule == DRMLT_SEED_TARGET;
    P.boot_weighted = weighted_seeds ? 1 : 0;
    DevBuf d_lum;
    HIP_TRY(ctx, d_lum.alloc((size_t) n * (weighted_seeds ? 2 : 1) * sizeof(float)));
    const bool bdpt = ctx->cfg.technique == DRMLT_TECH_BDPT;
    HIP_TRY(ctx, ensure_overflow(ctx, P, std::max<size_t>(n, 2 * (size_t) P.n_chains_alloc)));
    if (mmlt) launch_bootstrap_mmlt(P, n, d_lum.as<float>(), ctx->stream);
    else if (bdpt) launch_bootstrap_bdpt(P, n, d_lum.as<float>(), ctx->stream);
    else launch_bootstrap(P, n, d_lum.as<float>(), ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    std::vector<float> lum((size_t) n * (weighted_seeds ? 2 : 1));
    HIP_TRY(ctx, hipMemcpyAsync(lum.data(), d_lum.p, lum.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    P.boot_weighted = 0;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

    // generateSeeds, pathsampler.cpp:879-954: mean over non-NaN samples, CDF over the non-zero ones
    double sum = 0.0, tok = 0.0;
    std::vector<uint32_t> idx;
    std::vector<double> cdf;
    idx.reserve(n);
    cdf.reserve((size_t) n + 1); // (hundreds of millions of samples with technique=mmlt's derived chain count: no reallocation on the way)
    cdf.push_back(0.0);
    for (uint32_t i = 0; i < n; ++i) {
        float l = lum[i];
        if (std::isnan(l)) continue;
        tok += 1.0;
        sum += (double) l;
        const float lw = weighted_seeds ? lum[(size_t) n + i] : l; // what the seed is drawn in proportion to
        if (l != 0.f && lw > 0.f && std::isfinite(lw)) { idx.push_back(i); cdf.push_back(cdf.back() + (double) lw); }
    }
    double mean = tok > 0 ? sum / tok : 0.0;
    if (mmlt) mean *= (double) ctx->cfg.max_depth; // "As we split the path by corresponding depth"
    if (!(mean > 0.0))
        return ctx->fail(DRMLT_E_ZERO_LUM, "The average image luminance appears to be zero! This could indicate a problem with the scene setup.");
    if (idx.empty()) // mean(f) > 0, yet no sample can seed a chain: every contribution lies where the importance map is zero (or not finite)
        return ctx->fail(DRMLT_E_ZERO_LUM, "No bootstrap sample has a finite positive luminance under the importance map (average luminance %g): "
                                           "the map is zero wherever the scene contributes; firstStageSeeding=reference seeds from the plain luminance", mean);
    const double norm = 1.0 / cdf.back();
    for (size_t i = 1; i < cdf.size(); ++i) cdf[i] *= norm;
    cdf.back() = 1.0;
    // One pick per chain: DiscreteDistribution::sample on the uniform of its own stream address. The picks are SORTED afterwards
    // (PathSeedSortPredicate), so the order in which they are made is free: the uniforms are sorted first and the table is walked once,
    // each lower_bound galloping on from the previous one -- the same entry as a search of the whole table, without a million cold binary
    // searches through gigabytes (technique=mmlt derives a million chains and 50 x maxDepth bootstrap samples for each).
    std::vector<double> xis(n_select);
    for (uint32_t j = 0; j < n_select; ++j) {
        uint32_t r[4];
        philox_host(P.key0, P.key1, 0u, j, P.boot_stream, TAG_SEEDSEL, r);
        xis[j] = (double) ((float) (r[0] >> 8) * (1.0f / 16777216.0f));
    }
    std::sort(xis.begin(), xis.end());
    std::vector<uint32_t> seed_index(n_select);
    size_t from = 0; // lower_bound of the previous (smaller or equal) uniform: the next one's is not before it
    for (uint32_t j = 0; j < n_select; ++j) {
        const double xi = xis[j];
        size_t step = 1, hi = from;
        while (hi < cdf.size() && cdf[hi] < xi) { from = hi + 1; hi += step; step *= 2; }
        auto entry = std::lower_bound(cdf.begin() + from, cdf.begin() + std::min(hi, cdf.size()), xi);
        from = (size_t) (entry - cdf.begin());
        size_t index = (size_t) std::max<ptrdiff_t>(0, (entry - cdf.begin()) - 1);
        index = std::min(cdf.size() - 2, index);
        while (cdf[index + 1] - cdf[index] == 0 && index < cdf.size() - 1) ++index;
        seed_index[j] = idx[index];
    }
    std::sort(seed_index.begin(), seed_index.end()); // PathSeedSortPredicate
    if (pool) { // this context's slice of the job's seed list
        std::vector<uint32_t> mine(seed_index.begin() + chain_offset, seed_index.begin() + chain_offset + ctx->n_chains);
        seed_index.swap(mine);
    }
    ctx->seed_indices = seed_index;
    std::vector<float> seed_lum(ctx->n_chains);
    for (uint32_t j = 0; j < ctx->n_chains; ++j) seed_lum[j] = lum[seed_index[j]];

    DevBuf d_si, d_sl;
    HIP_TRY(ctx, d_si.alloc(seed_index.size() * sizeof(uint32_t)));
    HIP_TRY(ctx, d_sl.alloc(seed_lum.size() * sizeof(float)));
    HIP_TRY(ctx, hipMemcpyAsync(d_si.p, seed_index.data(), seed_index.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_sl.p, seed_lum.data(), seed_lum.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_err.p, 0, 64, ctx->stream));
    HIP_TRY(ctx, ensure_overflow(ctx, P, 2 * (size_t) P.n_chains_alloc));
    if (mmlt && !getenv("DRMLT_MMLT_NO_SORT")) {
        // execution order of k_mutate_mmlt: chains sorted by their (fixed) path depth, deepest first, whole waves (kernels_mmlt.hip)
        const uint32_t n = ctx->n_chains, padded = (n + 63u) / 64u * 64u;
        std::vector<uint32_t> order(padded, n);
        for (uint32_t j = 0; j < n; ++j) order[j] = j;
        const uint32_t md = (uint32_t) ctx->cfg.max_depth;
        std::stable_sort(order.begin(), order.begin() + n, [&](uint32_t a, uint32_t b) { return seed_index[a] % md > seed_index[b] % md; });
        HIP_TRY(ctx, ctx->d_order.alloc(order.size() * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_order.p, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // `order` is a local
        ctx->P.exec_order = P.exec_order = ctx->d_order.as<uint32_t>();
    }
    if (bdpt && !getenv("DRMLT_NO_REGROUP")) { // execution order of k_mutate_bdpt: identity until the first launch has told the chains' work apart (regroup_chains)
        const uint32_t n = ctx->n_chains, padded = (n + 63u) / 64u * 64u;
        std::vector<uint32_t> order(padded, n);
        for (uint32_t j = 0; j < n; ++j) order[j] = j;
        HIP_TRY(ctx, ctx->d_order.alloc(order.size() * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_order.p, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // `order` is a local
        ctx->P.exec_order = P.exec_order = ctx->d_order.as<uint32_t>();
    }
    if (mmlt) launch_init_chains_mmlt(P, d_si.as<uint32_t>(), d_sl.as<float>(), ctx->stream);
    else if (bdpt) launch_init_chains_bdpt(P, d_si.as<uint32_t>(), d_sl.as<float>(), ctx->stream);
    else launch_init_chains(P, d_si.as<uint32_t>(), d_sl.as<float>(), ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    int32_t flag = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&flag, ctx->d_err.p, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ev.b, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->seed_ms += ev.elapsed_ms();
    if (flag) return ctx->fail(DRMLT_E_REPLAY, "Error when reconstructing a seed path: luminance mismatch");

    ctx->b = mean;
    if (ctx->cfg.acceptance_map) ctx->b = 1.0;                                       // drmlt.cpp:550-552
    else if (ctx->cfg.average_luminance != -1.0f) ctx->b = ctx->cfg.average_luminance; // drmlt.cpp:555-558
    ctx->seeded = true;
    ctx->regrouped = false;
    ctx->mutation_base = 0;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_done.p, 0, ctx->d_done.bytes, ctx->stream));
    if (b_out) *b_out = ctx->b;
    return DRMLT_OK;
}

int drmlt_seed(drmlt_ctx *ctx, uint64_t seed, uint32_t chain_offset, double *b_out) { return seed_impl(ctx, seed, chain_offset, 0u, b_out); }

int drmlt_bootstrap_luminances(drmlt_ctx *ctx, uint64_t seed, uint32_t stream, uint32_t n, float *out) {
    if (!ctx || !out) return DRMLT_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DParams P = ctx->P;
    P.key0 = (uint32_t) seed; P.key1 = (uint32_t) (seed >> 32);
    P.boot_stream = stream;
    P.boot_weighted = 0;
    DevBuf d_lum;
    HIP_TRY(ctx, d_lum.alloc((size_t) n * sizeof(float)));
    HIP_TRY(ctx, ensure_overflow(ctx, P, std::max<size_t>(n, 2 * (size_t) P.n_chains_alloc)));
    if (ctx->cfg.technique == DRMLT_TECH_MMLT) launch_bootstrap_mmlt(P, n, d_lum.as<float>(), ctx->stream);
    else if (ctx->cfg.technique == DRMLT_TECH_BDPT) launch_bootstrap_bdpt(P, n, d_lum.as<float>(), ctx->stream);
    else launch_bootstrap(P, n, d_lum.as<float>(), ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, d_lum.p, (size_t) n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DRMLT_OK;
}

int drmlt_seed_indices(drmlt_ctx *ctx, uint32_t *out) {
    if (!ctx || !out) return DRMLT_E_INVALID;
    if (!ctx->seeded) return ctx->fail(DRMLT_E_STATE, "drmlt_seed_indices before drmlt_seed");
    memcpy(out, ctx->seed_indices.data(), ctx->seed_indices.size() * sizeof(uint32_t));
    return DRMLT_OK;
}

int drmlt_seed_pool(drmlt_ctx *ctx, uint64_t seed, uint32_t first_chain, uint32_t pool_chains, double *b_out) {
    if (!ctx) return DRMLT_E_INVALID;
    if (pool_chains == 0) return ctx->fail(DRMLT_E_INVALID, "drmlt_seed_pool: pool_chains must be positive");
    return seed_impl(ctx, seed, first_chain, pool_chains, b_out);
}

// Two-stage MLT (drmlt.cpp:406-418): the luminance image of the first stage weights the second stage's splats.
// Must be set before drmlt_seed: the chains' current states are normalised with it (drmlt_proc.cpp:514).
int drmlt_set_importance_map(drmlt_ctx *ctx, const float *lum_map_or_null) {
    if (!ctx) return DRMLT_E_INVALID;
    if (ctx->seeded) return ctx->fail(DRMLT_E_STATE, "the importance map must be set before drmlt_seed");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!lum_map_or_null) { ctx->P.importance = nullptr; return DRMLT_OK; }
    const size_t n = (size_t) ctx->P.width * ctx->P.height;
    // Zero entries are legal: a first-stage image with an unlit region has them (mltLuminancePass applies no floor,
    // util.cpp:190-196). SplatList::normalize then divides by zero, the list luminance becomes inf and the chain loop
    // rejects the proposal (isInvalid, drmlt_proc.cpp:428) -- the kernels do the same (normalize_splat, lum_invalid).
    for (size_t i = 0; i < n; ++i)
        if (!(lum_map_or_null[i] >= 0.f) || !std::isfinite(lum_map_or_null[i]))
            return ctx->fail(DRMLT_E_INVALID, "importance map must be non-negative and finite (pixel %zu)", i);
    HIP_TRY(ctx, ctx->d_importance.alloc(n * sizeof(float)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_importance.p, lum_map_or_null, n * sizeof(float), hipMemcpyHostToDevice));
    ctx->P.importance = ctx->d_importance.as<float>();
    return DRMLT_OK;
}

// Tail of BidirectionalUtils::mltLuminancePass (src/libbidir/util.cpp:179-196): luminance of the developed
// first-stage image, up-sampled with a Gaussian reconstruction filter (stddev 0.5, radius 2), clamped boundary
// lookups, results clamped to [0, inf). Separable Resampler of include/mitsuba/core/rfilter.h:123-198,232-290:
// horizontal pass, then vertical, as mitsuba::resample does (src/libcore/bitmap.cpp:2258-2330).
int drmlt_luminance_map(const float *rgb, int w, int h, int W, int H, float *out) {
    if (!rgb || !out || w <= 0 || h <= 0 || W <= 0 || H <= 0) return DRMLT_E_INVALID;
    std::vector<float> lum((size_t) w * h);
    for (size_t i = 0; i < lum.size(); ++i)
        lum[i] = rgb[3 * i] * 0.212671f + rgb[3 * i + 1] * 0.715160f + rgb[3 * i + 2] * 0.072169f;
    auto gauss = [](float x) {
        const float stddev = 0.5f, radius = 2.0f, alpha = -1.0f / (2.0f * stddev * stddev);
        return std::max(0.0f, std::exp(alpha * x * x) - std::exp(alpha * radius * radius));
    };
    // one 1-D pass: src (n_src samples, stride s_src) -> dst (n_dst samples, stride s_dst)
    auto pass = [&](const float *src, int n_src, size_t s_src, float *dst, int n_dst, size_t s_dst) {
        if (n_src == n_dst) { for (int i = 0; i < n_dst; ++i) dst[i * s_dst] = std::max(0.0f, src[i * s_src]); return; }
        float radius = 2.0f, scale = 1.0f, invScale = 1.0f;
        if (n_dst < n_src) { scale = (float) n_src / (float) n_dst; invScale = 1 / scale; radius *= scale; }
        const int taps = (int) std::ceil(radius * 2);
        for (int i = 0; i < n_dst; ++i) {
            const float center = (i + 0.5f) / n_dst * n_src;
            const int start = (int) std::floor(center - radius + 0.5f);
            float wsum = 0.f, wts[64];
            for (int j = 0; j < taps && j < 64; ++j) { wts[j] = gauss((start + j + 0.5f - center) * invScale); wsum += wts[j]; }
            const float norm = 1.0f / wsum;
            float r = 0.f;
            for (int j = 0; j < taps && j < 64; ++j) {
                const int k = std::min(std::max(start + j, 0), n_src - 1); // EClamp
                r += src[k * s_src] * (wts[j] * norm);
            }
            dst[i * s_dst] = std::max(0.0f, r);
        }
    };
    std::vector<float> tmp((size_t) W * h);
    for (int y = 0; y < h; ++y) pass(&lum[(size_t) y * w], w, 1, &tmp[(size_t) y * W], W, 1);
    for (int x = 0; x < W; ++x) pass(&tmp[x], h, (size_t) W, &out[x], H, (size_t) W);
    return DRMLT_OK;
}

int drmlt_set_luminance(drmlt_ctx *ctx, double b) {
    if (!ctx) return DRMLT_E_INVALID;
    if (!(b > 0)) return ctx->fail(DRMLT_E_INVALID, "luminance must be positive");
    ctx->b = b;
    return DRMLT_OK;
}

// k_mutate_mmlt's waves are made of chains of one depth (seed_impl); between the launches of a call they -- and k_mutate_bdpt's -- are
// ALSO regrouped by the work of the launch just done. Chains run free (one path evaluation per lane per pass), so a wave lasts as long as its slowest
// chain, and a chain parked on a glint or a caustic rejects nearly every first stage: two evaluations per mutation, launch after
// launch. Sorted by (depth, evaluations of the last launch), such chains share waves -- full ones, run first -- instead of holding
// sixty-three finished lanes each (config 5 with the E S* L paths counted: 2.33e9 -> see DESIGN 7a). Chain ids, states and
// streams are untouched: the same chains bit for bit, in other lanes. Counting sort, stable: deterministic.
static int regroup_chains(drmlt_ctx *ctx, uint32_t n_mut) {
    const uint32_t n = ctx->n_chains, padded = (n + 63u) / 64u * 64u, md = ctx->cfg.technique == DRMLT_TECH_MMLT ? (uint32_t) ctx->cfg.max_depth : 1u; // (bdpt: no depth classes)
    if (!getenv("DRMLT_REGROUP_ON_HOST")) {
        // on the device, enqueued behind the launch whose counts it reads (kernels_mmlt.hip: launch_regroup): no copy, no synchronisation
        const size_t words = regroup_scratch_words(n, md);
        if (ctx->d_regroup.bytes < words * sizeof(uint32_t)) HIP_TRY(ctx, ctx->d_regroup.alloc(words * sizeof(uint32_t)));
        launch_regroup(ctx->d_done.as<uint32_t>(), ctx->cfg.technique == DRMLT_TECH_MMLT ? ctx->P.chain_depth : nullptr, n, n_mut, md, ctx->d_order.as<uint32_t>(), padded,
                       ctx->d_regroup.as<uint32_t>(), ctx->stream);
        HIP_TRY(ctx, hipGetLastError());
        if (!getenv("DRMLT_REGROUP_CHECK")) return DRMLT_OK;
        // test hook: the device's permutation against the host's stable counting sort of the same counts
        std::vector<uint32_t> got(padded), work(n);
        HIP_TRY(ctx, hipMemcpyAsync(got.data(), ctx->d_order.p, (size_t) padded * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(work.data(), ctx->d_done.p, (size_t) n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<uint32_t> want(n);
        for (uint32_t j = 0; j < n; ++j) want[j] = j;
        auto key = [&](uint32_t j) {
            const uint32_t d = ctx->seed_indices[j] % md, extra = work[j] > n_mut ? work[j] - n_mut : 0u;
            return (md - 1u - d) * 16u + (15u - std::min<uint32_t>(15u, (uint32_t) ((uint64_t) extra * 16u / std::max(1u, n_mut))));
        };
        std::stable_sort(want.begin(), want.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
        for (uint32_t j = 0; j < padded; ++j)
            if (got[j] != (j < n ? want[j] : n)) return ctx->fail(DRMLT_E_DEVICE, "regroup: the device's order differs from the host's stable sort at slot %u (%u instead of %u)", j, got[j], j < n ? want[j] : n);
        ctx->regroup_checks++;
        return DRMLT_OK;
    }
    // the same permutation on the host (round 3's path, kept as the cross-check: tests/test_gpu_node.py compares the two)
    const uint32_t B = 16u;
    std::vector<uint32_t> work(n);
    HIP_TRY(ctx, hipMemcpyAsync(work.data(), ctx->d_done.p, (size_t) n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> start(md * B + 1u, 0u), key(n), order(padded, n);
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t d = ctx->seed_indices[j] % md;                     // deepest first, as at seed time
        const uint32_t extra = work[j] > n_mut ? work[j] - n_mut : 0u;    // second stages and reverse moves
        const uint32_t b = std::min<uint32_t>(B - 1u, (uint32_t) ((uint64_t) extra * B / std::max(1u, n_mut)));
        key[j] = (md - 1u - d) * B + (B - 1u - b);
        ++start[key[j] + 1u];
    }
    for (uint32_t k = 0; k < md * B; ++k) start[k + 1u] += start[k];
    for (uint32_t j = 0; j < n; ++j) order[start[key[j]]++] = j;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_order.p, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // `order` is a local
    return DRMLT_OK;
}

// waves of a chain-kernel launch: k_mutate_v4 carries 32 chains per wave (lane pairs), k_mutate_v5 64
static uint32_t chain_waves(const drmlt_ctx *ctx) { return ctx->P.kernel_variant == 5 ? (ctx->n_chains + 63u) / 64u : (ctx->n_chains + 31u) / 32u; }

int drmlt_run(drmlt_ctx *ctx, uint64_t total_mutations, volatile int *stop, drmlt_progress_cb cb, void *user) {
    if (!ctx) return DRMLT_E_INVALID;
    if (!ctx->seeded) return ctx->fail(DRMLT_E_STATE, "drmlt_run called before drmlt_seed");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t per_chain = total_mutations / ctx->n_chains; // nMutations, drmlt.cpp:475-476
    std::vector<EventPair> evs; // destroyed on every return path
    uint64_t done = 0;
    int rc = DRMLT_OK;
    // "timeout" (drmlt.cpp:296, drmlt_proc.cpp:519-521,868-877): equal-time mode. All chains run concurrently here,
    // so they all stop at the first launch boundary after the deadline (the reference stops handing out work units).
    const bool timed = ctx->cfg.timeout_s > 0;
    const auto t_start = std::chrono::steady_clock::now();
    // Run-ahead (k_mutate_v4, more than one launch to go): a launch ends when every chain has reached its target, and until then
    // chains that are there keep going -- towards the total of THIS call, at most 8192 mutations beyond the target (16-bit event
    // counters per chain and launch). The last launch has target = limit = total: every chain ends at exactly its count.
    // (A single launch has target = limit and is the plain fixed-count launch; the per-chain counts are kept either way.)
    const bool ahead = ctx->cfg.technique == DRMLT_TECH_PATH && ctx->cfg.algo != DRMLT_ALGO_PSSMLT && ctx->P.kernel_variant >= 4 &&
                       !getenv("DRMLT_NO_RUN_AHEAD");
    const uint64_t call_base = ctx->mutation_base, call_end = call_base + per_chain;
    const bool regroup = (ctx->cfg.technique == DRMLT_TECH_MMLT || ctx->cfg.technique == DRMLT_TECH_BDPT) && ctx->cfg.algo != DRMLT_ALGO_PSSMLT && ctx->P.exec_order && !getenv("DRMLT_NO_REGROUP");
    int since_regroup = 0;
    while (done < per_chain) {
        if (stop && *stop) { rc = DRMLT_E_CANCELLED; break; }
        if (timed && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() >= (double) ctx->cfg.timeout_s) break;
        // shorter launches when somebody is watching (cancellation / progress / deadline latency ~ tens of ms)
        // (regrouping: a short first launch -- an eighth of the call, 32 to 256 mutations -- to learn which chains are parked;
        // DRMLT_REGROUP_FIRST overrides its length, clamped to [1, slice]: a launch beyond 32 768 mutations would overflow the
        // kernels' 16-bit event counters)
        uint64_t first_len = std::max<uint64_t>(32, std::min<uint64_t>(256, per_chain / 8));
        if (const char *e = getenv("DRMLT_REGROUP_FIRST")) first_len = (uint64_t) std::max(1, std::min(ctx->slice, atoi(e)));
        const uint64_t slice = (stop || cb || timed) ? std::min(ctx->slice, 256) : (regroup && !ctx->regrouped ? std::min<uint64_t>(ctx->slice, first_len) : (uint64_t) ctx->slice);
        uint32_t n = (uint32_t) std::min<uint64_t>(slice, per_chain - done);
        evs.emplace_back();
        EventPair &ev = evs.back();
        HIP_TRY(ctx, ev.create());
        HIP_TRY(ctx, hipEventRecord(ev.a, ctx->stream));
        ctx->P.luminance_b = (float) ctx->b;
        HIP_TRY(ctx, ensure_overflow(ctx, ctx->P, 2 * (size_t) ctx->P.n_chains_alloc + 128));
        if (ctx->cfg.algo == DRMLT_ALGO_PSSMLT) launch_mutate_pssmlt(ctx->P, n, ctx->mutation_base, ctx->stream);
        else if (ctx->cfg.technique == DRMLT_TECH_MMLT) {
            DParams Q = ctx->P;
            if (regroup) Q.chain_done = ctx->d_done.as<uint32_t>(); // per-chain evaluation counts of this launch
            launch_mutate_mmlt(Q, n, ctx->mutation_base, ctx->stream);
        }
        else if (ctx->cfg.technique == DRMLT_TECH_BDPT) {
            DParams Q = ctx->P;
            if (regroup) Q.chain_done = ctx->d_done.as<uint32_t>(); // per-chain evaluation counts of this launch
            launch_mutate_bdpt(Q, n, ctx->mutation_base, ctx->stream);
        }
        else if (ahead) {
            DParams Q = ctx->P;
            const uint64_t target = call_base + done + n;
            Q.chain_done = ctx->d_done.as<uint32_t>();
            Q.run_limit = (uint32_t) std::min<uint64_t>(call_end, target + (getenv("DRMLT_AHEAD_CAP") ? (uint64_t) atoi(getenv("DRMLT_AHEAD_CAP")) : std::min<uint64_t>(8 * slice, 8192))); // at most eight launches ahead (the per-chain event counters of a launch are 16 bits wide); measured on config 3: 1024 6.7e8, 4096 7.1e8, 8192 7.14e8
            launch_set_u32(Q.waves_left, chain_waves(ctx), ctx->stream);
            launch_mutate(Q, (uint32_t) target, 0u, ctx->stream);
        } else launch_mutate(ctx->P, n, ctx->mutation_base, ctx->stream);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ev.b, ctx->stream));
        ctx->mutation_base += n;
        done += n;
        ctx->launches++;
        // Regrouping is three small kernels on the chains' stream (kernels_mmlt.hip: launch_regroup), after every launch. (Round 3
        // sorted on the host -- a D2H copy, an H2D copy and two stream synchronisations per launch; that path survives behind
        // DRMLT_REGROUP_ON_HOST as the cross-check and is thinned to every fourth launch when somebody is watching.) A failure
        // ends the loop through the normal exit below: counters and timings are kept.
        if (regroup && (!(stop || cb || timed) || !ctx->regrouped || !getenv("DRMLT_REGROUP_ON_HOST") || ++since_regroup >= 4)) {
            rc = regroup_chains(ctx, n);
            if (rc != DRMLT_OK) break;
            ctx->regrouped = true;
            since_regroup = 0;
        }
        if (stop || cb || timed) {
            const hipError_t se = hipStreamSynchronize(ctx->stream);
            if (se != hipSuccess) { rc = ctx->fail(DRMLT_E_DEVICE, "hipStreamSynchronize: %s", hipGetErrorString(se)); break; }
            if (cb) cb(done * ctx->n_chains, per_chain * ctx->n_chains, user);
        }
    }
    { const hipError_t se = hipStreamSynchronize(ctx->stream); if (se != hipSuccess && rc == DRMLT_OK) rc = ctx->fail(DRMLT_E_DEVICE, "hipStreamSynchronize: %s", hipGetErrorString(se)); }
    for (const EventPair &e : evs) {
        const float ms = e.elapsed_ms();
        ctx->kernel_ms += ms;
        ctx->kt_ms += ms;
        ctx->kt_launches++;
    }
    if (ahead && done < per_chain && done > 0 && (rc == DRMLT_OK || rc == DRMLT_E_CANCELLED)) {
        // stopped early (cancel / timeout): chains are at the last target or up to eight launches beyond it. One catch-up launch
        // brings everybody to the most advanced chain's count, so that a stopped render, too, has run every chain equally long.
        std::vector<uint32_t> h(ctx->n_chains);
        HIP_TRY(ctx, hipMemcpy(h.data(), ctx->d_done.p, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const uint32_t top = *std::max_element(h.begin(), h.end());
        if (top > ctx->mutation_base) {
            DParams Q = ctx->P;
            Q.chain_done = ctx->d_done.as<uint32_t>();
            Q.run_limit = top;
            launch_set_u32(Q.waves_left, chain_waves(ctx), ctx->stream);
            launch_mutate(Q, top, 0u, ctx->stream);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            done += top - ctx->mutation_base;
            ctx->mutation_base = top;
            ctx->launches++;
        }
    }
    ctx->mutations += done * ctx->n_chains; // whatever ended the loop: what was launched is counted
    if (rc == DRMLT_E_CANCELLED) return ctx->fail(rc, "cancelled");
    return rc;
}

int drmlt_kernel_time(drmlt_ctx *ctx, double *avg_ms, uint64_t *launches, int reset) {
    if (!ctx) return DRMLT_E_INVALID;
    if (avg_ms) *avg_ms = ctx->kt_launches ? ctx->kt_ms / (double) ctx->kt_launches : 0.0;
    if (launches) *launches = ctx->kt_launches;
    if (reset) { ctx->kt_ms = 0.0; ctx->kt_launches = 0; }
    return DRMLT_OK;
}

int drmlt_develop(drmlt_ctx *ctx, const float *direct_rgb_or_null, float *out_rgb) {
    if (!ctx || !out_rgb) return DRMLT_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t npix = (uint32_t) ctx->P.width * ctx->P.height, n = npix * 3;
    DevBuf d_sum, d_out, d_direct;
    HIP_TRY(ctx, d_sum.alloc(sizeof(double)));
    HIP_TRY(ctx, d_out.alloc((size_t) n * sizeof(float)));
    HIP_TRY(ctx, hipMemsetAsync(d_sum.p, 0, sizeof(double), ctx->stream));
    launch_lum_sum(ctx->P.film, ctx->P.importance, npix, d_sum.as<double>(), ctx->stream);
    double sum = 0.0;
    HIP_TRY(ctx, hipMemcpyAsync(&sum, d_sum.p, sizeof sum, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double avg = sum / (double) npix;
    double factor = ctx->cfg.acceptance_map ? 1.0 : ctx->b / avg; // drmlt_proc.cpp:834-839
    if (direct_rgb_or_null) {
        HIP_TRY(ctx, d_direct.alloc((size_t) n * sizeof(float)));
        HIP_TRY(ctx, hipMemcpyAsync(d_direct.p, direct_rgb_or_null, (size_t) n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    }
    launch_develop(ctx->P.film, d_direct.as<float>(), ctx->P.importance, (float) factor, n, d_out.as<float>(), ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_rgb, d_out.p, (size_t) n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DRMLT_OK;
}

int drmlt_stats_get(drmlt_ctx *ctx, drmlt_stats *o) {
    if (!ctx || !o) return DRMLT_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long v[32];
    HIP_TRY(ctx, hipMemcpyAsync(v, ctx->d_stats.p, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if ((ctx->P.debug & 128) && ctx->cfg.technique == DRMLT_TECH_BDPT) // diagnostic stamps of eval_bdpt / k_mutate_bdpt
        fprintf(stderr, "[drmlt stamps] bdpt cycles per wave, summed: walks %llu pair loop %llu whole chain loop %llu | evaluations %llu | stages (all evaluations) %llu, weights + splats %llu, counters + commit %llu\n",
                v[16], v[17], v[18], v[19], v[20], v[21], v[22]);
    else if ((ctx->P.debug & 128) && ctx->P.kernel_variant == 5) // diagnostic stamps of k_mutate_v5
        fprintf(stderr, "[drmlt v5] cycles per wave, summed: bookkeeping %llu step %llu trace %llu | outer iterations %llu, bookkeeping branches %llu (%.1f chains each), "
                        "stepping chains per iteration %.1f, trace phases %llu starting with %.1f lanes, refills %llu\n",
                v[16], v[18], v[17], v[19], v[23], v[23] ? (double) v[24] / v[23] : 0.0, v[19] ? (double) v[25] / v[19] : 0.0, v[20], v[20] ? (double) v[21] / v[20] : 0.0, v[22]);
    else if (ctx->P.debug & 128) // diagnostic stamps of k_mutate_v3 / v4
        fprintf(stderr, "[drmlt stamps] cycles: mh %llu trace %llu step %llu | iterations %llu mh-branches %llu tracing-lanes %llu\n",
                v[16], v[17], v[18], v[19], v[20], v[21]),
        fprintf(stderr, "[drmlt stamps] mh sections: decide+splat %llu commit %llu start %llu fill %llu\n", v[22], v[23], v[24], v[25]),
        fprintf(stderr, "[drmlt stamps] iterations by chains tracing (of 32): 0: %llu, 1-4: %llu, 5-8: %llu, 9-16: %llu, 17-24: %llu, 25-32: %llu\n", v[26], v[27], v[28], v[29], v[30], v[31]);
    if (getenv("DRMLT_VERBOSE") && v[12])
        fprintf(stderr, "[drmlt bvh] wave iterations: inner %llu (%.1f lanes each), leaf %llu (%.1f lanes each)\n", v[12], (double) v[10] / (double) v[12], v[13],
                v[13] ? (double) v[11] / (double) v[13] : 0.0);
    if ((ctx->P.debug & 1024) && v[20] && ctx->P.kernel_variant != 5)
        fprintf(stderr, "[drmlt bvh] lanes at slice start, of 64: tracing %.1f, chain waiting for its partner %.1f, chain parked for bookkeeping %.1f, helper idle %.1f, flush %.1f (%llu slices)\n",
                (double) v[21] / v[20], (double) v[22] / v[20], (double) v[23] / v[20], (double) v[24] / v[20], (double) v[25] / v[20], v[20]);
    memset(o, 0, sizeof *o);
    const uint64_t M = ctx->mutations;
    const uint64_t n_large = v[0], acc1_l = v[1], acc1_b = v[2], sec_l = v[3], sec_b = v[4], acc2_l = v[5], acc2_b = v[6], n_rev = v[7];
    if (ctx->cfg.algo == DRMLT_ALGO_PSSMLT) { // pssmlt_proc.cpp:230-260: one acceptance test per mutation
        o->overall_base = M;               o->overall_acc = acc1_l + acc1_b;
        o->large_base = n_large;           o->large_acc = acc1_l;
        o->bold_base = M - n_large;        o->bold_acc = acc1_b;
    } else if (!ctx->cfg.use_mixture) { // drmlt_proc.cpp:715-768
        o->first_base = M;                 o->first_acc = acc1_l + acc1_b;
        o->large_base = n_large;           o->large_acc = acc1_l;
        o->bold_base = M - n_large;        o->bold_acc = acc1_b;
        o->second_base = sec_l + sec_b;    o->second_acc = acc2_l + acc2_b;
        o->second_large_base = sec_l;      o->second_large_acc = acc2_l;
        o->second_bold_base = sec_b;       o->second_bold_acc = acc2_b;
        o->overall_base = M + sec_l + sec_b;
        o->overall_acc = acc1_l + acc1_b + acc2_l + acc2_b;
    } else { // drmlt_proc.cpp:342-377
        o->second_base = sec_b;            o->second_acc = acc2_b;
        o->first_base = M - sec_b;         o->first_acc = acc1_l + acc1_b;
        o->large_base = n_large;           o->large_acc = acc1_l;
        o->bold_base = M - n_large - sec_b; o->bold_acc = acc1_b;
        o->overall_base = M;               o->overall_acc = acc1_l + acc1_b + acc2_b;
    }
    o->mutations = M;
    o->path_evals = M + sec_l + sec_b + n_rev;
    o->rays = v[8];
    o->accepted = acc1_l + acc1_b + acc2_l + acc2_b;
    o->kernel_ms = ctx->kernel_ms;
    o->seed_ms = ctx->seed_ms;
    o->n_chains = ctx->n_chains;
    o->max_dim = (uint32_t) ctx->P.max_dim;
    o->launches = ctx->launches;
    o->bvh_node_visits = v[10]; o->bvh_prim_tests = v[11]; o->bvh_node_iterations = v[12]; o->bvh_leaf_iterations = v[13];
    return DRMLT_OK;
}

int drmlt_eval_paths(drmlt_ctx *ctx, const float *u, uint32_t n, uint32_t dim, drmlt_splat *out) {
    if (!ctx || !u || !out) return DRMLT_E_INVALID;
    const bool mmlt = ctx->cfg.technique == DRMLT_TECH_MMLT;
    if (ctx->cfg.technique == DRMLT_TECH_BDPT) return ctx->fail(DRMLT_E_INVALID, "technique=bdpt evaluates to splat lists: use drmlt_eval_lists");
    const int need = ctx->P.eff_dim + (mmlt ? 1 : 0); // mmlt: [sensor S | emitter E | direct | depth]
    if ((int) dim < need) return ctx->fail(DRMLT_E_INVALID, "eval_paths: need at least %d PSS dimensions per point", need);
    if (n == 0) return DRMLT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf d_u, d_o;
    HIP_TRY(ctx, d_u.alloc((size_t) n * dim * sizeof(float)));
    HIP_TRY(ctx, d_o.alloc((size_t) n * 8 * sizeof(float)));
    HIP_TRY(ctx, hipMemcpyAsync(d_u.p, u, (size_t) n * dim * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, ensure_overflow(ctx, ctx->P, std::max<size_t>(n, (size_t) ctx->P.n_chains_alloc)));
    if (mmlt) launch_eval_paths_mmlt(ctx->P, d_u.as<float>(), n, dim, d_o.as<float>(), ctx->stream);
    else launch_eval_paths(ctx->P, d_u.as<float>(), n, dim, d_o.as<float>(), ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    std::vector<float> h((size_t) n * 8);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), d_o.p, h.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t i = 0; i < n; ++i) {
        const float *r = &h[(size_t) i * 8];
        out[i].luminance = r[0]; out[i].x = r[1]; out[i].y = r[2];
        out[i].rgb[0] = r[3]; out[i].rgb[1] = r[4]; out[i].rgb[2] = r[5];
        memcpy(&out[i].n_dims, &r[6], 4);
        memcpy(&out[i].n_rays, &r[7], 4);
    }
    return DRMLT_OK;
}

int drmlt_film_read(drmlt_ctx *ctx, float *out_rgb) {
    if (!ctx || !out_rgb) return DRMLT_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(out_rgb, ctx->d_film.p, ctx->film_floats * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DRMLT_OK;
}

int drmlt_film_clear(drmlt_ctx *ctx) {
    if (!ctx) return DRMLT_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_film.p, 0, ctx->d_film.bytes, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DRMLT_OK;
}

void *drmlt_film_device_ptr(drmlt_ctx *ctx) { return ctx ? ctx->d_film.p : nullptr; }

int drmlt_render_pt(drmlt_ctx *ctx, uint32_t spp, uint64_t seed, float *out_rgb) {
    if (!ctx || !out_rgb || spp == 0) return DRMLT_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DParams P = ctx->P;
    DevBuf film;
    const size_t bytes = (size_t) P.width * P.height * 3 * sizeof(float);
    HIP_TRY(ctx, film.alloc(bytes));
    HIP_TRY(ctx, hipMemsetAsync(film.p, 0, bytes, ctx->stream));
    P.film = film.as<float>();
    P.key0 = (uint32_t) seed; P.key1 = (uint32_t) (seed >> 32);
    const uint64_t n = (uint64_t) spp * P.width * P.height;
    HIP_TRY(ctx, ensure_overflow(ctx, P, (size_t) 16384 * 64));
    launch_render_pt(P, n, 0u, 1.0f / (float) spp, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_rgb, film.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DRMLT_OK;
}

int drmlt_eval_lists(drmlt_ctx *ctx, const float *u, uint32_t n, uint32_t dim, float *out, uint32_t stride) {
    if (!ctx || !u || !out) return DRMLT_E_INVALID;
    if (ctx->cfg.technique != DRMLT_TECH_BDPT) return ctx->fail(DRMLT_E_INVALID, "drmlt_eval_lists is for technique=bdpt");
    if ((int) dim < ctx->P.eff_dim) return ctx->fail(DRMLT_E_INVALID, "eval_lists: need at least %d PSS dimensions per point", ctx->P.eff_dim);
    if (stride < 10) return ctx->fail(DRMLT_E_INVALID, "eval_lists: stride must be at least 10 floats");
    if (n == 0) return DRMLT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf d_u, d_o;
    HIP_TRY(ctx, d_u.alloc((size_t) n * dim * sizeof(float)));
    HIP_TRY(ctx, d_o.alloc((size_t) n * stride * sizeof(float)));
    HIP_TRY(ctx, hipMemcpyAsync(d_u.p, u, (size_t) n * dim * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, ensure_overflow(ctx, ctx->P, (size_t) ctx->P.n_chains_alloc));
    launch_eval_lists_bdpt(ctx->P, d_u.as<float>(), n, dim, d_o.as<float>(), stride, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, d_o.p, (size_t) n * stride * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DRMLT_OK;
}

int drmlt_chain_state(drmlt_ctx *ctx, drmlt_splat *cur, float *u, uint32_t dim) {
    if (!ctx) return DRMLT_E_INVALID;
    if (!ctx->seeded) return ctx->fail(DRMLT_E_STATE, "chain_state before seed");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n = ctx->n_chains;
    if (cur && ctx->cfg.technique == DRMLT_TECH_BDPT) {
        // current splat list (slot 0, unnormalised): luminance, main splat; n_dims = has main splat, n_rays = light-image splats
        std::vector<float> h((size_t) 7 * n), l(n);
        HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->d_bd_lists.p, h.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(l.data(), ctx->d_cur.p, l.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t i = 0; i < n; ++i) {
            int32_t meta; memcpy(&meta, &h[(size_t) n + i], 4);
            const float inv = l[i] > 0.f ? 1.f / l[i] : 0.f;
            cur[i].luminance = l[i]; cur[i].x = h[2 * (size_t) n + i]; cur[i].y = h[3 * (size_t) n + i];
            cur[i].rgb[0] = h[4 * (size_t) n + i] * inv; cur[i].rgb[1] = h[5 * (size_t) n + i] * inv; cur[i].rgb[2] = h[6 * (size_t) n + i] * inv;
            cur[i].n_dims = meta & 1; cur[i].n_rays = meta >> 1;
        }
        cur = nullptr;
    }
    if (cur) {
        std::vector<float> h((size_t) 6 * n);
        HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->d_cur.p, h.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t i = 0; i < n; ++i) {
            cur[i].luminance = h[i]; cur[i].x = h[n + i]; cur[i].y = h[2 * (size_t) n + i];
            cur[i].rgb[0] = h[3 * (size_t) n + i]; cur[i].rgb[1] = h[4 * (size_t) n + i]; cur[i].rgb[2] = h[5 * (size_t) n + i];
            cur[i].n_dims = 0; cur[i].n_rays = 0;
        }
        if (ctx->cfg.technique == DRMLT_TECH_MMLT) { // n_dims: the chain's path depth, n_rays: t of the current state
            std::vector<int32_t> ci((size_t) 2 * n);
            HIP_TRY(ctx, hipMemcpyAsync(ci.data(), ctx->d_chain_i.p, ci.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            for (uint32_t i = 0; i < n; ++i) { cur[i].n_dims = ci[i]; cur[i].n_rays = ci[n + i]; }
        }
    }
    if (u) {
        std::vector<float> h((size_t) ctx->P.eff_dim * n);
        HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->d_x.p, h.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t k = 0; k < dim; ++k) u[(size_t) i * dim + k] = k < (uint32_t) ctx->P.eff_dim ? h[(size_t) k * n + i] : 0.f;
    }
    return DRMLT_OK;
}

} // extern "C"
