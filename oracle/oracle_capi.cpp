// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
//
// C entry points over the templated restatement (float and double builds),
// loaded by tests/ through ctypes. Consumes the same POD structs as the product
// ABI (include/drmlt_abi.h) so both sides see identical inputs.
#include "oracle_process.hpp"
#include "oracle_bidir.hpp"
#include <atomic>
#include <cstdio>
#include <thread>

using namespace oracle;

namespace {

template <typename F> struct ArraySampler : Sampler<F> {
    const float *u;
    size_t n;
    ArraySampler(const float *p, size_t cnt) : u(p), n(cnt) {}
    F next1D() override {
        size_t k = this->sampleIndex++;
        if (k >= n) throw std::runtime_error("ArraySampler: out of dimensions");
        return (F) u[k];
    }
};

struct CtxBase {
    virtual ~CtxBase() = default;
    virtual int evalPaths(const float *u, uint32_t n, uint32_t dim, drmlt_splat *out) = 0;
    virtual int seed(uint64_t seed, uint32_t chainOffset, double *b, uint32_t poolChains = 0, const uint32_t *indices = nullptr) = 0;
    virtual int run(uint64_t total, int nthreads) = 0;
    virtual int filmRead(float *out) = 0;
    virtual int developImage(const float *direct, float *out) = 0;
    virtual int stats(drmlt_stats *out) = 0;
    virtual int chainState(drmlt_splat *cur, float *u, uint32_t dim) = 0;
    virtual int renderPT(uint32_t spp, uint64_t seed, int nthreads, float *out) = 0;
    virtual int bootstrapLum(uint64_t seed, uint32_t stream, uint32_t n, float *out) = 0;
    virtual int setImportance(const float *map) = 0;
    std::vector<uint32_t> pickedSeeds; // bootstrap sample indices of this context's chains after the last seed()
    virtual int bdptRender(uint64_t n, uint64_t seed, int nthreads, float *out) = 0;
    virtual int bdptEval(const float *uSensor, const float *uEmitter, const float *uDirect, uint32_t n, uint32_t dim, float *out, uint32_t stride) = 0;
    virtual int mmltRender(int depth, uint64_t n, uint64_t seed, int lightImage, int nthreads, float *out, double *strat) = 0;
    virtual int mmltEval(int depth, int lightImage, const float *uSensor, const float *uEmitter, const float *uDirect,
                         uint32_t n, uint32_t dim, drmlt_splat *out, int *st) = 0;
    std::string error;
};

template <typename F> struct Ctx : CtxBase {
    drmlt_config cfg;
    Config<F> c;
    Scene<F> scene;
    SceneEvaluator<F> eval;
    std::vector<std::unique_ptr<DRChain<F, SceneEvaluator<F>>>> chains;
    std::vector<std::unique_ptr<PSSMLTChain<F, SceneEvaluator<F>>>> pchains;
    using MChain = DRChain<F, MMLTEvaluator<F>, MMLTSamplers<F>>;
    using BChain = DRChain<F, BDPTEvaluator<F>, MMLTSamplers<F>>;
    MMLTEvaluator<F> meval;
    BDPTEvaluator<F> beval;
    std::vector<std::unique_ptr<MChain>> mchains;
    std::vector<std::unique_ptr<BChain>> bchains;
    bool mmlt = false, bdpt = false;
    std::vector<double> accum;
    Stats st;
    double b = 0;
    bool seeded = false;

    std::string init(const drmlt_config &in, const drmlt_scene &s) {
        cfg = in;
        mmlt = in.technique == DRMLT_TECH_MMLT;
        bdpt = in.technique == DRMLT_TECH_BDPT;
        if (bdpt && in.algo == DRMLT_ALGO_PSSMLT) return "oracle: pssmlt over technique=bdpt is not restated";
        if (mmlt && in.max_depth <= 0) return "Impossible to use MMLT with no max depth"; // drmlt.cpp:213-215
        if (mmlt && in.algo == DRMLT_ALGO_PSSMLT) return "oracle: pssmlt over technique=mmlt is not restated";
        // A rejected large step re-draws the strategy; its second stage (and Green's reverse path) then reads an
        // emitter state that may be empty (drmlt_sampler.cpp:189-191 with an unused emitter sampler): undefined
        // behaviour in the reference, refused here.
        if (mmlt && in.timid_after_large) return "timidAfterLarge is not defined for technique=mmlt";
        if (!mmlt && in.fix_emitter_path) return "Impossible to use fixEmitterPath without MMLT"; // drmlt.cpp:333-337
        if (in.max_depth <= 0) return "technique=path requires a finite maxDepth (pssmlt_utils.h:63)";
        if (in.scale_second > 1) return "scaleSecond is bigger than the first stage";
        if (in.work_units <= 0) return "work_units must be positive";
        std::string e = scene.load(s);
        if (!e.empty()) return e;
        c.algo = in.algo; c.type = in.type; c.maxDepth = in.max_depth; c.rrDepth = in.rr_depth;
        c.separateDirect = in.direct_samples >= 0;
        c.acceptanceMap = in.acceptance_map != 0; c.timidAfterLarge = in.timid_after_large != 0;
        c.useMixture = in.use_mixture != 0; c.kelemenWeights = in.kelemen_style_weights != 0;
        c.kelemenMutation = in.kelemen_style_mutation != 0;
        c.pLarge = in.p_large; c.sigma = in.sigma; c.scaleSecond = in.scale_second;
        c.maxDim = findMaxDimensionsPath(in.max_depth, in.rr_depth);
        c.fixEmitterPath = in.fix_emitter_path != 0; c.lightImage = in.no_light_image == 0; c.technique = in.technique;
        c.directSampling = bdpt && !in.no_direct_sampling; // drmlt.cpp:228-231 (forced off for mmlt)
        if (in.seed_rule != DRMLT_SEED_TARGET && in.seed_rule != DRMLT_SEED_REFERENCE) return "Unknown seeding rule";
        c.seedByTarget = in.seed_rule == DRMLT_SEED_TARGET; // the product's default; DRMLT_SEED_REFERENCE = pathsampler.cpp:901-905
        beval = BDPTEvaluator<F>{&scene, c.maxDepth, c.rrDepth, c.separateDirect, c.lightImage, c.directSampling};
        meval = MMLTEvaluator<F>{&scene, c.maxDepth, c.separateDirect, c.lightImage};
        if (c.acceptanceMap && scene.filterType != DRMLT_FILTER_BOX) return "Box filter required for acceptance map!";
        eval = SceneEvaluator<F>{&scene, c.maxDepth, c.rrDepth, c.separateDirect};
        accum.assign((size_t) scene.width * scene.height * 3, 0.0);
        return "";
    }

    int evalPaths(const float *u, uint32_t n, uint32_t dim, drmlt_splat *out) override {
        for (uint32_t i = 0; i < n; ++i) {
            ArraySampler<F> s(u + (size_t) i * dim, dim);
            SplatList<F> l;
            eval(s, l, nullptr);
            out[i].luminance = (float) l.luminance;
            out[i].x = (float) l.px; out[i].y = (float) l.py;
            out[i].rgb[0] = (float) l.value.x; out[i].rgb[1] = (float) l.value.y; out[i].rgb[2] = (float) l.value.z;
            out[i].n_dims = l.nDims; out[i].n_rays = l.nRays;
        }
        return 0;
    }

    int bootstrapLum(uint64_t seedv, uint32_t stream, uint32_t n, float *out) override {
        Random boot(seedv, stream);
        ReplayableSampler<F> s(&boot);
        SplatList<F> l;
        for (uint32_t i = 0; i < n; ++i) {
            s.setSampleIndex(i);
            if (mmlt) { s.depth = (int) (i % (uint32_t) cfg.max_depth) + 1; meval(s, l, nullptr); out[i] = (float) l.luminance; continue; }
            if (bdpt) { beval(s, l, nullptr); out[i] = (float) l.luminance; continue; }
            eval(s, l, nullptr);
            out[i] = (float) l.luminance;
        }
        return 0;
    }

    // poolChains > 0: ONE seed pool for a job split over several participants (the product's drmlt_seed_pool):
    // bootstrap stream 0, sized for poolChains chains, poolChains seeds drawn and sorted; this context takes seeds and
    // chain ids [chainOffset, chainOffset + work_units). Every participant finds the same list and the same b.
    // indices != nullptr: the chains start from THESE bootstrap samples (work_units sample indices, e.g. the ones a device
    // context picked: its fp32 luminances shift the CDF under the picks) instead of the oracle's own
    // picks; b still comes from the oracle's bootstrap. Lets a test run the very same chains on both sides.
    int seed(uint64_t seedv, uint32_t chainOffset, double *bOut, uint32_t poolChains = 0, const uint32_t *indices = nullptr) override {
        const bool pool = poolChains > 0;
        if (pool && (uint64_t) chainOffset + (uint64_t) cfg.work_units > poolChains) { error = "seed pool does not cover this context's chains"; return DRMLT_E_INVALID; }
        const uint32_t bootStream = pool ? 0u : chainOffset;
        const size_t nSelect = pool ? (size_t) poolChains : (size_t) cfg.work_units;
        Random boot(seedv, bootStream);
        std::vector<PathSeed> seeds;
        // drmlt.cpp:454-473; technique=mmlt: x50 and one share per depth (initialisation on one core)
        size_t lumSamples = (size_t) std::max<long long>(cfg.luminance_samples, (long long) nSelect * (mmlt ? 50 : 10));
        if (mmlt) lumSamples *= (size_t) cfg.max_depth;
        const float *target = c.seedByTarget ? c.importance : nullptr; // two-stage MLT under DRMLT_SEED_TARGET: seeds from f / importance
        b = mmlt ? generateSeeds<F>(meval, boot, lumSamples, nSelect, seeds, nullptr, cfg.max_depth, target, c.impW, c.impH)
            : bdpt ? generateSeeds<F>(beval, boot, lumSamples, nSelect, seeds, nullptr, 0, target, c.impW, c.impH)
                   : generateSeeds<F>(eval, boot, lumSamples, nSelect, seeds, nullptr, 0, target, c.impW, c.impH);
        if (pool && b != 0) { // this context's slice of the job's list (a temporary: assign() from a vector's own iterators is undefined)
            std::vector<PathSeed> mine(seeds.begin() + chainOffset, seeds.begin() + chainOffset + cfg.work_units);
            seeds.swap(mine);
        }
        if (indices && b != 0) {
            ReplayableSampler<F> rs(&boot);
            SplatList<F> list;
            seeds.clear();
            for (int i = 0; i < cfg.work_units; ++i) {
                const uint32_t idx = indices[i];
                const int depth = mmlt ? (int) (idx % (uint32_t) cfg.max_depth) + 1 : -1; // pathsampler.cpp:884-890
                rs.setSampleIndex(idx);
                rs.depth = depth;
                if (mmlt) meval(rs, list, nullptr); else if (bdpt) beval(rs, list, nullptr); else eval(rs, list, nullptr);
                if (!(list.luminance > 0)) { error = "seed index with zero luminance in the oracle's arithmetic"; return DRMLT_E_REPLAY; }
                seeds.push_back(PathSeed{idx, (double) list.luminance, depth});
            }
        }
        if (b == 0) { error = "The average image luminance appears to be zero!"; return DRMLT_E_ZERO_LUM; }
        pickedSeeds.clear();
        for (const PathSeed &ps : seeds) pickedSeeds.push_back(ps.sampleIndex);
        if (cfg.acceptance_map) b = 1.0;                               // drmlt.cpp:550-552
        else if (cfg.average_luminance != -1.0f) b = cfg.average_luminance; // :555-558
        c.luminance = (F) b;
        chains.clear(); pchains.clear(); mchains.clear(); bchains.clear();
        for (int i = 0; i < cfg.work_units; ++i) {
            bool ok;
            if (bdpt) {
                bchains.emplace_back(new BChain(c, beval, seedv, chainOffset + i, bootStream));
                ok = bchains.back()->init(seeds[i]);
            } else if (mmlt) {
                mchains.emplace_back(new MChain(c, meval, seedv, chainOffset + i, bootStream));
                ok = mchains.back()->init(seeds[i]);
            } else if (cfg.algo == DRMLT_ALGO_PSSMLT) {
                pchains.emplace_back(new PSSMLTChain<F, SceneEvaluator<F>>(c, eval, seedv, chainOffset + i, bootStream));
                ok = pchains.back()->init(seeds[i]);
            } else {
                chains.emplace_back(new DRChain<F, SceneEvaluator<F>>(c, eval, seedv, chainOffset + i, bootStream));
                ok = chains.back()->init(seeds[i]);
            }
            if (!ok) { error = "Error when reconstructing a seed path"; return DRMLT_E_REPLAY; }
        }
        seeded = true;
        if (bOut) *bOut = b;
        return 0;
    }

    int run(uint64_t total, int nthreads) override {
        if (!seeded) { error = "run before seed"; return DRMLT_E_STATE; }
        uint64_t perChain = total / (uint64_t) cfg.work_units; // drmlt.cpp:475-476
        nthreads = std::max(1, nthreads);
        std::vector<Stats> tstats(nthreads);
        std::vector<std::unique_ptr<Film<F>>> films;
        for (int t = 0; t < nthreads; ++t)
            films.emplace_back(new Film<F>(scene.width, scene.height, scene.filterType, scene.filterParam));
        std::atomic<int> next(0);
        auto work = [&](int t) {
            for (;;) {
                int i = next.fetch_add(1);
                if (i >= cfg.work_units) break;
                if (bdpt) bchains[i]->run(perChain, *films[t], tstats[t]);
                else if (mmlt) mchains[i]->run(perChain, *films[t], tstats[t]);
                else if (cfg.algo == DRMLT_ALGO_PSSMLT) pchains[i]->run(perChain, *films[t], tstats[t]);
                else chains[i]->run(perChain, *films[t], tstats[t]);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        for (int t = 0; t < nthreads; ++t) { films[t]->accumulateInto(accum); st.add(tstats[t]); }
        return 0;
    }

    std::vector<float> importance;
    int setImportance(const float *map) override {
        if (seeded) { error = "the importance map must be set before seed"; return DRMLT_E_STATE; }
        if (!map) { importance.clear(); c.importance = nullptr; return 0; }
        importance.assign(map, map + (size_t) scene.width * scene.height);
        c.importance = importance.data(); c.impW = scene.width; c.impH = scene.height;
        return 0;
    }
    int filmRead(float *out) override {
        for (size_t i = 0; i < accum.size(); ++i) out[i] = (float) accum[i];
        return 0;
    }
    int developImage(const float *direct, float *out) override {
        develop(accum, scene.width, scene.height, b, cfg.acceptance_map != 0, direct, out, c.importance);
        return 0;
    }
    int stats(drmlt_stats *o) override {
        std::memset(o, 0, sizeof(*o));
        std::memcpy(&o->first_acc, &st.first_acc, sizeof(uint64_t) * 18);
        o->n_chains = (uint32_t) cfg.work_units;
        o->max_dim = (uint32_t) c.maxDim;
        return 0;
    }
    int chainState(drmlt_splat *cur, float *u, uint32_t dim) override {
        if (mmlt) { // the product's layout: [sensor S | emitter E | direct], S = 2 (maxDepth + 1), E = 2 maxDepth
            const uint32_t S = 2u * (uint32_t) (cfg.max_depth + 1), E = 2u * (uint32_t) cfg.max_depth;
            if (u && dim < S + E + 1) throw std::runtime_error("chain_state: need S + E + 1 dims for technique=mmlt");
            for (int i = 0; i < cfg.work_units; ++i) {
                const SplatList<F> &l = mchains[i]->current();
                const MMLTSamplers<F> &ms = mchains[i]->sampler();
                if (cur) {
                    cur[i].luminance = (float) l.luminance; cur[i].x = (float) l.px; cur[i].y = (float) l.py;
                    cur[i].rgb[0] = (float) l.value.x; cur[i].rgb[1] = (float) l.value.y; cur[i].rgb[2] = (float) l.value.z;
                    cur[i].n_dims = ms.depth; cur[i].n_rays = l.t; // as drmlt_chain_state: depth and t
                }
                if (u) {
                    float *row = u + (size_t) i * dim;
                    for (uint32_t k = 0; k < dim; ++k) row[k] = 0.f;
                    for (uint32_t k = 0; k < S && k < ms.sensor.uCurrent.size(); ++k) row[k] = (float) ms.sensor.uCurrent[k];
                    for (uint32_t k = 0; k < E && k < ms.emitter.uCurrent.size(); ++k) row[S + k] = (float) ms.emitter.uCurrent[k];
                    if (!ms.direct.uCurrent.empty()) row[S + E] = (float) ms.direct.uCurrent[0];
                }
            }
            return 0;
        }
        if (bdpt) { // [sensor S | emitter E | direct Dd]: the components the walks / direct strategies can consume (see drmlt_abi.h)
            const uint32_t rr = (uint32_t) std::max(0, cfg.max_depth + 1 - std::max(cfg.rr_depth, 0));
            uint32_t S = 2u * (uint32_t) (cfg.max_depth + 1) + rr, E = 2u * (uint32_t) cfg.max_depth + (rr > 0 ? rr - 1 : 0);
            S += S & 1u; E += E & 1u;
            const uint32_t Dd = c.directSampling ? (uint32_t) findDirectDimensionsBDPT(cfg.max_depth) : 0u;
            if (u && dim < S + E + Dd) throw std::runtime_error("chain_state: need S + E + Dd dims for technique=bdpt");
            for (int i = 0; i < cfg.work_units; ++i) {
                const SplatList<F> &l = bchains[i]->current();
                const MMLTSamplers<F> &ms = bchains[i]->sampler();
                if (cur) {
                    cur[i].luminance = (float) l.luminance; cur[i].x = (float) l.px; cur[i].y = (float) l.py;
                    cur[i].rgb[0] = (float) l.value.x; cur[i].rgb[1] = (float) l.value.y; cur[i].rgb[2] = (float) l.value.z;
                    cur[i].n_dims = l.hasMain ? 1 : 0; cur[i].n_rays = (int) l.more.size(); // as drmlt_chain_state
                }
                if (u) {
                    float *row = u + (size_t) i * dim;
                    for (uint32_t k = 0; k < dim; ++k) row[k] = 0.f;
                    for (uint32_t k = 0; k < S && k < ms.sensor.uCurrent.size(); ++k) row[k] = (float) ms.sensor.uCurrent[k];
                    for (uint32_t k = 0; k < E && k < ms.emitter.uCurrent.size(); ++k) row[S + k] = (float) ms.emitter.uCurrent[k];
                    for (uint32_t k = 0; k < Dd && k < ms.direct.uCurrent.size(); ++k) row[S + E + k] = (float) ms.direct.uCurrent[k];
                }
            }
            return 0;
        }
        for (int i = 0; i < cfg.work_units; ++i) {
            const SplatList<F> &l = mmlt ? mchains[i]->current() : cfg.algo == DRMLT_ALGO_PSSMLT ? pchains[i]->current() : chains[i]->current();
            const std::vector<F> x = mmlt ? mchains[i]->state() : cfg.algo == DRMLT_ALGO_PSSMLT ? pchains[i]->state() : chains[i]->state();
            if (cur) {
                cur[i].luminance = (float) l.luminance; cur[i].x = (float) l.px; cur[i].y = (float) l.py;
                cur[i].rgb[0] = (float) l.value.x; cur[i].rgb[1] = (float) l.value.y; cur[i].rgb[2] = (float) l.value.z;
                cur[i].n_dims = l.nDims; cur[i].n_rays = l.nRays;
            }
            if (u) for (uint32_t k = 0; k < dim; ++k) u[(size_t) i * dim + k] = k < x.size() ? (float) x[k] : 0.f;
        }
        return 0;
    }

    // independent-sample rendering of the same integrand f(u): mean over spp*W*H
    // uniform PSS points, splatted through the same film, scaled to radiance units.
    int renderPT(uint32_t spp, uint64_t seedv, int nthreads, float *out) override {
        nthreads = std::max(1, nthreads);
        uint64_t total = (uint64_t) spp * scene.width * scene.height;
        std::vector<std::unique_ptr<Film<F>>> films;
        for (int t = 0; t < nthreads; ++t)
            films.emplace_back(new Film<F>(scene.width, scene.height, scene.filterType, scene.filterParam));
        auto work = [&](int t) {
            Random r(seedv, (uint32_t) t);
            ReplayableSampler<F> s(&r);
            SplatList<F> l;
            for (uint64_t i = (uint64_t) t; i < total; i += (uint64_t) nthreads) {
                r.seek(TAG_PT, (uint32_t) (i / (uint64_t) nthreads), 0);
                s.sampleIndex = 0;
                eval(s, l, nullptr);
                if (l.luminance > 0 && spectrumValid(l.value)) films[t]->put(l.px, l.py, l.value);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        std::vector<double> acc((size_t) scene.width * scene.height * 3, 0.0);
        for (auto &f : films) f->accumulateInto(acc);
        // E[f(u) on pixel p] * (W*H) = pixel radiance estimate (samplePos uniform over the film)
        double scale = 1.0 / (double) spp;
        for (size_t i = 0; i < acc.size(); ++i) out[i] = (float) (acc[i] * scale);
        return 0;
    }

    // independent-sample rendering with the multiplexed estimator at one path depth: n uniform PSS points,
    // box-splatted, scaled to radiance units. strat[s] = mean luminance contributed by strategy s (MIS-weighted).
    int mmltRender(int depth, uint64_t n, uint64_t seedv, int lightImage, int nthreads, float *out, double *strat) override {
        nthreads = std::max(1, nthreads);
        Bidir<F> bd(scene);
        std::vector<std::vector<double>> acc(nthreads, std::vector<double>((size_t) scene.width * scene.height * 3, 0.0));
        std::vector<std::vector<double>> ss(nthreads, std::vector<double>(depth + 2, 0.0));
        auto work = [&](int t) {
            Random rs(seedv, (uint32_t) (3 * t)), re(seedv, (uint32_t) (3 * t + 1)), rd(seedv, (uint32_t) (3 * t + 2));
            ReplayableSampler<F> sensor(&rs), emitter(&re), direct(&rd);
            SplatList<F> l;
            for (uint64_t i = (uint64_t) t; i < n; i += (uint64_t) nthreads) {
                uint32_t major = (uint32_t) (i / (uint64_t) nthreads);
                rs.seek(TAG_PT, major, 0); re.seek(TAG_PT, major, 0); rd.seek(TAG_PT, major, 0);
                int s_, t_;
                bd.sampleSplatsMMLT(emitter, sensor, direct, depth, cfg.max_depth, cfg.direct_samples >= 0, lightImage != 0, l, s_, t_);
                if (l.luminance > 0 && spectrumValid(l.value)) {
                    int x = std::min(std::max((int) std::floor(l.px), 0), scene.width - 1);
                    int y = std::min(std::max((int) std::floor(l.py), 0), scene.height - 1);
                    double *px = &acc[t][((size_t) y * scene.width + x) * 3];
                    px[0] += l.value.x; px[1] += l.value.y; px[2] += l.value.z;
                    ss[t][s_] += l.luminance;
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        double scale = (double) scene.width * scene.height / (double) n;
        for (size_t i = 0; i < acc[0].size(); ++i) {
            double v = 0;
            for (int t = 0; t < nthreads; ++t) v += acc[t][i];
            out[i] = (float) (v * scale);
        }
        if (strat) for (int s_ = 0; s_ <= depth + 1; ++s_) {
            double v = 0;
            for (int t = 0; t < nthreads; ++t) v += ss[t][s_];
            strat[s_] = v / (double) n;
        }
        return 0;
    }

    // independent-sample rendering with the bidirectional estimator (all its splats), radiance units
    int bdptRender(uint64_t n, uint64_t seedv, int nthreads, float *out) override {
        nthreads = std::max(1, nthreads);
        std::vector<std::unique_ptr<Film<F>>> films;
        for (int t = 0; t < nthreads; ++t) films.emplace_back(new Film<F>(scene.width, scene.height, scene.filterType, scene.filterParam));
        auto work = [&](int t) {
            Random rs(seedv, (uint32_t) (2 * t)), re(seedv, (uint32_t) (2 * t + 1)), rd(seedv, 0x40000000u + (uint32_t) t);
            ReplayableSampler<F> sensor(&rs), emitter(&re), direct(&rd);
            SplatList<F> l;
            Bidir<F> bd(scene);
            for (uint64_t i = (uint64_t) t; i < n; i += (uint64_t) nthreads) {
                uint32_t major = (uint32_t) (i / (uint64_t) nthreads);
                rs.seek(TAG_PT, major, 0); re.seek(TAG_PT, major, 0); rd.seek(TAG_PT, major, 0);
                bd.sampleSplatsBDPT(emitter, sensor, cfg.max_depth, cfg.rr_depth, cfg.direct_samples >= 0, cfg.no_light_image == 0, l, c.directSampling ? &direct : nullptr);
                if (l.hasMain && spectrumValid(l.value)) films[t]->put(l.px, l.py, l.value);
                for (const auto &sp : l.more) if (spectrumValid(sp.value)) films[t]->put(sp.px, sp.py, sp.value);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        std::vector<double> acc((size_t) scene.width * scene.height * 3, 0.0);
        for (auto &f : films) f->accumulateInto(acc);
        double scale = (double) scene.width * scene.height / (double) n;
        for (size_t i = 0; i < acc.size(); ++i) out[i] = (float) (acc[i] * scale);
        return 0;
    }
    // out row: [lum, hasMain, px, py, r, g, b, nMore, nDims, nRays, then nMore x (px, py, r, g, b)] (stride floats)
    // uDirect: n x dim components of the direct sampler (directSampling = true), or NULL
    int bdptEval(const float *uSensor, const float *uEmitter, const float *uDirect, uint32_t n, uint32_t dim, float *out, uint32_t stride) override {
        Bidir<F> bd(scene);
        if (c.directSampling && !uDirect) throw std::runtime_error("bdpt_eval: directSampling=true needs the direct sampler's components");
        for (uint32_t i = 0; i < n; ++i) {
            ArraySampler<F> sensor(uSensor + (size_t) i * dim, dim), emitter(uEmitter + (size_t) i * dim, dim);
            ArraySampler<F> direct(uDirect ? uDirect + (size_t) i * dim : uSensor, dim);
            SplatList<F> l;
            bd.sampleSplatsBDPT(emitter, sensor, cfg.max_depth, cfg.rr_depth, cfg.direct_samples >= 0, cfg.no_light_image == 0, l, c.directSampling ? &direct : nullptr);
            float *o = out + (size_t) i * stride;
            for (uint32_t k = 0; k < stride; ++k) o[k] = 0.f;
            o[0] = (float) l.luminance; o[1] = l.hasMain ? 1.f : 0.f; o[2] = (float) l.px; o[3] = (float) l.py;
            o[4] = (float) l.value.x; o[5] = (float) l.value.y; o[6] = (float) l.value.z;
            o[7] = (float) l.more.size(); o[8] = (float) (sensor.sampleIndex + emitter.sampleIndex + (c.directSampling ? direct.sampleIndex : 0)); o[9] = (float) l.nRays;
            for (size_t k = 0; k < l.more.size() && 10 + 5 * (k + 1) <= stride; ++k) {
                float *q = o + 10 + 5 * k;
                q[0] = (float) l.more[k].px; q[1] = (float) l.more[k].py;
                q[2] = (float) l.more[k].value.x; q[3] = (float) l.more[k].value.y; q[4] = (float) l.more[k].value.z;
            }
        }
        return 0;
    }

    int mmltEval(int depth, int lightImage, const float *uSensor, const float *uEmitter, const float *uDirect, uint32_t n,
                 uint32_t dim, drmlt_splat *out, int *st) override {
        Bidir<F> bd(scene);
        for (uint32_t i = 0; i < n; ++i) {
            ArraySampler<F> sensor(uSensor + (size_t) i * dim, dim), emitter(uEmitter + (size_t) i * dim, dim), direct(uDirect + i, 1);
            SplatList<F> l;
            int s_, t_;
            bd.sampleSplatsMMLT(emitter, sensor, direct, depth, cfg.max_depth, cfg.direct_samples >= 0, lightImage != 0, l, s_, t_);
            out[i].luminance = (float) l.luminance;
            out[i].x = (float) l.px; out[i].y = (float) l.py;
            out[i].rgb[0] = (float) l.value.x; out[i].rgb[1] = (float) l.value.y; out[i].rgb[2] = (float) l.value.z;
            out[i].n_dims = (int) (sensor.sampleIndex + emitter.sampleIndex + direct.sampleIndex); out[i].n_rays = l.nRays;
            if (st) { st[2 * i] = s_; st[2 * i + 1] = t_; }
        }
        return 0;
    }
};

} // namespace

namespace {
template <typename F> std::unique_ptr<TransitionKernel<F>> makeKernel(int kind, double p0, double p1) {
    switch (kind) {
        case 0: return std::make_unique<GaussianKernel<F>>((F) p0);
        case 1: return std::make_unique<KelemenKernel<F>>((F) p0, (F) p1);
        case 2: return std::make_unique<IdentityKernel<F>>();
        default: return std::make_unique<WrappedCauchyKernel<F>>((F) p0);
    }
}
} // namespace

extern "C" {

void *oracle_create(const drmlt_config *cfg, const drmlt_scene *scene, int precision, char *err, size_t errlen) {
    std::string e;
    CtxBase *ctx = nullptr;
    try {
        if (precision == 32) { auto *c = new Ctx<float>(); e = c->init(*cfg, *scene); ctx = c; }
        else { auto *c = new Ctx<double>(); e = c->init(*cfg, *scene); ctx = c; }
    } catch (const std::exception &ex) { e = ex.what(); }
    if (!e.empty()) {
        if (err && errlen) std::snprintf(err, errlen, "%s", e.c_str());
        delete ctx;
        return nullptr;
    }
    return ctx;
}
void oracle_destroy(void *p) { delete static_cast<CtxBase *>(p); }
const char *oracle_last_error(void *p) { return static_cast<CtxBase *>(p)->error.c_str(); }

#define GUARD(expr) try { return (expr); } catch (const std::exception &ex) { static_cast<CtxBase *>(p)->error = ex.what(); return DRMLT_E_INVALID; }

int oracle_eval_paths(void *p, const float *u, uint32_t n, uint32_t dim, drmlt_splat *out) { GUARD(static_cast<CtxBase *>(p)->evalPaths(u, n, dim, out)) }
int oracle_seed(void *p, uint64_t seed, uint32_t chain_offset, double *b) { GUARD(static_cast<CtxBase *>(p)->seed(seed, chain_offset, b)) }
int oracle_seed_indices(void *p, uint64_t seed, uint32_t chain_offset, uint32_t pool_chains, const uint32_t *indices, double *b) { GUARD(static_cast<CtxBase *>(p)->seed(seed, chain_offset, b, pool_chains, indices)) }
// the bootstrap sample indices the last seed call gave this context's chains (work_units values): mirrors drmlt_seed_indices
int oracle_picked_seeds(void *p, uint32_t *out) {
    CtxBase *c = static_cast<CtxBase *>(p);
    if (!c || !out || c->pickedSeeds.empty()) return DRMLT_E_STATE;
    std::memcpy(out, c->pickedSeeds.data(), c->pickedSeeds.size() * sizeof(uint32_t));
    return 0;
}
int oracle_seed_pool(void *p, uint64_t seed, uint32_t first_chain, uint32_t pool_chains, double *b) { GUARD(static_cast<CtxBase *>(p)->seed(seed, first_chain, b, pool_chains)) }
int oracle_run(void *p, uint64_t total, int nthreads) { GUARD(static_cast<CtxBase *>(p)->run(total, nthreads)) }
int oracle_film_read(void *p, float *out) { GUARD(static_cast<CtxBase *>(p)->filmRead(out)) }
int oracle_develop(void *p, const float *direct, float *out) { GUARD(static_cast<CtxBase *>(p)->developImage(direct, out)) }
int oracle_stats_get(void *p, drmlt_stats *out) { GUARD(static_cast<CtxBase *>(p)->stats(out)) }
int oracle_chain_state(void *p, drmlt_splat *cur, float *u, uint32_t dim) { GUARD(static_cast<CtxBase *>(p)->chainState(cur, u, dim)) }
int oracle_render_pt(void *p, uint32_t spp, uint64_t seed, int nthreads, float *out) { GUARD(static_cast<CtxBase *>(p)->renderPT(spp, seed, nthreads, out)) }
int oracle_bootstrap_lum(void *p, uint64_t seed, uint32_t stream, uint32_t n, float *out) { GUARD(static_cast<CtxBase *>(p)->bootstrapLum(seed, stream, n, out)) }
int oracle_mmlt_render(void *p, int depth, uint64_t n, uint64_t seed, int lightImage, int nthreads, float *out, double *strat) { GUARD(static_cast<CtxBase *>(p)->mmltRender(depth, n, seed, lightImage, nthreads, out, strat)) }
int oracle_mmlt_eval(void *p, int depth, int lightImage, const float *uSensor, const float *uEmitter, const float *uDirect, uint32_t n, uint32_t dim, drmlt_splat *out, int *st) { GUARD(static_cast<CtxBase *>(p)->mmltEval(depth, lightImage, uSensor, uEmitter, uDirect, n, dim, out, st)) }

// ---- unit-level entry points ------------------------------------------------

// The selection half of PathSampler::generateSeeds (pathsampler.cpp:936-957) on caller-supplied luminance samples: the
// bootstrap test hands in the DEVICE's luminances, so that the picks themselves (CDF, lower_bound, zero-mass skipping,
// the TAG_SEEDSEL stream) are compared deterministically. out: nSeeds sample indices, sorted.
void oracle_select_seeds(const float *lum, uint32_t n, uint64_t seed, uint32_t stream, uint32_t nSeeds, uint32_t *out) {
    std::vector<PathSeed> temp, seeds;
    for (uint32_t i = 0; i < n; ++i)
        if (!std::isnan(lum[i]) && lum[i] != 0) temp.push_back(PathSeed{i, (double) lum[i], -1});
    Random boot(seed, stream);
    selectSeeds<double>(temp, boot, nSeeds, seeds);
    std::sort(seeds.begin(), seeds.end(), [](const PathSeed &a, const PathSeed &b) { return a.sampleIndex < b.sampleIndex; });
    for (uint32_t j = 0; j < nSeeds; ++j) out[j] = seeds[j].sampleIndex;
}

void oracle_philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t *out) {
    auto r = Philox::block(k0, k1, c0, c1, c2, c3);
    for (int i = 0; i < 4; ++i) out[i] = r[i];
}

// uniforms of the addressed stream: out[i] = U(seed, chain, tag, major, idx0 + i)
void oracle_uniforms(uint64_t seed, uint32_t chain, uint32_t tag, uint32_t major, uint32_t idx0, uint32_t n, float *out) {
    Random r(seed, chain);
    r.seek(tag, major, idx0);
    for (uint32_t i = 0; i < n; ++i) out[i] = r.nextFloat();
}

// kind: 0 gaussian(sigma) 1 kelemen(s1,s2) 2 identity 3 wrappedcauchy(rho). Samples n values drawing
// from the addressed stream (seed, chain=0, TAG_S1, major=0) sequentially; precision 32|64.
void oracle_kernel_sample(int kind, double p0, double p1, int precision, uint64_t seed, uint32_t n, double *out) {
    Random r(seed, 0);
    r.seek(TAG_S1, 0, 0);
    if (precision == 32) { auto k = makeKernel<float>(kind, p0, p1); for (uint32_t i = 0; i < n; ++i) out[i] = k->sample(&r); }
    else { auto k = makeKernel<double>(kind, p0, p1); for (uint32_t i = 0; i < n; ++i) out[i] = k->sample(&r); }
}
void oracle_kernel_pdf(int kind, double p0, double p1, int precision, uint32_t n, const double *du, double *pdf, double *logpdf) {
    if (precision == 32) { auto k = makeKernel<float>(kind, p0, p1); for (uint32_t i = 0; i < n; ++i) { pdf[i] = k->pdf((float) du[i]); logpdf[i] = k->logPdf((float) du[i]); } }
    else { auto k = makeKernel<double>(kind, p0, p1); for (uint32_t i = 0; i < n; ++i) { pdf[i] = k->pdf(du[i]); logpdf[i] = k->logPdf(du[i]); } }
}

// Sampler trace: one DRMLTSampler of `dim` dims with current state x; returns the wrapped
// first-stage proposal y, the second-stage proposal z, Green's y*, and Mira's kernel ratio.
// `used` = number of dims consumed by each stage evaluation.
int oracle_sampler_trace(int type, double sigma, double scaleSecond, int precision, uint64_t seed, uint32_t chain,
                         uint32_t mutation, int largeStep, uint32_t dim, uint32_t used, const double *x, double *y,
                         double *z, double *ystar, double *ratio, double *xAcc1, double *xAcc2) {
    auto body = [&](auto tag) {
        using F = decltype(tag);
        Random r(seed, chain);
        DRMLTSampler<F> s((DRType) type, (F) sigma, (F) scaleSecond, &r);
        s.setMaxDim(dim);
        for (uint32_t k = 0; k < dim; ++k) s.uCurrent.push_back((F) x[k]);
        s.setMutation(mutation);
        s.setLargeStep(largeStep != 0);
        for (uint32_t k = 0; k < used; ++k) y[k] = s.next1D();
        if (!largeStep) {
            s.nextStage();
            for (uint32_t k = 0; k < used; ++k) z[k] = s.next1D();
            if (type == EGreen) {
                s.setReverse(true);
                for (uint32_t k = 0; k < used; ++k) ystar[k] = s.next1D();
                s.setReverse(false);
            }
            *ratio = s.getTransitionRatio();
            std::vector<F> first = s.uFirst, second = s.uSecond;
            for (uint32_t k = 0; k < dim; ++k) { xAcc1[k] = DRMLTSampler<F>::wrap(first[k]); xAcc2[k] = DRMLTSampler<F>::wrap(second[k]); }
        } else {
            std::vector<F> first = s.uFirst;
            for (uint32_t k = 0; k < dim; ++k) xAcc1[k] = DRMLTSampler<F>::wrap(first[k]);
        }
    };
    try {
        if (precision == 32) body(float(0)); else body(double(0));
    } catch (const std::exception &) { return -1; }
    return 0;
}

// Toy-target chains (analytic 2-D density) through the SAME acceptance code:
// returns the W x H histogram of expectation-weighted splats (sum over chains).
int oracle_toy_run(int type, int useMixture, int timidAfterLarge, double pLarge, double sigma, double scaleSecond,
                   uint64_t seed, uint32_t nChains, uint64_t nMutations, int w, int h, double *hist, drmlt_stats *stats) {
    using F = double;
    Config<F> c{};
    c.algo = 0; c.type = type; c.maxDepth = 1; c.rrDepth = 1; c.separateDirect = false; c.acceptanceMap = false;
    c.timidAfterLarge = timidAfterLarge != 0; c.useMixture = useMixture != 0; c.kelemenWeights = false; c.kelemenMutation = true;
    c.pLarge = pLarge; c.sigma = sigma; c.scaleSecond = scaleSecond; c.maxDim = 2;
    ToyEvaluator<F> eval{w, h};
    Film<F> film(w, h, DRMLT_FILTER_BOX, 0.5);
    Stats st;
    try {
        Random boot(seed, 0);
        std::vector<PathSeed> seeds;
        generateSeeds<F>(eval, boot, 4096, nChains, seeds);
        for (uint32_t i = 0; i < nChains; ++i) {
            DRChain<F, ToyEvaluator<F>> chain(c, eval, seed, i, 0);
            if (!chain.init(seeds[i])) return DRMLT_E_REPLAY;
            chain.run(nMutations, film, st);
        }
    } catch (const std::exception &) { return -1; }
    std::vector<double> acc((size_t) w * h * 3, 0.0);
    film.accumulateInto(acc);
    for (size_t i = 0; i < (size_t) w * h; ++i) hist[i] = acc[i * 3];
    if (stats) { std::memset(stats, 0, sizeof(*stats)); std::memcpy(&stats->first_acc, &st.first_acc, sizeof(uint64_t) * 18); }
    return 0;
}
double oracle_toy_target(double x, double y) { return ToyEvaluator<double>::target(x, y); }

// Acceptance-map run of the delayed-rejection loop on the toy target (drmlt_proc.cpp:443-450,693-709): returns the
// W x H x 3 map and every chain's INITIAL state x0 (nChains x 2), so that a test can replay the chains step by step in
// its own words (tests/test_oracle_process.py: the reference's swap-then-mark order of operations, literally).
int oracle_toy_amap(int type, double pLarge, double sigma, double scaleSecond, uint64_t seed, uint32_t nChains, uint64_t nMutations,
                    int w, int h, double *rgb, double *x0) {
    using F = double;
    Config<F> c{};
    c.algo = 0; c.type = type; c.maxDepth = 1; c.rrDepth = 1; c.separateDirect = false; c.acceptanceMap = true;
    c.timidAfterLarge = false; c.useMixture = false; c.kelemenWeights = false; c.kelemenMutation = true;
    c.pLarge = pLarge; c.sigma = sigma; c.scaleSecond = scaleSecond; c.maxDim = 2;
    ToyEvaluator<F> eval{w, h};
    Film<F> film(w, h, DRMLT_FILTER_BOX, 0.5);
    Stats st;
    try {
        Random boot(seed, 0);
        std::vector<PathSeed> seeds;
        generateSeeds<F>(eval, boot, 4096, nChains, seeds);
        for (uint32_t i = 0; i < nChains; ++i) {
            DRChain<F, ToyEvaluator<F>> chain(c, eval, seed, i, 0);
            if (!chain.init(seeds[i])) return DRMLT_E_REPLAY;
            const std::vector<F> x = chain.state();
            x0[2 * i] = x[0]; x0[2 * i + 1] = x[1];
            chain.run(nMutations, film, st);
        }
    } catch (const std::exception &) { return -1; }
    std::vector<double> acc((size_t) w * h * 3, 0.0);
    film.accumulateInto(acc);
    for (size_t i = 0; i < acc.size(); ++i) rgb[i] = acc[i];
    return 0;
}

// ImageBlock::put through the discretised filter: splat n samples, return interior W x H x 3
int oracle_film_put(int w, int h, int filter, double param, uint32_t n, const float *xy, const float *rgb, float *out) {
    Film<float> film(w, h, filter, (float) param);
    for (uint32_t i = 0; i < n; ++i) film.put(xy[2 * i], xy[2 * i + 1], V3<float>(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]));
    std::vector<double> acc((size_t) w * h * 3, 0.0);
    film.accumulateInto(acc);
    for (size_t i = 0; i < acc.size(); ++i) out[i] = (float) acc[i];
    return 0;
}

// rough conductor in local coordinates (double): eval/pdf of (wi, wo) pairs and samples for (sx, sy) pairs
void oracle_roughconductor(int ggx, double alpha, const double *eta, const double *k, const double *wi, uint32_t n, const double *wo_in,
                           const double *sxy, double *eval_out, double *pdf_out, double *wo_out, double *weight_out, double *spdf_out) {
    RoughConductor<double> rc{Microfacet<double>(ggx != 0, alpha), V3<double>(eta[0], eta[1], eta[2]), V3<double>(k[0], k[1], k[2]), V3<double>(1, 1, 1)};
    V3<double> w(wi[0], wi[1], wi[2]);
    for (uint32_t i = 0; i < n; ++i) {
        if (wo_in) {
            V3<double> wo(wo_in[3 * i], wo_in[3 * i + 1], wo_in[3 * i + 2]);
            V3<double> e = rc.eval(w, wo);
            eval_out[3 * i] = e.x; eval_out[3 * i + 1] = e.y; eval_out[3 * i + 2] = e.z;
            pdf_out[i] = rc.pdf(w, wo);
        }
        if (sxy) {
            V3<double> wo;
            double pdf = 0;
            V3<double> wt = rc.sample(w, sxy[2 * i], sxy[2 * i + 1], wo, pdf);
            wo_out[3 * i] = wo.x; wo_out[3 * i + 1] = wo.y; wo_out[3 * i + 2] = wo.z;
            weight_out[3 * i] = wt.x; weight_out[3 * i + 1] = wt.y; weight_out[3 * i + 2] = wt.z;
            spdf_out[i] = wt.isZero() ? 0.0 : pdf;
        }
    }
}

int oracle_bdpt_render(void *p, uint64_t n, uint64_t seed, int nthreads, float *out) { GUARD(static_cast<CtxBase *>(p)->bdptRender(n, seed, nthreads, out)) }
int oracle_bdpt_eval(void *p, const float *uSensor, const float *uEmitter, const float *uDirect, uint32_t n, uint32_t dim, float *out, uint32_t stride) { GUARD(static_cast<CtxBase *>(p)->bdptEval(uSensor, uEmitter, uDirect, n, dim, out, stride)) }
int oracle_set_importance_map(void *p, const float *map) { GUARD(static_cast<CtxBase *>(p)->setImportance(map)) }
void oracle_luminance_map(const float *rgb, int w, int h, int W, int H, float *out) { luminanceMap(rgb, w, h, W, H, out); }
int oracle_find_max_dim(int maxDepth, int rrDepth) { return findMaxDimensionsPath(maxDepth, rrDepth); }
int oracle_find_max_dim_mmlt(int depth) { return findMaxDimensionsMMLT(depth); }

} // extern "C"
