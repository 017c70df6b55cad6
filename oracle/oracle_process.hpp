// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
//
// The chain loops and their film:
//   findMaxDimensions         src/integrators/pssmlt_utils.h:27-77
//   ImageBlock::put           include/mitsuba/render/imageblock.h:150-216
//   ReconstructionFilter      src/libcore/rfilter.cpp:37-55, core/rfilter.h:76-77
//   generateSeeds             src/libbidir/pathsampler.cpp:859-960
//   DRMLTRenderer::process    src/integrators/drmlt/drmlt_proc.cpp:386-771
//   processMixture            src/integrators/drmlt/drmlt_proc.cpp:161-380
//   PSSMLTRenderer::process   src/integrators/pssmlt/pssmlt_proc.cpp:113-297
//   DRMLTProcess::develop     src/integrators/drmlt/drmlt_proc.cpp:813-854
// The loops are templated on an Evaluator (sampler -> SplatList) so the same
// acceptance code can be run on an analytic toy target in detailed-balance tests.
#pragma once
#include "oracle_scene.hpp"
#include <cstring>

namespace oracle {

inline int findMaxDimensionsPath(int maxDepth, int rrDepth, bool hasRoughDielectric = false) {
    // EUnidirectional branch: (maxDepth + 2) * (4 + medium + roughDielectric + RR), rounded up to even
    int offsetRR = rrDepth < maxDepth ? 1 : 0;
    int maxDim = (maxDepth + 2) * (4 + 0 + (hasRoughDielectric ? 1 : 0) + offsetRR);
    if (maxDim % 2 == 1) ++maxDim;
    return maxDim;
}

// ---------------------------------------------------------------- film
template <typename F> class Film {
public:
    static constexpr int kRes = 31; // MTS_FILTER_RESOLUTION
    int width, height, border;
    F radius, scaleFactor;
    F values[kRes + 1];
    std::vector<F> data; // (W+2b) x (H+2b) x 3

    Film(int w, int h, int filterType, F param) : width(w), height(h) {
        bool gauss = filterType == DRMLT_FILTER_GAUSSIAN;
        F stddev = param;
        radius = gauss ? 4 * stddev : param + F(1e-5); // gaussian.cpp:33-37, box.cpp:38
        auto eval = [&](F x) -> F {
            if (!gauss) return std::abs(x) <= radius ? F(1) : F(0);
            F alpha = F(-1) / (2 * stddev * stddev);
            return std::max(F(0), std::exp(alpha * x * x) - std::exp(alpha * radius * radius));
        };
        F sum = 0;
        for (int i = 0; i < kRes; ++i) {
            values[i] = eval((radius * i) / kRes);
            sum += values[i];
        }
        values[kRes] = 0;
        scaleFactor = kRes / radius;
        border = (int) std::ceil(radius - F(0.5));
        sum *= 2 * radius / kRes;
        F normalization = F(1) / sum;
        for (int i = 0; i < kRes; ++i) values[i] *= normalization;
        data.assign((size_t) (width + 2 * border) * (height + 2 * border) * 3, F(0));
    }
    F evalDiscretized(F x) const { return values[std::min((int) std::abs(x * scaleFactor), kRes)]; }
    void clear() { std::fill(data.begin(), data.end(), F(0)); }

    bool put(F px, F py, const V3<F> &value) {
        for (int i = 0; i < 3; ++i)
            if (!std::isfinite(value[i]) || value[i] < 0) return false;
        int sx = width + 2 * border, sy = height + 2 * border;
        F posx = px - F(0.5) + border, posy = py - F(0.5) + border;
        int minx = std::max((int) std::ceil(posx - radius), 0), miny = std::max((int) std::ceil(posy - radius), 0);
        int maxx = std::min((int) std::floor(posx + radius), sx - 1), maxy = std::min((int) std::floor(posy + radius), sy - 1);
        for (int y = miny; y <= maxy; ++y) {
            F wy = evalDiscretized(y - posy);
            for (int x = minx; x <= maxx; ++x) {
                F w = evalDiscretized(x - posx) * wy;
                F *dest = &data[((size_t) y * sx + x) * 3];
                dest[0] += w * value.x; dest[1] += w * value.y; dest[2] += w * value.z;
            }
        }
        return true;
    }
    // m_accum->put(result): add the interior of a work-unit block, dropping its border
    void accumulateInto(std::vector<double> &accum) const {
        int sx = width + 2 * border;
        for (int y = 0; y < height; ++y)
            for (int x = 0; x < width; ++x)
                for (int c = 0; c < 3; ++c)
                    accum[((size_t) y * width + x) * 3 + c] += (double) data[((size_t) (y + border) * sx + (x + border)) * 3 + c];
    }
};

struct Stats {
    uint64_t first_acc = 0, first_base = 0, large_acc = 0, large_base = 0, bold_acc = 0, bold_base = 0;
    uint64_t second_acc = 0, second_base = 0, second_large_acc = 0, second_large_base = 0;
    uint64_t second_bold_acc = 0, second_bold_base = 0, overall_acc = 0, overall_base = 0;
    uint64_t mutations = 0, path_evals = 0, rays = 0, accepted = 0;
    void add(const Stats &o) {
        const uint64_t *s = &o.first_acc;
        uint64_t *d = &first_acc;
        for (int i = 0; i < 18; ++i) d[i] += s[i];
    }
};

struct PathSeed {
    uint32_t sampleIndex;
    double luminance;
    int depth = -1; // technique=mmlt: the chain's fixed path depth
    double weight = -1; // what the seed is resampled in proportion to; < 0: the luminance (the reference's rule)
};

template <typename F> struct Config {
    int algo, type, maxDepth, rrDepth;
    bool separateDirect, acceptanceMap, timidAfterLarge, useMixture, kelemenWeights, kelemenMutation;
    F pLarge, sigma, scaleSecond;
    F luminance = 1; // b (pssmlt Kelemen weights)
    int maxDim;
    int technique = 0;                              // DRMLT_TECH_*
    bool fixEmitterPath = false, lightImage = true; // technique=mmlt
    bool directSampling = false;                    // technique=bdpt: the s = 1 / t = 1 strategies of pathsampler.cpp:424-452
    const float *importance = nullptr;              // two-stage MLT luminance image (W x H), drmlt.cpp:406-418
    int impW = 0, impH = 0;
    // Two-stage seeding (drmlt_config.seed_rule). false = DRMLT_SEED_REFERENCE: seeds in proportion to lum(f), what this fork's
    // generateSeeds does (pathsampler.cpp:901-905: the luminance is read BEFORE SplatList::normalize(importanceMap)). true =
    // DRMLT_SEED_TARGET, the product's default: in proportion to lum(f / importance), the chains' own target. Only the resampling
    // weight changes; b stays the mean of lum(f) and the seed keeps lum(f) for the replay check (drmlt_proc.cpp:509-512).
    bool seedByTarget = false;
};

// Evaluator over a scene: PathSampler::sampleSplats(EUnidirectional)
template <typename F> struct SceneEvaluator {
    const Scene<F> *scene;
    int maxDepth, rrDepth;
    bool excludeDirect;
    void operator()(Sampler<F> &sampler, SplatList<F> &list, Stats *st) const {
        sampleSplats(*scene, sampler, maxDepth, rrDepth, excludeDirect, list);
        if (st) { st->path_evals++; st->rays += (uint64_t) list.nRays; }
    }
    int width() const { return scene->width; }
    int height() const { return scene->height; }
};

// Analytic toy target on [0,1]^2 (uses the first two PSS dims): a mixture of two
// anisotropic Gaussian bumps plus a floor; "pixel" = position on a W x H grid.
template <typename F> struct ToyEvaluator {
    int w, h;
    static F target(F x, F y) {
        auto g = [](F x, F y, F cx, F cy, F sx, F sy) {
            F dx = (x - cx) / sx, dy = (y - cy) / sy;
            return std::exp(F(-0.5) * (dx * dx + dy * dy));
        };
        return F(0.05) + g(x, y, F(0.3), F(0.35), F(0.05), F(0.12)) + F(0.6) * g(x, y, F(0.72), F(0.7), F(0.15), F(0.04));
    }
    void operator()(Sampler<F> &sampler, SplatList<F> &list, Stats *st) const {
        F x, y;
        sampler.next2D(x, y);
        F f = target(x, y);
        list.px = x * w; list.py = y * h;
        list.value = V3<F>(f, f, f);
        list.luminance = luminance(list.value);
        list.nDims = 2; list.nRays = 0;
        if (st) st->path_evals++;
    }
    int width() const { return w; }
    int height() const { return h; }
};

// Second half of generateSeeds (pathsampler.cpp:936-954): DiscreteDistribution over the non-zero luminance samples
// (pmf.h:109-121,164-188) and `seedCount` luminance-proportional picks with replacement. Unsorted.
template <typename F>
inline void selectSeeds(const std::vector<PathSeed> &tempSeeds, Random &bootRandom, size_t seedCount, std::vector<PathSeed> &seeds) {
    std::vector<F> cdf(tempSeeds.size() + 1);
    cdf[0] = 0;
    for (size_t i = 0; i < tempSeeds.size(); ++i) cdf[i + 1] = cdf[i] + (F) (tempSeeds[i].weight >= 0 ? tempSeeds[i].weight : tempSeeds[i].luminance);
    F norm = F(1) / cdf.back();
    for (size_t i = 1; i < cdf.size(); ++i) cdf[i] *= norm;
    cdf.back() = 1;
    seeds.clear();
    seeds.reserve(seedCount);
    for (size_t j = 0; j < seedCount; ++j) {
        bootRandom.seek(TAG_SEEDSEL, (uint32_t) j, 0);
        F xi = (F) bootRandom.nextFloat();
        auto entry = std::lower_bound(cdf.begin(), cdf.end(), xi);
        size_t index = (size_t) std::max((ptrdiff_t) 0, (ptrdiff_t) (entry - cdf.begin()) - 1);
        index = std::min(cdf.size() - 2, index);
        while (cdf[index + 1] - cdf[index] == 0 && index < cdf.size() - 1) ++index;
        seeds.push_back(tempSeeds.at(index));
    }
}

// pathsampler.cpp:859-960. Returns b; seeds sorted by sample index.
template <typename F, typename Eval>
inline double generateSeeds(const Eval &eval, Random &bootRandom, size_t sampleCount, size_t seedCount,
                            std::vector<PathSeed> &seeds, std::vector<float> *lumOut = nullptr, int mmltMaxDepth = 0,
                            const float *targetImportance = nullptr, int impW = 0, int impH = 0) {
    ReplayableSampler<F> sampler(&bootRandom);
    std::vector<PathSeed> tempSeeds;
    tempSeeds.reserve(sampleCount);
    SplatList<F> list;
    F mean = 0, variance = 0, tok = 0;
    if (lumOut) lumOut->assign(sampleCount, 0.f);
    for (size_t i = 0; i < sampleCount; ++i) {
        sampler.setSampleIndex((uint32_t) i);
        int depth = mmltMaxDepth > 0 ? (int) (i % (size_t) mmltMaxDepth) + 1 : -1; // :884-890
        sampler.depth = depth;
        eval(sampler, list, nullptr);
        F lum = list.luminance;
        if (lumOut) (*lumOut)[i] = (float) lum;
        if (std::isnan(lum)) continue;
        tok += 1;
        if (lum != 0 && !targetImportance) tempSeeds.push_back(PathSeed{(uint32_t) i, (double) lum, depth});
        if (lum != 0 && targetImportance) { // DRMLT_SEED_TARGET: the sample's luminance under the map (SplatList::normalize, :1001-1020)
            SplatList<F> weighted = list;
            weighted.normalize(targetImportance, impW, impH);
            const F lw = weighted.luminance;
            if (lw > 0 && std::isfinite(lw)) tempSeeds.push_back(PathSeed{(uint32_t) i, (double) lum, depth, (double) lw}); // a sample on a zero of the map seeds nothing
        }
        F delta = lum - mean; // Knuth / Welford
        mean += delta / tok;
        variance += delta * (lum - mean);
    }
    if (mmltMaxDepth > 0) mean *= (F) mmltMaxDepth; // "As we split the path by corresponding depth", :932-934
    if (mean == 0 || tempSeeds.empty()) return 0;
    selectSeeds<F>(tempSeeds, bootRandom, seedCount, seeds);
    std::sort(seeds.begin(), seeds.end(), [](const PathSeed &a, const PathSeed &b) { return a.sampleIndex < b.sampleIndex; });
    return (double) mean;
}

template <typename F> inline bool flipCoin(F x, Random &random, uint32_t mutation, uint32_t slot) {
    if (x >= 1) return true; // no draw, as the reference (:419-422)
    random.seek(TAG_COIN, mutation, slot);
    return (F) random.nextFloat() < x;
}

// One DRMLT Markov chain: drmlt_proc.cpp:386-771 (and :161-380 when useMixture).
// `mutationBase` is the chain-local index of the first mutation (continues across calls).
// `SamplerT` is one DRMLTSampler (technique=path) or the sensor/emitter/direct triple of technique=mmlt.
template <typename F, typename Eval, typename SamplerT = DRMLTSampler<F>> class DRChain {
public:
    DRChain(const Config<F> &cfg, const Eval &eval, uint64_t seed, uint32_t chainId, uint32_t bootStream)
        : m_cfg(cfg), m_eval(eval), m_random(seed, chainId), m_boot(seed, bootStream), m_sampler(cfg, &m_random) {
        m_sampler.setMaxDim((size_t) cfg.maxDim);
    }

    // seed replay, drmlt_proc.cpp:467-514. Returns false on luminance mismatch.
    bool init(const PathSeed &seed) {
        m_sampler.configureForSeed(seed.depth); // per-depth dimensions, :452-464
        m_sampler.reset();
        m_boot.seek(TAG_BOOT, seed.sampleIndex, 0);
        m_sampler.setRandom(&m_boot);
        m_sampler.setReplay(true);
        m_eval(m_sampler, m_current, nullptr);
        m_sampler.setReplay(false);
        m_sampler.accept(true);
        m_sampler.fillReplay(); // tops up from the same addressed stream (dims >= consumed)
        m_sampler.setRandom(&m_random);
        bool ok = std::abs((m_current.luminance - (F) seed.luminance) / (F) seed.luminance) <= Consts<F>::Epsilon;
        m_current.normalize(m_cfg.importance, m_cfg.impW, m_cfg.impH);
        return ok;
    }

    void run(uint64_t nMutations, Film<F> &film, Stats &st) {
        if (m_cfg.useMixture) runMixture(nMutations, film, st);
        else runDR(nMutations, film, st);
    }

    const SplatList<F> &current() const { return m_current; }
    std::vector<F> state() const { return m_sampler.stateVector(); }
    const SamplerT &sampler() const { return m_sampler; }
    uint32_t mutationIndex() const { return m_mutation; }

private:
    static F clamp1(F x) { return std::min(F(1), x); }

    // drmlt_proc.cpp:432-450: every splat of the list
    void splat(Film<F> &film, const SplatList<F> &l, F weight) {
        if (!m_cfg.acceptanceMap && weight > 0) {
            if (l.hasMain) { V3<F> v = l.value * weight; if (spectrumValid(v)) film.put(l.px, l.py, v); }
            for (const auto &sp : l.more) { V3<F> v = sp.value * weight; if (spectrumValid(v)) film.put(sp.px, sp.py, v); }
        }
    }
    void splatAlways(Film<F> &film, const SplatList<F> &l, F weight) {
        if (!(weight > 0)) return;
        if (l.hasMain) { V3<F> v = l.value * weight; if (spectrumValid(v)) film.put(l.px, l.py, v); }
        for (const auto &sp : l.more) { V3<F> v = sp.value * weight; if (spectrumValid(v)) film.put(sp.px, sp.py, v); }
    }
    void splatAcceptance(Film<F> &film, const SplatList<F> &l, int stage) {
        if (!m_cfg.acceptanceMap) return;
        const V3<F> c = stage == 0 ? V3<F>(1, 0, 0) : V3<F>(0, 1, 0);
        if (l.hasMain) film.put(l.px, l.py, c);
        for (const auto &sp : l.more) film.put(sp.px, sp.py, c);
    }

    void runDR(uint64_t nMutations, Film<F> &film, Stats &st) {
        auto isInvalid = [](F x) { return std::isnan(x) || std::isinf(x) || x <= 0; };
        SplatList<F> first, second, reverse;
        for (uint64_t it = 0; it < nMutations; ++it, ++m_mutation) {
            const uint32_t m = m_mutation;
            F a1 = 0, a2 = 0;
            bool acc1 = false, acc2 = false;
            m_sampler.setMutation(m);
            m_random.seek(TAG_COIN, m, 0);
            bool largeStep = (F) m_random.nextFloat() < m_cfg.pLarge;
            m_sampler.setLargeStep(largeStep);

            m_eval(m_sampler, first, &st);
            first.normalize(m_cfg.importance, m_cfg.impW, m_cfg.impH);
            st.mutations++;
            if (!isInvalid(first.luminance)) {
                a1 = clamp1(first.luminance / m_current.luminance);
                acc1 = flipCoin(a1, m_random, m, 1);
            }
            bool doSecond = !acc1;
            if (!m_cfg.timidAfterLarge) doSecond = doSecond && !largeStep;

            if (doSecond) {
                // fixEmitterPath: the emitter sampler moves in the second stage only for pure light tracing (:566-573)
                m_sampler.nextStage(m_cfg.fixEmitterPath && m_current.t == 1);
                m_eval(m_sampler, second, &st);
                second.normalize(m_cfg.importance, m_cfg.impW, m_cfg.impH);
                if (!isInvalid(second.luminance)) {
                    if (m_cfg.type == EGreen) {
                        m_sampler.setReverse(true);
                        m_eval(m_sampler, reverse, &st);
                        reverse.normalize(m_cfg.importance, m_cfg.impW, m_cfg.impH);
                        F aReverse = isInvalid(reverse.luminance) ? F(0) : clamp1(reverse.luminance / second.luminance);
                        if (aReverse != 1) {
                            F lumRatio = second.luminance / m_current.luminance;
                            a2 = clamp1(lumRatio * (1 - aReverse) / (1 - a1));
                            acc2 = flipCoin(a2, m_random, m, 2);
                        }
                        m_sampler.setReverse(false);
                    } else if (m_cfg.type == EMira) {
                        F aReverse = clamp1(first.luminance / second.luminance);
                        if (!(aReverse >= 1)) {
                            F ratio = largeStep ? F(1) : m_sampler.getTransitionRatio();
                            if (!(std::isnan(ratio) || std::isinf(ratio) || ratio <= 0)) {
                                F lumRatio = second.luminance / m_current.luminance;
                                a2 = clamp1(lumRatio * ratio * (1 - aReverse) / (1 - a1));
                                acc2 = flipCoin(a2, m_random, m, 2);
                            }
                        }
                    } else { // orbital, DRMLT Eq. 11
                        if (second.luminance < first.luminance) {
                            a2 = 0;
                        } else if (second.luminance >= m_current.luminance) {
                            a2 = 1; acc2 = true;
                        } else {
                            a2 = (second.luminance - first.luminance) / (m_current.luminance - first.luminance);
                            acc2 = flipCoin(a2, m_random, m, 2);
                        }
                    }
                }
            }

            F w1 = a1, w2 = (1 - a1) * a2, w0 = 1 - w1 - w2;
            splat(film, m_current, w0);
            splat(film, first, w1);
            if (doSecond) splat(film, second, w2); // weight is 0 otherwise

            if (acc1 || acc2) {
                // drmlt_proc.cpp:693-709: `proposed.first.swap(current); splatAcceptanceOnly(proposed.first.get(), 0)` --
                // after the swap `proposed.first` owns the list that WAS current, so the mark lands on every splat position
                // of the state being LEFT, not of the one adopted (same for the second stage with `proposed.second`).
                if (acc1) {
                    std::swap(first, m_current);
                    if (!largeStep) splatAcceptance(film, first, 0);
                } else {
                    std::swap(second, m_current);
                    splatAcceptance(film, second, 1);
                }
                m_sampler.accept(acc1);
                st.accepted++;
                st.overall_base++; st.overall_acc++;
                if (acc1) {
                    st.first_base++; st.first_acc++;
                    if (largeStep) { st.large_base++; st.large_acc++; } else { st.bold_base++; st.bold_acc++; }
                } else {
                    st.overall_base++; st.first_base++;
                    st.second_base++; st.second_acc++;
                    if (largeStep) { st.large_base++; st.second_large_base++; st.second_large_acc++; }
                    else { st.bold_base++; st.second_bold_base++; st.second_bold_acc++; }
                }
            } else {
                m_sampler.reject();
                st.overall_base++; st.first_base++;
                if (largeStep) {
                    st.large_base++;
                    if (doSecond) { st.second_base++; st.second_large_base++; st.overall_base++; }
                } else {
                    st.bold_base++;
                    if (doSecond) { st.second_base++; st.second_bold_base++; st.overall_base++; }
                }
            }
        }
    }

    void runMixture(uint64_t nMutations, Film<F> &film, Stats &st) {
        auto isInvalid = [](F x) { return std::isnan(x) || std::isinf(x) || x < 0; }; // "<", :181
        SplatList<F> proposed;
        for (uint64_t it = 0; it < nMutations; ++it, ++m_mutation) {
            const uint32_t m = m_mutation;
            F a = 0;
            bool accept = false;
            m_sampler.setMutation(m);
            m_random.seek(TAG_COIN, m, 0);
            bool largeStep = (F) m_random.nextFloat() < m_cfg.pLarge;
            m_sampler.setLargeStep(largeStep);
            m_eval(m_sampler, proposed, &st);
            proposed.normalize(m_cfg.importance, m_cfg.impW, m_cfg.impH);
            st.mutations++;
            if (!isInvalid(proposed.luminance)) {
                a = clamp1(proposed.luminance / m_current.luminance);
                accept = flipCoin(a, m_random, m, 1);
            }
            bool doSecond = false;
            if (!largeStep) doSecond = flipCoin(F(0.5), m_random, m, 3);
            if (doSecond) {
                m_sampler.nextStage(m_cfg.fixEmitterPath && m_current.t == 1); // :306-312
                m_eval(m_sampler, proposed, &st);
                proposed.normalize(m_cfg.importance, m_cfg.impW, m_cfg.impH);
                if (isInvalid(proposed.luminance)) { a = 0; accept = false; }
                else {
                    a = clamp1(proposed.luminance / m_current.luminance);
                    accept = flipCoin(a, m_random, m, 2);
                }
            }
            splatAlways(film, m_current, 1 - a); // the mixture loop splats whatever acceptanceMap says (:183-194)
            splatAlways(film, proposed, a);
            st.overall_base++;
            if (!doSecond) { st.first_base++; if (largeStep) st.large_base++; else st.bold_base++; }
            else st.second_base++;
            if (accept) {
                m_current = proposed;
                m_sampler.accept(!doSecond);
                st.accepted++; st.overall_acc++;
                if (!doSecond) { st.first_acc++; if (largeStep) st.large_acc++; else st.bold_acc++; }
                else st.second_acc++;
            } else {
                m_sampler.reject();
            }
        }
    }

    Config<F> m_cfg;
    Eval m_eval;
    Random m_random, m_boot;
    SamplerT m_sampler;
    SplatList<F> m_current;
    uint32_t m_mutation = 0;
};

// One PSSMLT chain: pssmlt_proc.cpp:113-297
template <typename F, typename Eval> class PSSMLTChain {
public:
    PSSMLTChain(const Config<F> &cfg, const Eval &eval, uint64_t seed, uint32_t chainId, uint32_t bootStream)
        : m_cfg(cfg), m_eval(eval), m_random(seed, chainId), m_boot(seed, bootStream),
          m_sampler(F(1) / F(1024), F(1) / F(64), cfg.sigma, &m_random) {
        m_sampler.setMaxDim((size_t) cfg.maxDim);
        m_sampler.setMutationType(cfg.kelemenMutation);
    }
    bool init(const PathSeed &seed) {
        m_sampler.reset();
        m_boot.seek(TAG_BOOT, seed.sampleIndex, 0);
        m_sampler.setRandom(&m_boot);
        m_sampler.setReplay(true);
        m_eval(m_sampler, m_current, nullptr);
        m_sampler.setReplay(false);
        m_sampler.accept();
        m_sampler.setRandom(&m_random);
        bool ok = std::abs((m_current.luminance - (F) seed.luminance) / (F) seed.luminance) <= Consts<F>::Epsilon;
        m_current.normalize(m_cfg.importance, m_cfg.impW, m_cfg.impH);
        return ok;
    }
    void run(uint64_t nMutations, Film<F> &film, Stats &st) {
        SplatList<F> proposed;
        F cumulativeWeight = 0;
        const F b = m_cfg.luminance, pLarge = m_cfg.pLarge;
        for (uint64_t it = 0; it < nMutations; ++it, ++m_mutation) {
            const uint32_t m = m_mutation;
            m_sampler.setMutation(m);
            m_random.seek(TAG_COIN, m, 0);
            bool largeStep = (F) m_random.nextFloat() < pLarge;
            m_sampler.setLargeStep(largeStep);
            m_eval(m_sampler, proposed, &st);
            proposed.normalize(m_cfg.importance, m_cfg.impW, m_cfg.impH);
            st.mutations++;
            F a = std::min(F(1), proposed.luminance / m_current.luminance);
            if (std::isnan(proposed.luminance) || proposed.luminance < 0) a = 0;
            bool accept;
            F currentWeight, proposedWeight;
            if (a > 0) {
                if (m_cfg.kelemenWeights && !m_cfg.importance) { // pssmlt_proc.cpp:203: "Kelemen-style weights don't work for 2-stage MLT"
                    currentWeight = (1 - a) * m_current.luminance / (m_current.luminance / b + pLarge);
                    proposedWeight = (a + (largeStep ? 1 : 0)) * proposed.luminance / (proposed.luminance / b + pLarge);
                } else {
                    currentWeight = 1 - a;
                    proposedWeight = a;
                }
                accept = (a == 1);
                if (!accept) { m_random.seek(TAG_COIN, m, 1); accept = (F) m_random.nextFloat() < a; }
            } else {
                currentWeight = m_cfg.kelemenWeights ? m_current.luminance / (m_current.luminance / b + pLarge) : F(1);
                proposedWeight = 0;
                accept = false;
            }
            cumulativeWeight += currentWeight;
            st.overall_base++;
            if (largeStep) st.large_base++; else st.bold_base++;
            if (accept) {
                V3<F> v = m_current.value * cumulativeWeight;
                if (!v.isZero()) film.put(m_current.px, m_current.py, v);
                cumulativeWeight = proposedWeight;
                m_current = proposed;
                m_sampler.accept();
                st.accepted++; st.overall_acc++;
                if (largeStep) st.large_acc++; else st.bold_acc++;
            } else {
                V3<F> v = proposed.value * proposedWeight;
                if (!v.isZero()) film.put(proposed.px, proposed.py, v);
                m_sampler.reject();
            }
        }
        V3<F> v = m_current.value * cumulativeWeight; // "Perform the last splat"
        if (!v.isZero()) film.put(m_current.px, m_current.py, v);
    }
    const SplatList<F> &current() const { return m_current; }
    const std::vector<F> &state() const { return m_sampler.u; }

private:
    Config<F> m_cfg;
    Eval m_eval;
    Random m_random, m_boot;
    PSSMLTSampler<F> m_sampler;
    SplatList<F> m_current;
    uint32_t m_mutation = 0;
};

// develop(): out = accum * (b / mean_lum(accum)) + direct   (drmlt_proc.cpp:824-849)
inline void develop(const std::vector<double> &accum, int w, int h, double b, bool acceptanceMap, const float *direct,
                    float *out, const float *importance = nullptr) {
    size_t n = (size_t) w * h;
    double avg = 0;
    for (size_t i = 0; i < n; ++i)
        avg += (accum[i * 3] * 0.212671 + accum[i * 3 + 1] * 0.715160 + accum[i * 3 + 2] * 0.072169) *
               (importance ? (double) importance[i] : 1.0);
    avg /= (double) n;
    double factor = acceptanceMap ? 1.0 : b / avg;
    for (size_t i = 0; i < n * 3; ++i)
        out[i] = (float) (accum[i] * factor * (importance ? (double) importance[i / 3] : 1.0) + (direct ? (double) direct[i] : 0.0));
}

// Tail of BidirectionalUtils::mltLuminancePass (src/libbidir/util.cpp:179-196): luminance of the first-stage image,
// up-sampled by Bitmap::resample (bitmap.cpp:2230-2330: X pass, then Y pass) through Resampler
// (core/rfilter.h:123-198,232-290) with the gaussian filter (gaussian.cpp: stddev 0.5, radius 2), EClamp lookups,
// values clamped to [0, inf).
inline void luminanceMap(const float *rgb, int w, int h, int W, int H, float *out) {
    std::vector<double> lum((size_t) w * h);
    for (size_t i = 0; i < lum.size(); ++i)
        lum[i] = (double) rgb[3 * i] * 0.212671 + (double) rgb[3 * i + 1] * 0.715160 + (double) rgb[3 * i + 2] * 0.072169;
    const double stddev = 0.5, fradius = 4 * stddev, alpha = -1.0 / (2 * stddev * stddev), bias = std::exp(alpha * fradius * fradius);
    auto rf = [&](double x) { return std::max(0.0, std::exp(alpha * x * x) - bias); };
    auto resample1d = [&](const std::vector<double> &src, int nSrc, size_t strideSrc, size_t offSrc, std::vector<double> &dst, int nDst,
                          size_t strideDst, size_t offDst) {
        double radius = fradius, invScale = 1;
        if (nDst < nSrc) { double scale = (double) nSrc / nDst; invScale = 1 / scale; radius *= scale; }
        int taps = (int) std::ceil(radius * 2);
        std::vector<double> wts(taps);
        for (int i = 0; i < nDst; ++i) {
            double center = (i + 0.5) / nDst * nSrc;
            int start = (int) std::floor(center - radius + 0.5);
            double sum = 0;
            for (int j = 0; j < taps; ++j) { wts[j] = rf((start + j + 0.5 - center) * invScale); sum += wts[j]; }
            double r = 0;
            for (int j = 0; j < taps; ++j) r += src[offSrc + (size_t) std::min(std::max(start + j, 0), nSrc - 1) * strideSrc] * (wts[j] / sum);
            dst[offDst + (size_t) i * strideDst] = std::max(0.0, r);
        }
    };
    std::vector<double> tmp((size_t) W * h), res((size_t) W * H);
    if (w != W) { for (int y = 0; y < h; ++y) resample1d(lum, w, 1, (size_t) y * w, tmp, W, 1, (size_t) y * W); }
    else tmp = lum;
    if (h != H) { for (int x = 0; x < W; ++x) resample1d(tmp, h, (size_t) W, (size_t) x, res, H, (size_t) W, (size_t) x); }
    else res = tmp;
    for (size_t i = 0; i < res.size(); ++i) out[i] = (float) res[i];
}

} // namespace oracle
