// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
//
// Random source, Markov transition kernels and the primary-sample-space
// samplers of the reference:
//   tools/transition.h:54-190        Gaussian / Kelemen / Identity / WrappedCauchy
//   drmlt_sampler.h:37-298, .cpp     DRMLTSampler + Green / Mira / Orbital
//   pssmlt_sampler.h:113-143, .cpp   PSSMLTSampler
//
// The reference draws from one sequential SFMT stream seeded from /dev/urandom
// (random.cpp:473-487), i.e. only the U[0,1) distribution is contractual. So
// that GPU and oracle chains can be compared mutation by mutation, the oracle
// draws the SAME uniforms the HIP kernels do: a counter-based Philox4x32-10
// stream addressed by (tag, chain, mutation, linear index). Within one
// (tag, chain, mutation) the draws are sequential exactly as in the reference
// code (fillSpace draws per dimension in order), so the sampler code below reads
// like the reference's.
#pragma once
#include "oracle_math.hpp"
#include <array>
#include <memory>
#include <stdexcept>
#include <utility>
#include <vector>

namespace oracle {

// ---------------------------------------------------------------- Philox4x32-10
// Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11).
struct Philox {
    static inline void mulhilo(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
        uint64_t p = (uint64_t) a * (uint64_t) b;
        hi = (uint32_t) (p >> 32);
        lo = (uint32_t) p;
    }
    static std::array<uint32_t, 4> block(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                                         uint32_t c3) {
        for (int r = 0; r < 10; ++r) {
            uint32_t hi0, lo0, hi1, lo1;
            mulhilo(0xD2511F53u, c0, hi0, lo0);
            mulhilo(0xCD9E8D57u, c2, hi1, lo1);
            uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        return {c0, c1, c2, c3};
    }
};

// Stream tags (counter word 3). Shared numbering with the HIP kernels
// (DESIGN.md "RNG addressing").
enum : uint32_t {
    TAG_BOOT = 0,    // bootstrap sample i: linear index = PSS dimension
    TAG_SEEDSEL = 1, // seed slot j: one draw for the luminance-proportional pick
    TAG_COIN = 2,    // per mutation: [0] large step, [1] accept 1, [2] accept 2, [3] mixture
    TAG_S1 = 3,      // first-stage perturbation draws
    TAG_S2 = 4,      // second-stage perturbation draws
    TAG_PT = 5       // independent path-tracing reference samples
};

// The reference's `Random` (nextFloat), backed by the addressed Philox stream.
class Random {
public:
    Random(uint64_t seed = 0, uint32_t chain = 0) : m_k0((uint32_t) seed), m_k1((uint32_t) (seed >> 32)), m_chain(chain) {}
    void setChain(uint32_t chain) { m_chain = chain; m_have = false; }
    // position the stream: subsequent nextFloat() calls return linear index idx, idx+1, ...
    void seek(uint32_t tag, uint32_t major, uint32_t idx = 0) {
        m_tag = tag; m_major = major; m_idx = idx; m_have = false;
    }
    // 24-bit uniform in [0,1): identical value in float and double builds
    float nextFloat() {
        uint32_t blk = m_idx >> 2;
        if (!m_have || blk != m_blkIdx) {
            m_blk = Philox::block(m_k0, m_k1, blk, m_major, m_chain, m_tag);
            m_blkIdx = blk;
            m_have = true;
        }
        uint32_t w = m_blk[m_idx & 3];
        ++m_idx;
        return (float) (w >> 8) * (1.0f / 16777216.0f);
    }
private:
    uint32_t m_k0, m_k1, m_chain, m_tag = 0, m_major = 0, m_idx = 0, m_blkIdx = 0;
    std::array<uint32_t, 4> m_blk{};
    bool m_have = false;
};

// ---------------------------------------------------------------- kernels
template <typename F> struct TransitionKernel {
    virtual ~TransitionKernel() = default;
    virtual F sample(Random *random) const = 0;
    virtual F pdf(F du) const = 0;
    virtual F logPdf(F du) const = 0;
    virtual bool isIdentity() const { return false; }
};

// transition.h:54-84
template <typename F> struct GaussianKernel : TransitionKernel<F> {
    F sigma, sigma_log;
    explicit GaussianKernel(F s) : sigma(s), sigma_log(std::log(s * s)) {}
    F sample(Random *random) const override {
        // Box-Muller, two uniforms, cosine branch only
        F u1 = random->nextFloat();
        F tmp = std::sqrt(F(-2) * std::log(F(1) - u1));
        F u2 = random->nextFloat();
        return tmp * std::cos(F(2 * kPi) * u2) * sigma;
    }
    F pdf(F du) const override {
        F inv = F(1) / sigma;
        return F(kSqrt1_2Pi) * inv * std::exp(F(-0.5) * du * du * inv * inv);
    }
    F logPdf(F du) const override {
        F r = du / sigma;
        return F(-0.5) * (r * r + std::log(F(2 * kPi)) + sigma_log);
    }
};

// transition.h:90-127: |d| log-uniform on [s1, s2], random sign
template <typename F> struct KelemenKernel : TransitionKernel<F> {
    F s1, s2, logRatio;
    KelemenKernel(F a, F b) : s1(a), s2(b), logRatio(-std::log(b / a)) {}
    F sample(Random *random) const override {
        F xi = random->nextFloat();
        F sign;
        if (xi < F(0.5)) { sign = 1; xi *= 2; }
        else { sign = -1; xi = 2 * (xi - F(0.5)); }
        return sign * s2 * std::exp((1 - xi) * logRatio);
    }
    F pdf(F du) const override {
        F d = std::abs(du);
        if (d < s1 || d > s2) return 0;
        return F(1) / (2 * d * (-logRatio));
    }
    F logPdf(F du) const override { return std::log(pdf(du)); }
};

// transition.h:133-142
template <typename F> struct IdentityKernel : TransitionKernel<F> {
    F sample(Random *) const override { return 0; }
    F pdf(F) const override { return 1; }
    F logPdf(F) const override { return 0; }
    bool isIdentity() const override { return true; }
};

// transition.h:150-190: inverse-CDF sample of a zero-mean wrapped Cauchy
template <typename F> struct WrappedCauchyKernel : TransitionKernel<F> {
    F rho, dispersion;
    explicit WrappedCauchyKernel(F r) : rho(r), dispersion(2 * r / (1 + r * r)) {}
    F sample(Random *random) const override {
        F xi = random->nextFloat();
        F sign = 1;
        if (xi < F(0.5)) { xi *= 2; }
        else { sign = -1; xi = 2 * (xi - F(0.5)); }
        F V = std::cos(F(2 * kPi) * xi);
        return sign * safe_acos((V + dispersion) / (1 + dispersion * V));
    }
    F pdf(F du) const override {
        F r2 = rho * rho;
        return F(0.5 * kInvPi) * (1 - r2) / (1 + r2 - 2 * rho * std::cos(du));
    }
    F logPdf(F du) const override { return std::log(pdf(du)); }
};

// ---------------------------------------------------------------- Sampler base
// include/mitsuba/render/sampler.h:114-117 (only what the path code calls)
template <typename F> struct Sampler {
    virtual ~Sampler() = default;
    virtual F next1D() = 0;
    virtual void next2D(F &a, F &b) {
        a = next1D(); // "Enforce a specific order of evaluation"
        b = next1D();
    }
    size_t sampleIndex = 0;
};

// rsampler.cpp:82-107 -- the bootstrap sampler. setSampleIndex(i) positions the
// stream on bootstrap sample i (the reference rewinds one long stream; here a
// sample's draws are addressed by its index).
template <typename F> struct ReplayableSampler : Sampler<F> {
    Random *random;
    uint32_t current = 0;
    int depth = -1; // technique=mmlt: path depth of the bootstrap sample being drawn (pathsampler.cpp:884-890)
    explicit ReplayableSampler(Random *r) : random(r) {}
    void setSampleIndex(uint32_t i) { current = i; random->seek(TAG_BOOT, i, 0); this->sampleIndex = 0; }
    F next1D() override { ++this->sampleIndex; return (F) random->nextFloat(); }
};

enum DRType { EGreen = 0, EMira = 1, EOrbital = 2 };

// ---------------------------------------------------------------- DRMLTSampler
// drmlt_sampler.h:37-214 / drmlt_sampler.cpp:189-263,313-332,416-425
template <typename F> class DRMLTSampler : public Sampler<F> {
public:
    DRMLTSampler(DRType type, F sigma, F scaleSecond, Random *random)
        : m_random(random), m_type(type), m_sigma(sigma), m_scaleSecond(scaleSecond) {
        configureStages();
    }
    template <typename Cfg>
    DRMLTSampler(const Cfg &cfg, Random *random) : DRMLTSampler((DRType) cfg.type, cfg.sigma, cfg.scaleSecond, random) {}
    void configureForSeed(int /*depth*/) {} // technique=path: dimensions do not depend on the seed

    // drmlt_sampler.cpp:121-164
    void configureStages() {
        if (m_type == EOrbital) {
            F s1 = m_s1 * m_kelemenScale, s2 = m_s2 * m_kelemenScale; // pairwise compensation
            stage1 = std::make_unique<KelemenKernel<F>>(s1, s2);
            stage2 = std::make_unique<WrappedCauchyKernel<F>>(m_rho);
        } else {
            stage1 = std::make_unique<KelemenKernel<F>>(m_s1, m_s2);
            stage2 = std::make_unique<GaussianKernel<F>>(m_scaleSecond * m_sigma);
        }
    }
    // drmlt_sampler.cpp:112-116
    void setStagesToIdentity() {
        stage1 = std::make_unique<IdentityKernel<F>>();
        stage2 = std::make_unique<IdentityKernel<F>>();
        stageLT = std::make_unique<IdentityKernel<F>>();
    }
    // drmlt_sampler.cpp:130-134,148-152,170-177
    void handleLightTracing() {
        configureStages();
        stageLT = std::move(stage2);
        stage2 = std::make_unique<IdentityKernel<F>>();
    }

    void setLargeStep(bool v) { m_largeStep = v; }
    void setReplay(bool v) { m_replay = v; }
    void setMaxDim(size_t d) { m_maxDim = d; }
    void setRandom(Random *r) { m_random = r; }
    void setMutation(uint32_t m) { m_mutation = m; }
    // first draw index of this sampler inside a (tag, mutation) stream: the three samplers of
    // technique=mmlt share one Random (drmlt_proc.cpp:489-492) and get disjoint index ranges here
    void setDrawBase(uint32_t b) { m_drawBase = b; }
    std::vector<F> stateVector() const { return uCurrent; }
    void setReverse(bool v) { this->sampleIndex = 0; isReverse = v; } // Green only
    void nextStage(bool lightTracing = false) {
        this->sampleIndex = 0;
        isFirst = false;
        isLightTracing = lightTracing;
    }

    static F wrap(F y) { return y > 1 ? F(2) - y : (y <= 0 ? std::abs(y) : y); } // drmlt_sampler.h:140-144

    void accept(bool acceptFirst) { // drmlt_sampler.cpp:189-199
        uCurrent = acceptFirst ? uFirst : uSecond;
        for (F &v : uCurrent) v = wrap(v);
        clearProposals();
    }
    void reset() { uCurrent.clear(); clearProposals(); }
    void reject() { clearProposals(); }
    // Generate this stage's proposal NOW -- the eager form of primarySample's k == 0 branch. The reference fills a sampler's
    // proposal when its first component is requested; a sampler no component of which is requested during an evaluation
    // keeps an EMPTY proposal, accept() then empties its current state (drmlt_sampler.cpp:189-191) and the next fillSpace reads
    // out of bounds. Harmless where usage is fixed along a chain (mmlt: strategy kept), fatal for bdpt's direct sampler, whose
    // use depends on the proposed path. Priming makes the proposal what a request for component 0 would have made it (the
    // draws are addressed, so a later request regenerates the same values); the device evaluates proposals as pure functions
    // and commits every dimension on acceptance, which is the same thing.
    void prime() {
        if (m_replay || m_maxDim == 0) return;
        std::vector<F> &uProposed = isFirst ? uFirst : uSecond;
        uProposed.clear();
        fillSpace(isFirst);
    }
    void fillReplay() { // drmlt_sampler.h:127-131
        while (uCurrent.size() < m_maxDim) uCurrent.push_back((F) m_random->nextFloat());
    }

    F next1D() override { return primarySample(this->sampleIndex++); }

    // drmlt_sampler.cpp:231-263 (base) and :271-307 (Green: reverse mode)
    F primarySample(size_t k) {
        std::vector<F> &uProposed = isFirst ? uFirst : uSecond;
        if (isFirst) dimStage1 = std::max(k, dimStage1); else dimStage2 = std::max(k, dimStage2);
        bool greenReverse = (m_type == EGreen) && isReverse;
        if (k == 0 && !greenReverse) uProposed.clear();
        if (m_replay) {
            uProposed.push_back((F) m_random->nextFloat());
            return wrap(uProposed[k]);
        }
        if (greenReverse) { // y* = z - (y - x), DRMLT Sec. 4.4
            F du = uFirst[k] - uCurrent[k];
            return wrap(uSecond[k] - du);
        }
        if (k == 0) fillSpace(isFirst);
        if (k > m_maxDim) throw std::runtime_error("Exceeded maximum dimension");
        return wrap(uProposed.at(k));
    }

    // drmlt_sampler.cpp:400-414 (Mira): prod Q1(z-y) / Q1(x-y) over used dims
    F getTransitionRatio() const {
        if (m_type != EMira) return 1;
        if (stage1->isIdentity()) return 1;
        size_t dimStage = std::max(dimStage1, dimStage2);
        F num = 0, denum = 0;
        for (size_t i = 0; i < dimStage; ++i) {
            num += stage1->logPdf(uSecond[i] - uFirst[i]);
            denum += stage1->logPdf(uCurrent[i] - uFirst[i]);
        }
        return std::exp(num - denum);
    }

    std::vector<F> uCurrent, uFirst, uSecond;
    size_t dimStage1 = 0, dimStage2 = 0;
    std::unique_ptr<TransitionKernel<F>> stage1, stage2, stageLT;
    bool isFirst = true, isReverse = false, isLightTracing = false;

private:
    const TransitionKernel<F> &currentKernel() const {
        if (isFirst) return *stage1;
        return isLightTracing ? *stageLT : *stage2;
    }
    void clearProposals() {
        this->sampleIndex = 0;
        isFirst = true;
        isReverse = false;
        uFirst.clear();
        uSecond.clear();
        dimStage1 = dimStage2 = 0;
    }

    // drmlt_sampler.cpp:313-332 (iid) and :339-394 (orbital, pairwise)
    void fillSpace(bool first) {
        std::vector<F> &uProposed = first ? uFirst : uSecond;
        m_random->seek(first ? TAG_S1 : TAG_S2, m_mutation, m_drawBase);
        const TransitionKernel<F> &kern = currentKernel();
        for (size_t i = 0; i < m_maxDim; ++i) {
            if (m_largeStep) {
                // The reference has SAssert(isFirst) here (drmlt_sampler.cpp:320,346): with
                // timidAfterLarge a rejected large step reaches this branch in its second stage and
                // a build with assertions (MTS_DEBUG, the CMake default) aborts. Restated is what
                // the code does with assertions compiled out: another uniform proposal.
                uProposed.push_back((F) m_random->nextFloat());
            } else if (kern.isIdentity()) {
                uProposed.push_back(uCurrent.at(i));
            } else if (m_type != EOrbital) {
                uProposed.push_back(uCurrent.at(i) + kern.sample(m_random));
            } else if (first) {
                // pair (i, i+1): radius from the (scaled) Kelemen kernel, uniform angle
                F d = kern.sample(m_random);
                F a = (F) m_random->nextFloat() * F(2) * F(kPi);
                uProposed.push_back(uCurrent.at(i) + d * std::cos(a));
                uProposed.push_back(uCurrent.at(i + 1) + d * std::sin(a));
                ++i;
            } else {
                // orbital step: rotate x-y about y in the (i, i+1) plane by theta ~ wrapped Cauchy
                F theta = kern.sample(m_random);
                F du1 = uFirst.at(i) - uCurrent.at(i);
                F du2 = uFirst.at(i + 1) - uCurrent.at(i + 1);
                F norm = std::sqrt(du1 * du1 + du2 * du2);
                F mu = safe_acos(-du1 / norm);
                if (-du2 < 0) mu = F(2 * kPi) - mu;
                uProposed.push_back(uFirst[i] + std::cos(theta + mu) * norm);
                uProposed.push_back(uFirst[i + 1] + std::sin(theta + mu) * norm);
                ++i;
            }
        }
    }

    Random *m_random;
    DRType m_type;
    F m_sigma, m_scaleSecond;
    const F m_s1 = F(1) / F(1024), m_s2 = F(1) / F(64);      // drmlt_sampler.h:201-202
    const F m_rho = std::exp(F(-0.25)), m_kelemenScale = F(1.9); // :204-205
    size_t m_maxDim = 80;
    bool m_largeStep = false, m_replay = false;
    uint32_t m_mutation = 0, m_drawBase = 0;
};

// ---------------------------------------------------------------- PSSMLTSampler
// pssmlt_sampler.cpp:93-168, pssmlt_sampler.h:113-143
template <typename F> class PSSMLTSampler : public Sampler<F> {
public:
    PSSMLTSampler(F s1, F s2, F sigma, Random *random)
        : m_random(random), m_s1(s1), m_s2(s2), m_logRatio(-std::log(s2 / s1)), m_sigma(sigma) {}
    void setLargeStep(bool v) { m_largeStep = v; }
    void setReplay(bool v) { m_replay = v; }
    void setMaxDim(size_t d) { m_maxDim = d; }
    void setMutationType(bool kelemen) { m_useKelemen = kelemen; }
    void setRandom(Random *r) { m_random = r; }
    void setMutation(uint32_t m) { m_mutation = m; }
    void reset() { u.clear(); m_backup.clear(); this->sampleIndex = 0; }
    void accept() { m_backup.clear(); this->sampleIndex = 0; }
    void reject() {
        for (auto &b : m_backup) u[b.first] = b.second;
        m_backup.clear();
        this->sampleIndex = 0;
    }
    F next1D() override { return primarySample(this->sampleIndex++); }

    F primarySample(size_t i) {
        if (m_replay) {
            u.push_back((F) m_random->nextFloat());
            return u[i];
        }
        if (i == 0) {
            m_random->seek(TAG_S1, m_mutation, 0);
            for (size_t k = 0; k < m_maxDim; ++k) {
                if (k == u.size()) {
                    u.push_back((F) m_random->nextFloat());
                } else {
                    m_backup.emplace_back(k, u[k]);
                    u[k] = m_largeStep ? (F) m_random->nextFloat() : mutate(u[k]);
                }
            }
        }
        if (i >= m_maxDim) throw std::runtime_error("PSSMLT sampler out of bounds");
        return u[i];
    }
    std::vector<F> u;

private:
    F mutate(F value) {
        if (m_useKelemen) { // toroidal wrap
            F xi = (F) m_random->nextFloat();
            bool add;
            if (xi < F(0.5)) { add = true; xi *= 2; }
            else { add = false; xi = 2 * (xi - F(0.5)); }
            F dv = m_s2 * std::exp(xi * m_logRatio);
            if (add) { value += dv; if (value > 1) value -= 1; }
            else { value -= dv; if (value < 0) value += 1; }
        } else {
            F u1 = (F) m_random->nextFloat();
            F tmp = std::sqrt(-2 * std::log(1 - u1));
            F u2 = (F) m_random->nextFloat();
            F v = value + m_sigma * tmp * std::cos(F(2 * kPi) * u2);
            value = v - std::floor(v); // math::modulo(v, 1)
        }
        return value;
    }
    Random *m_random;
    F m_s1, m_s2, m_logRatio, m_sigma;
    size_t m_maxDim = 80;
    bool m_largeStep = false, m_replay = false, m_useKelemen = true;
    uint32_t m_mutation = 0;
    std::vector<std::pair<size_t, F>> m_backup;
};

} // namespace oracle
