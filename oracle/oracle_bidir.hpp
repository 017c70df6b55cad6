// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
//
// Multiplexed MLT path evaluation (technique=mmlt): PathSampler::sampleSplats, EMMLT branch
// (src/libbidir/pathsampler.cpp:84-320) with the pieces of libbidir it calls, restated for what
// the configs use: pinhole perspective sensor, area emitters on rectangles / triangles, no media,
// no null BSDFs, sampleDirect = false (forced off for mmlt, drmlt.cpp:229-231).
//   Path::randomWalk              src/libbidir/path.cpp:500-535
//   PathVertex::sampleNext        src/libbidir/vertex.cpp:37-350
//   PathEdge::sampleNext          src/libbidir/edge.cpp:27-84
//   PathVertex::eval / evalPdf    src/libbidir/vertex.cpp:958-1205
//   PathVertex::cast              src/libbidir/vertex.cpp:1384-1434
//   pathConnectAndCollapse        src/libbidir/edge.cpp:558-690
//   PathEdge::evalCached          src/libbidir/edge.cpp:221-271
//   Path::miWeight                src/libbidir/path.cpp:763-1028
//   perspective sensor            src/sensors/perspective.cpp:191-245,300-385
//   area emitter                  src/emitters/area.cpp:96-189; scene.cpp:1066-1087
#pragma once
#include "oracle_process.hpp"

namespace oracle {

enum { EImportance = 0, ERadiance = 1 };                                      // ETransportMode
enum { MInvalid = 0, MSolidAngle = 1, MLength = 2, MArea = 3, MDiscrete = 4 }; // EMeasure
enum { VEmitterSupernode, VSensorSupernode, VEmitterSample, VSensorSample, VSurface };

template <typename F> struct PVertex {
    int type = VSurface;
    V3<F> p, n;          // position; pRec.n (emitter / sensor sample) or geometric normal (surface)
    Frame<F> shFrame;    // surface interactions
    int shape = -1, bsdf = -1, emitter = -1;
    V3<F> weight[2];
    F pdf[2] = {0, 0};
    F rrWeight = 1; // russian roulette compensation of the random walk (vertex.cpp:46,314-321)
    int measure = MInvalid;
    bool degenerate = false;
    bool isConnectable() const { return !degenerate && measure != MDiscrete; }
    bool isOnSurface() const { return type == VSurface || type == VEmitterSample || type == VSensorSample; }
    bool isSupernode() const { return type == VEmitterSupernode || type == VSensorSupernode; }
};
template <typename F> struct PEdge {
    F length = 0;
    V3<F> d; // along the light path (emitter -> sensor)
};
template <typename F> struct SubPath {
    std::vector<PVertex<F>> v;
    std::vector<PEdge<F>> e;
};

template <typename F> class Bidir {
public:
    const Scene<F> &scene;
    F normalization; // 1 / area of the image rectangle at z = 1 (perspective.cpp:171-175)
    explicit Bidir(const Scene<F> &s) : scene(s) {
        F w = 2 * s.tanHalfFov, h = 2 * s.tanHalfFov / s.aspect;
        normalization = F(1) / (w * h);
    }

    // ---- sensor -----------------------------------------------------------------------------
    V3<F> camPos() const { return V3<F>(scene.camToWorld[3], scene.camToWorld[7], scene.camToWorld[11]); }
    V3<F> camDir() const { return V3<F>(scene.camToWorld[2], scene.camToWorld[6], scene.camToWorld[10]); }
    V3<F> toCamera(const V3<F> &d) const { // inverse rotation (no scale allowed in toWorld)
        const F *m = scene.camToWorld;
        return V3<F>(m[0] * d.x + m[4] * d.y + m[8] * d.z, m[1] * d.x + m[5] * d.y + m[9] * d.z, m[2] * d.x + m[6] * d.y + m[10] * d.z);
    }
    V3<F> toWorld(const V3<F> &d) const {
        const F *m = scene.camToWorld;
        return V3<F>(m[0] * d.x + m[1] * d.y + m[2] * d.z, m[4] * d.x + m[5] * d.y + m[6] * d.z, m[8] * d.x + m[9] * d.y + m[10] * d.z);
    }
    F importance(const V3<F> &dLocal) const { // perspective.cpp:191-245
        F cosTheta = dLocal.z;
        if (cosTheta <= 0) return 0;
        F inv = 1 / cosTheta;
        F px = dLocal.x * inv, py = dLocal.y * inv;
        F hx = scene.tanHalfFov, hy = scene.tanHalfFov / scene.aspect;
        if (px < -hx || px > hx || py < -hy || py > hy) return 0;
        return normalization * inv * inv * inv;
    }
    bool samplePositionOf(const V3<F> &dWorld, F &sx, F &sy) const { // getSamplePosition, :355-372
        V3<F> l = toCamera(dWorld);
        if (l.z <= 0) return false;
        F u = F(0.5) - F(0.5) * (l.x / l.z) / scene.tanHalfFov;
        F v = F(0.5) - F(0.5) * (l.y / l.z) * scene.aspect / scene.tanHalfFov;
        if (u < 0 || u > 1 || v < 0 || v > 1) return false;
        sx = u * scene.width; sy = v * scene.height;
        return true;
    }

    // ---- vertex sampling (PathVertex::sampleNext + PathEdge::sampleNext) ---------------------
    bool sampleNext(PVertex<F> &cur, const PVertex<F> *pred, const PEdge<F> *predEdge, PEdge<F> &succEdge, PVertex<F> &succ,
                    int mode, Sampler<F> &sampler, uint64_t *rays, bool russianRoulette = false, V3<F> *throughput = nullptr) const {
        succ = PVertex<F>();
        succEdge = PEdge<F>();
        cur.rrWeight = 1;
        Ray<F> ray;
        switch (cur.type) {
            case VEmitterSupernode: {
                F sx, sy;
                sampler.next2D(sx, sy);
                F emPdf;
                size_t index = scene.sampleEmitterIndex(sx, emPdf);
                const Emitter<F> &em = scene.emitters[index];
                const Shape<F> &sh = scene.shapes[em.shape];
                F pdfPos;
                scene.samplePosition(sh, sx, sy, succ.p, succ.n, pdfPos);
                V3<F> power = em.radiance * (F(kPi) / sh.invArea); // m_power = radiance * pi * area
                cur.weight[EImportance] = power / emPdf;
                cur.pdf[EImportance] = pdfPos * emPdf;
                cur.measure = MArea;
                succ.type = VEmitterSample;
                succ.emitter = (int) index; succ.shape = em.shape;
                succ.degenerate = false;
                return true;
            }
            case VSensorSupernode: {
                F sx, sy;
                sampler.next2D(sx, sy); // consumed, unused by a pinhole (samplePosition, :300-308)
                succ.p = camPos(); succ.n = camDir();
                cur.weight[ERadiance] = V3<F>(1);
                cur.pdf[ERadiance] = 1;
                cur.measure = MDiscrete;
                succ.type = VSensorSample;
                succ.degenerate = false;
                return true;
            }
            case VEmitterSample: {
                F sx, sy;
                sampler.next2D(sx, sy);
                V3<F> local = squareToCosineHemisphere(sx, sy);
                V3<F> d = Frame<F>(cur.n).toWorld(local);
                F pdfDir = squareToCosineHemispherePdf(local);
                cur.weight[EImportance] = V3<F>(1);
                cur.weight[ERadiance] = V3<F>(pdfDir / absDot(d, cur.n));
                cur.pdf[EImportance] = pdfDir;
                cur.pdf[ERadiance] = 1;
                cur.measure = MSolidAngle;
                ray = Ray<F>{cur.p, d, Consts<F>::Epsilon, std::numeric_limits<F>::infinity()};
                break;
            }
            case VSensorSample: {
                F sx, sy;
                sampler.next2D(sx, sy); // needsDirectionSample(): the film position
                V3<F> nearP((1 - 2 * sx) * scene.tanHalfFov * scene.nearClip,
                            (1 - 2 * sy) * scene.tanHalfFov / scene.aspect * scene.nearClip, scene.nearClip);
                V3<F> dl = normalize(nearP);
                V3<F> d = toWorld(dl);
                F pdfDir = normalization / (dl.z * dl.z * dl.z);
                cur.weight[EImportance] = V3<F>(pdfDir / absDot(d, cur.n));
                cur.weight[ERadiance] = V3<F>(1);
                cur.pdf[EImportance] = 1;
                cur.pdf[ERadiance] = pdfDir;
                cur.measure = MSolidAngle;
                ray = Ray<F>{cur.p, d, Consts<F>::Epsilon, std::numeric_limits<F>::infinity()};
                break;
            }
            default: { // surface interaction
                const Bsdf<F> &bsdf = scene.bsdfs[cur.bsdf];
                V3<F> wiW = normalize(pred->p - cur.p);
                V3<F> wi = cur.shFrame.toLocal(wiW);
                F sx, sy;
                sampler.next2D(sx, sy);
                V3<F> woL;
                F pdf = 0, eta;
                bool delta;
                V3<F> w = scene.bsdfSample(bsdf, wi, sx, sy, woL, pdf, eta, delta, mode);
                if (w.isZero()) return false;

                cur.weight[mode] = w;
                cur.pdf[mode] = pdf;
                cur.measure = delta ? MDiscrete : MSolidAngle;
                V3<F> wo = cur.shFrame.toWorld(woL);
                F wiDotGeoN = dot(cur.n, wiW), woDotGeoN = dot(cur.n, wo);
                if (wiDotGeoN * wi.z <= 0 || woDotGeoN * woL.z <= 0) return false;
                // reverse quantities
                F pdfRev = delta ? scene.bsdfPdfDelta(bsdf, woL, wi) : scene.bsdfPdf(bsdf, woL, wi);
                cur.pdf[1 - mode] = pdfRev;
                if (pdfRev <= (sizeof(F) == 4 ? F(2.93873587705571876e-39) : F(5.56268464626800345e-309))) return false; // RCPOVERFLOW
                if (bsdf.type != DRMLT_BSDF_DIELECTRIC) { // symmetric BSDFs
                    cur.weight[1 - mode] = w * (pdf / pdfRev);
                    if (!delta) cur.weight[1 - mode] *= std::abs(woL.z / wi.z);
                } else { // ENonSymmetric: eval in the reverse direction / pdf
                    cur.weight[1 - mode] = scene.bsdfEvalDelta(bsdf, woL, wi, 1 - mode) / pdfRev;
                }
                // "For BDPT & russian roulette, track radiance * eta^2" (vertex.cpp:263-265)
                if (throughput && mode == ERadiance && eta != 1) *throughput *= eta * eta;
                // adjoint BSDF for shading normals (adjointComp = true); 1 up to rounding for flat shading
                if (mode == EImportance) cur.weight[EImportance] *= std::abs((wi.z * woDotGeoN) / (woL.z * wiDotGeoN));
                else cur.weight[EImportance] *= std::abs((woL.z * wiDotGeoN) / (wi.z * woDotGeoN));
                ray = Ray<F>{cur.p, wo, Consts<F>::Epsilon, std::numeric_limits<F>::infinity()};
                break;
            }
        }
        if (throughput) { // vertex.cpp:310-324: the random walk's russian roulette
            *throughput *= cur.weight[mode];
            if (russianRoulette) {
                F q = std::min(throughput->max(), F(0.95));
                if (sampler.next1D() > q) { cur.measure = MInvalid; return false; }
                cur.rrWeight = F(1) / q;
                *throughput *= cur.rrWeight;
            }
        }
        // PathEdge::sampleNext: next surface along the ray
        Intersection<F> its;
        if (!scene.rayIntersect(ray, its, rays)) { cur.measure = MInvalid; return false; }
        const Shape<F> &hs = scene.shapes[its.shape];
        const Bsdf<F> &hb = scene.bsdfs[hs.bsdf];
        succ.type = VSurface;
        succ.p = its.p; succ.n = its.geoFrame.n; succ.shFrame = its.shFrame;
        succ.shape = its.shape; succ.bsdf = hs.bsdf; succ.emitter = hs.emitter;
        succ.degenerate = !(hb.smooth() || hs.emitter >= 0);
        succEdge.length = its.t;
        if (succEdge.length == 0) { cur.measure = MInvalid; return false; }
        succEdge.d = mode == ERadiance ? -ray.d : ray.d;
        // solid angle -> area measure
        if (cur.measure == MSolidAngle) {
            cur.measure = MArea;
            cur.pdf[mode] /= succEdge.length * succEdge.length;
            cur.pdf[mode] *= absDot(ray.d, succ.n);
            if (predEdge && predEdge->length != 0) {
                cur.pdf[1 - mode] /= predEdge->length * predEdge->length;
                if (pred->isOnSurface()) cur.pdf[1 - mode] *= absDot(predEdge->d, pred->n);
            }
        }
        return true;
    }

    int randomWalk(SubPath<F> &path, Sampler<F> &sampler, int nSteps, int mode, uint64_t *rays, int rrStart = -1) const {
        V3<F> throughput(1);
        for (int i = 0; i < nSteps; ++i) {
            size_t last = path.v.size() - 1;
            PVertex<F> succ;
            PEdge<F> succEdge;
            const PVertex<F> *pred = last >= 1 ? &path.v[last - 1] : nullptr;
            const PEdge<F> *predEdge = path.e.empty() ? nullptr : &path.e.back();
            PVertex<F> cur = path.v[last];
            if (!sampleNext(cur, pred, predEdge, succEdge, succ, mode, sampler, rays, rrStart != -1 && i >= rrStart, &throughput)) return i;
            path.v[last] = cur;
            path.v.push_back(succ);
            path.e.push_back(succEdge);
        }
        return nSteps;
    }

    // ---- PathVertex::eval ---------------------------------------------------------------------
    V3<F> eval(const PVertex<F> &v, const PVertex<F> *pred, const PVertex<F> *succ, int mode) const {
        switch (v.type) {
            case VEmitterSupernode:
                if (mode != EImportance || pred || succ->type != VEmitterSample) return V3<F>(0);
                return scene.emitters[succ->emitter].radiance * F(kPi); // evalPosition
            case VSensorSupernode:
                return V3<F>(0); // evalPosition under the area measure: a pinhole has none
            case VEmitterSample: {
                V3<F> target;
                if (mode == EImportance && pred->type == VEmitterSupernode) target = succ->p;
                else if (mode == ERadiance && succ->type == VEmitterSupernode) target = pred->p;
                else return V3<F>(0);
                V3<F> wo = normalize(target - v.p);
                F dp = dot(wo, v.n);
                F r = dp < 0 ? F(0) : F(kInvPi) * dp; // evalDirection
                F adp = std::abs(dp);
                if (adp != 0) r /= adp;
                return V3<F>(r);
            }
            case VSensorSample: {
                V3<F> target;
                if (mode == ERadiance && pred->type == VSensorSupernode) target = succ->p;
                else if (mode == EImportance && succ->type == VSensorSupernode) target = pred->p;
                else return V3<F>(0);
                V3<F> wo = normalize(target - v.p);
                F r = importance(toCamera(wo));
                F dp = absDot(v.n, wo);
                if (dp != 0) r /= dp;
                return V3<F>(r);
            }
            default: {
                const Bsdf<F> &bsdf = scene.bsdfs[v.bsdf];
                V3<F> wiW = normalize(pred->p - v.p), woW = normalize(succ->p - v.p);
                V3<F> wi = v.shFrame.toLocal(wiW), wo = v.shFrame.toLocal(woW);
                V3<F> r = scene.bsdfEval(bsdf, wi, wo); // f * |cos wo|; symmetric for the smooth BSDFs here
                F wiDotGeoN = dot(v.n, wiW), woDotGeoN = dot(v.n, woW);
                if (wiDotGeoN * wi.z <= 0 || woDotGeoN * wo.z <= 0) return V3<F>(0);
                if (mode == EImportance) r *= std::abs((wi.z * woDotGeoN) / (wo.z * wiDotGeoN));
                if (wo.z != 0) r /= std::abs(wo.z);
                return r;
            }
        }
    }

    // ---- PathVertex::evalPdf (area measure) -----------------------------------------------------
    F evalPdf(const PVertex<F> &v, const PVertex<F> *pred, const PVertex<F> *succ, int mode) const {
        F result = 0, dist = 0;
        V3<F> wo;
        switch (v.type) {
            case VEmitterSupernode: {
                if (mode != EImportance || pred || succ->type != VEmitterSample) return 0;
                const Emitter<F> &em = scene.emitters[succ->emitter];
                return scene.shapes[em.shape].invArea * (scene.emitterCdf[succ->emitter + 1] - scene.emitterCdf[succ->emitter]);
            }
            case VSensorSupernode:
                return 0; // pdfSensorPosition under the area measure
            case VEmitterSample: {
                if (mode == ERadiance && succ->type == VEmitterSupernode) return 1;
                if (mode != EImportance || pred->type != VEmitterSupernode) return 0;
                wo = succ->p - v.p; dist = wo.length(); wo /= dist;
                F dp = dot(wo, v.n);
                result = dp < 0 ? F(0) : F(kInvPi) * dp;
                break;
            }
            case VSensorSample: {
                if (mode == EImportance && succ->type == VSensorSupernode) return 1;
                if (mode != ERadiance || pred->type != VSensorSupernode) return 0;
                wo = succ->p - v.p; dist = wo.length(); wo /= dist;
                result = importance(toCamera(wo));
                break;
            }
            default: {
                const Bsdf<F> &bsdf = scene.bsdfs[v.bsdf];
                wo = succ->p - v.p; dist = wo.length(); wo /= dist;
                V3<F> wiW = normalize(pred->p - v.p);
                V3<F> wi = v.shFrame.toLocal(wiW), woL = v.shFrame.toLocal(wo);
                result = scene.bsdfPdf(bsdf, wi, woL);
                F wiDotGeoN = dot(v.n, wiW), woDotGeoN = dot(v.n, wo);
                if (wiDotGeoN * wi.z <= 0 || woDotGeoN * woL.z <= 0) return 0;
                break;
            }
        }
        result /= dist * dist;
        if (succ->isOnSurface()) result *= absDot(wo, succ->n);
        return result;
    }

    // PathVertex::evalPdfDirect(sample, EImportance, EArea), vertex.cpp:1355-1382: density with which direct (emitter)
    // sampling from `ref` produces the emitter sample, in area measure
    F evalPdfDirectEmitter(const PVertex<F> &ref, const PVertex<F> &sample) const {
        typename Scene<F>::DirectSample dRec;
        dRec.ref = ref.p;
        dRec.refN = V3<F>(0);
        if (ref.type == VSurface && !scene.bsdfs[ref.bsdf].transmissiveOrBackside()) dRec.refN = ref.shFrame.n; // records.inl:160-164
        dRec.p = sample.p; dRec.n = sample.n; dRec.emitter = sample.emitter;
        dRec.d = sample.p - ref.p;
        dRec.dist = dRec.d.length();
        dRec.d /= dRec.dist;
        return scene.pdfEmitterDirectArea(dRec);
    }

    // ---- Path::miWeight (no null interactions; sampleDirect: the s = 1 / t = 1 strategies use direct sampling) ----------
    F miWeight(const SubPath<F> &em, const SubPath<F> &se, int s, int t, bool lightImage, bool sampleDirect = false) const {
        const int k = s + t + 1, n = k + 1;
        if (k <= 3) sampleDirect = false; // path.cpp:799-800
        const PVertex<F> *vsPred = s >= 1 ? &em.v[s - 1] : nullptr, *vtPred = t >= 1 ? &se.v[t - 1] : nullptr;
        const PVertex<F> &vs = em.v[s], &vt = se.v[t];
        std::vector<F> pdfImp(n), pdfRad(n);
        std::vector<char> connectable(n);
        int pos = 0;
        for (int i = 0; i <= s; ++i) connectable[pos++] = em.v[i].isConnectable();
        for (int i = t; i >= 0; --i) connectable[pos++] = se.v[i].isConnectable();
        pos = 0;
        pdfImp[pos++] = 1;
        for (int i = 0; i < s; ++i) pdfImp[pos++] = em.v[i].pdf[EImportance];
        pdfImp[pos++] = evalPdf(vs, vsPred, &vt, EImportance);
        if (t > 0) {
            pdfImp[pos++] = evalPdf(vt, &vs, vtPred, EImportance);
            for (int i = t - 1; i > 0; --i) pdfImp[pos++] = se.v[i].pdf[EImportance];
        }
        pos = 0;
        if (s > 0) {
            for (int i = 0; i < s - 1; ++i) pdfRad[pos++] = em.v[i + 1].pdf[ERadiance];
            pdfRad[pos++] = evalPdf(vs, &vt, vsPred, ERadiance);
        }
        pdfRad[pos++] = evalPdf(vt, vtPred, &vs, ERadiance);
        for (int i = t; i > 0; --i) pdfRad[pos++] = se.v[i - 1].pdf[ERadiance];
        pdfRad[pos++] = 1;
        // densities next to specular chains: area -> projected solid angle (the geometric terms cancel)
        auto vertexAt = [&](int i) -> const PVertex<F> & { return i <= s ? em.v[i] : se.v[k - i]; };
        for (int i = 1; i <= k - 3; ++i) {
            if (i == s || !(connectable[i] && !connectable[i + 1])) continue;
            const PVertex<F> &cur = vertexAt(i), &succ = vertexAt(i + 1);
            const PEdge<F> &edge = i < s ? em.e[i] : se.e[k - i - 1];
            pdfImp[i + 1] *= edge.length * edge.length /
                             std::abs((succ.isOnSurface() ? dot(edge.d, succ.n) : F(1)) * (cur.isOnSurface() ? dot(edge.d, cur.n) : F(1)));
        }
        for (int i = k - 1; i >= 3; --i) {
            if (i - 1 == s || !(connectable[i] && !connectable[i - 1])) continue;
            const PVertex<F> &cur = vertexAt(i), &succ = vertexAt(i - 1);
            const PEdge<F> &edge = i <= s ? em.e[i - 1] : se.e[k - i];
            pdfRad[i - 1] *= edge.length * edge.length /
                             std::abs((succ.isOnSurface() ? dot(edge.d, succ.n) : F(1)) * (cur.isOnSurface() ? dot(edge.d, cur.n) : F(1)));
        }
        // Direct sampling strategies (path.cpp:803-824,936-965). Area emitters (direct measure: solid angle) and a pinhole
        // (direct measure: discrete) leave `connectable` as it is: [0] and [1] stay true, [k - 1] stays true, [k] stays false.
        // No null interactions: the reference vertices are 2 and k - 2. The sensor's ratio is pdfSensorDirect (1 under the
        // discrete measure, perspective.cpp:386-390) / pdfRad[k - 1] (the pinhole's position density, 1).
        F ratioEmitterDirect = 0, ratioSensorDirect = 0;
        double initial = 1;
        const int sensorRef = k - 2;
        if (sampleDirect) {
            const PVertex<F> &sample = s > 0 ? em.v[1] : vt;
            const PVertex<F> &ref = 2 <= s ? em.v[2] : se.v[k - 2];
            if (connectable[1] && connectable[2]) ratioEmitterDirect = evalPdfDirectEmitter(ref, sample) / pdfImp[1];
            if (connectable[k - 1] && connectable[sensorRef]) ratioSensorDirect = F(1) / pdfRad[k - 1];
            if (s == 1) initial /= ratioEmitterDirect;
            else if (t == 1) initial /= ratioSensorDirect;
        }
        double weight = 1, pdf = initial;
        for (int i = s + 1; i < k; ++i) {
            double next = pdf * (double) pdfImp[i] / (double) pdfRad[i], value = next;
            if (sampleDirect) {
                if (i == 1) value *= ratioEmitterDirect;
                else if (i == sensorRef) value *= ratioSensorDirect;
            }
            int tPrime = k - i - 1;
            if (connectable[i] && connectable[i + 1] && (lightImage || tPrime > 1)) weight += value * value;
            pdf = next;
        }
        pdf = initial;
        for (int i = s - 1; i >= 0; --i) {
            double next = pdf * (double) pdfRad[i + 1] / (double) pdfImp[i + 1], value = next;
            if (sampleDirect) {
                if (i == 1) value *= ratioEmitterDirect;
                else if (i == sensorRef) value *= ratioSensorDirect;
            }
            int tPrime = k - i - 1;
            if (connectable[i] && connectable[i + 1] && (lightImage || tPrime > 1)) weight += value * value;
            pdf = next;
        }
        return (F) (1.0 / weight);
    }

    // ---- PathSampler::sampleSplats, EMMLT branch -----------------------------------------------------
    // emitter / sensor / direct samplers as in the reference; `depth` fixed per chain. Sets list.s / list.t.
    void sampleSplatsMMLT(Sampler<F> &emitterSampler, Sampler<F> &sensorSampler, Sampler<F> &directSampler, int depth,
                          int maxDepth, bool excludeDirect, bool lightImage, SplatList<F> &list, int &sOut, int &tOut) const {
        list.px = list.py = 0; list.value = V3<F>(0); list.luminance = 0; list.nDims = 0; list.nRays = 0;
        uint64_t rays = 0;
        int s, t, nStrats;
        F decision = directSampler.next1D();
        if (lightImage) { nStrats = depth + 1; s = std::min(int(nStrats * decision), nStrats - 1); t = nStrats - s; }
        else { nStrats = depth; s = std::min(int(nStrats * decision), nStrats - 1); t = 1 + (nStrats - s); }
        sOut = s; tOut = t;
        list.s = s; list.t = t;
        if (depth == 1) return;
        SubPath<F> em, se;
        em.v.emplace_back(); em.v[0].type = VEmitterSupernode; em.v[0].degenerate = false;      // area emitters
        se.v.emplace_back(); se.v[0].type = VSensorSupernode; se.v[0].degenerate = true;        // pinhole
        int tSampled = randomWalk(se, sensorSampler, t, ERadiance, &rays);
        int sSampled = randomWalk(em, emitterSampler, s, EImportance, &rays);
        list.nRays = (int) rays;
        if (tSampled != t || sSampled != s) return;
        bool unconnectable = true;
        for (size_t i = 2; i < em.v.size(); ++i) unconnectable &= !em.v[i].isConnectable();
        for (size_t i = 2; i < se.v.size(); ++i) unconnectable &= !se.v[i].isConnectable();
        if (unconnectable) return;
        V3<F> weight(1);
        for (size_t i = 1; i < em.v.size(); ++i) weight *= em.v[i - 1].weight[EImportance];
        for (size_t i = 1; i < se.v.size(); ++i) weight *= se.v[i - 1].weight[ERadiance];
        PVertex<F> &vs = em.v[s], &vt = se.v[t];
        const PVertex<F> *vsPred = s >= 1 ? &em.v[s - 1] : nullptr, *vtPred = t >= 1 ? &se.v[t - 1] : nullptr;
        V3<F> value;
        F sx = 0, sy = 0;
        F geo = 1;
        if (vs.type == VEmitterSupernode) { // pure sensor path: vt must lie on an emitter
            if (vt.type != VSurface || vt.emitter < 0) return;
            vt.type = VEmitterSample; // cast(): pRec from the intersection, shading normal, area measure
            vt.n = vt.shFrame.n; vt.measure = MArea; vt.degenerate = false;
            value = weight * eval(vs, vsPred, &vt, EImportance) * eval(vt, vtPred, &vs, ERadiance);
        } else {
            if (vs.degenerate || vt.degenerate) return;
            value = weight * eval(vs, vsPred, &vt, EImportance) * eval(vt, vtPred, &vs, ERadiance);
            vs.measure = vt.measure = MArea;
            if (value.isZero()) return;
            // pathConnectAndCollapse: mutual visibility
            V3<F> d = vs.p - vt.p;
            F length = d.length();
            if (length == 0) return;
            d /= length;
            Ray<F> ray{vt.p, d, vt.isOnSurface() ? Consts<F>::Epsilon : F(0), length * (vs.isOnSurface() ? (1 - Consts<F>::ShadowEpsilon) : F(1))};
            Intersection<F> its;
            ++rays;
            list.nRays = (int) rays;
            if (scene.rayIntersect(ray, its)) return;
            // generalised geometric term (edge.cpp:221-271 with both cosine flags)
            geo = F(1) / (length * length);
            if (vs.isOnSurface() && vs.isConnectable()) geo *= absDot(vsShadingNormal(vs), d);
            if (vt.isOnSurface() && vt.isConnectable()) geo *= absDot(vsShadingNormal(vt), d);
        }
        if (value.isZero()) return;
        if (excludeDirect && depth <= 2) return;
        value *= geo;
        // connection edge for the MI weight: supernode connections have length 0
        value *= miWeight(em, se, s, t, lightImage);
        value *= F(nStrats);
        if (t >= 2) {
            samplePositionOf(se.v[2].p - se.v[1].p, sx, sy); // result unchecked (pathsampler.cpp:305-308)
        } else {
            if (!samplePositionOf(vs.p - vt.p, sx, sy)) return; // light tracing: vs seen from the pinhole
        }
        list.px = sx; list.py = sy;
        list.value = value;
        list.luminance = oracle::luminance(value);
    }

    // ---- PathSampler::sampleSplats, EBidirectional branch (pathsampler.cpp:321-527). directSampler != nullptr:
    // directSampling = true, the s = 1 / t = 1 strategies of :424-452 (two components of the direct sampler each).
    void sampleSplatsBDPT(Sampler<F> &emitterSampler, Sampler<F> &sensorSampler, int maxDepth, int rrDepth, bool excludeDirect,
                          bool lightImage, SplatList<F> &list, Sampler<F> *directSampler = nullptr) const {
        const bool sampleDirect = directSampler != nullptr;
        list.px = list.py = 0; list.value = V3<F>(0); list.luminance = 0; list.nDims = 0; list.nRays = 0;
        list.more.clear(); list.hasMain = false; list.s = list.t = 0;
        uint64_t rays = 0;
        SubPath<F> em, se;
        em.v.emplace_back(); em.v[0].type = VEmitterSupernode; em.v[0].degenerate = false;
        se.v.emplace_back(); se.v[0].type = VSensorSupernode; se.v[0].degenerate = true;
        // m_emitterDepth = maxDepth (pinhole: degenerate sensor), m_sensorDepth = maxDepth + 1 (area emitters), :53-71
        randomWalk(em, emitterSampler, maxDepth, EImportance, &rays, rrDepth);
        randomWalk(se, sensorSampler, maxDepth + 1, ERadiance, &rays, rrDepth);
        std::vector<V3<F>> impW(em.v.size()), radW(se.v.size());
        impW[0] = radW[0] = V3<F>(1);
        for (size_t i = 1; i < em.v.size(); ++i) impW[i] = impW[i - 1] * em.v[i - 1].weight[EImportance] * em.v[i - 1].rrWeight;
        for (size_t i = 1; i < se.v.size(); ++i) radW[i] = radW[i - 1] * se.v[i - 1].weight[ERadiance] * se.v[i - 1].rrWeight;
        if (se.v.size() > 2) {
            F sx = 0, sy = 0;
            samplePositionOf(se.v[2].p - se.v[1].p, sx, sy);
            list.hasMain = true; list.px = sx; list.py = sy; list.value = V3<F>(0);
        }
        for (int s = (int) em.v.size() - 1; s >= 0; --s) {
            int minT = std::max(2 - s, lightImage ? 0 : 2), maxT = (int) se.v.size() - 1;
            if (maxDepth != -1) maxT = std::min(maxT, maxDepth + 1 - s);
            for (int t = maxT; t >= minT; --t) {
                PVertex<F> &vs = em.v[s], &vt = se.v[t];
                const PVertex<F> *vsPred = s >= 1 ? &em.v[s - 1] : nullptr, *vtPred = t >= 1 ? &se.v[t - 1] : nullptr;
                struct Restore { PVertex<F> &v; int m; ~Restore() { v.measure = m; } } r0{vs, vs.measure}, r1{vt, vt.measure};
                int depth = s + t - 1;
                V3<F> value;
                F geo = 1;
                F sx = 0, sy = 0;
                if (vs.type == VEmitterSupernode) {
                    if (vt.type != VSurface || vt.emitter < 0) continue;       // cast(EEmitterSample)
                    vt.type = VEmitterSample; vt.n = vt.shFrame.n; vt.measure = MArea; vt.degenerate = false;
                    r1.m = MArea; // the cast is permanent (vertex.cpp:1397-1403)
                    value = radW[t] * eval(vs, vsPred, &vt, EImportance) * eval(vt, vtPred, &vs, ERadiance);
                    if (value.isZero()) continue;
                } else if (vt.type == VSensorSupernode) {
                    continue; // cast(ESensorSample) needs a sensor shape: a pinhole has none
                } else if (sampleDirect && s == 1 && t > 1) {
                    // s = 1: a position on an emitter by direct sampling from vt (vertex.cpp:1285-1346, scene.cpp:879-904
                    // without the visibility test), which replaces the emitter subpath's own first vertex for this strategy
                    if (vt.degenerate) continue;
                    F sx, sy;
                    directSampler->next2D(sx, sy);
                    typename Scene<F>::DirectSample dRec;
                    dRec.ref = vt.p;
                    dRec.refN = (vt.type == VSurface && !scene.bsdfs[vt.bsdf].transmissiveOrBackside()) ? vt.shFrame.n : V3<F>(0);
                    V3<F> direct = scene.sampleEmitterDirect(dRec, sx, sy, nullptr, false);
                    if (direct.isZero()) continue;
                    if ((dRec.ref - dRec.p).lengthSquared() <= 0) continue;
                    const Emitter<F> &emr = scene.emitters[dRec.emitter];
                    SubPath<F> em1; // tempEndpoint, tempEdge, tempSample swapped in for miWeight (pathsampler.cpp:486-503)
                    em1.v.resize(2); em1.e.resize(1);
                    PVertex<F> &tempEndpoint = em1.v[0], &tempSample = em1.v[1];
                    tempEndpoint.type = VEmitterSupernode; tempEndpoint.measure = MArea; tempEndpoint.degenerate = false;
                    tempEndpoint.pdf[EImportance] = scene.shapes[emr.shape].invArea * (scene.emitterCdf[dRec.emitter + 1] - scene.emitterCdf[dRec.emitter]);
                    tempEndpoint.weight[EImportance] = emr.radiance * F(kPi) / tempEndpoint.pdf[EImportance];
                    tempSample.type = VEmitterSample; tempSample.measure = MArea; tempSample.degenerate = false;
                    tempSample.p = dRec.p; tempSample.n = dRec.n; tempSample.emitter = dRec.emitter; tempSample.shape = emr.shape;
                    value = radW[t] * direct * eval(vt, vtPred, &tempSample, ERadiance);
                    vt.measure = MArea;
                    if (value.isZero()) continue;
                    V3<F> d = tempSample.p - vt.p;
                    F length = d.length();
                    if (length == 0) continue;
                    d /= length;
                    Ray<F> ray{vt.p, d, vt.isOnSurface() ? Consts<F>::Epsilon : F(0), length * (1 - Consts<F>::ShadowEpsilon)};
                    Intersection<F> its;
                    ++rays;
                    if (scene.rayIntersect(ray, its)) continue;
                    if (excludeDirect && depth <= 2) continue;
                    if (vt.isOnSurface() && vt.isConnectable()) value *= absDot(vsShadingNormal(vt), d); // ETransmittance | ECosineRad
                    value *= miWeight(em1, se, s, t, lightImage, true);
                    list.value += value; // t >= 2: the sensor-side pixel
                    list.luminance += oracle::luminance(value);
                    continue;
                } else {
                    // t = 1 with direct sampling (s > 1): a pinhole's sampleDirect returns the one point the sensor subpath's
                    // own vertex 1 already is (perspective.cpp:386-420): importance / dist^2 times the cosine at vs equals the
                    // generic connection below term by term -- what remains is that the strategy consumes two components
                    if (sampleDirect && t == 1 && s > 1) {
                        if (vs.degenerate) continue;
                        F sx, sy;
                        directSampler->next2D(sx, sy);
                    }
                    if (vs.degenerate || vt.degenerate) continue;
                    value = impW[s] * radW[t] * eval(vs, vsPred, &vt, EImportance) * eval(vt, vtPred, &vs, ERadiance);
                    vs.measure = vt.measure = MArea;
                    if (value.isZero()) continue;
                    V3<F> d = vs.p - vt.p;
                    F length = d.length();
                    if (length == 0) continue;
                    d /= length;
                    Ray<F> ray{vt.p, d, vt.isOnSurface() ? Consts<F>::Epsilon : F(0), length * (vs.isOnSurface() ? (1 - Consts<F>::ShadowEpsilon) : F(1))};
                    Intersection<F> its;
                    ++rays;
                    if (scene.rayIntersect(ray, its)) continue;
                    geo = F(1) / (length * length);
                    if (vs.isOnSurface() && vs.isConnectable()) geo *= absDot(vsShadingNormal(vs), d);
                    if (vt.isOnSurface() && vt.isConnectable()) geo *= absDot(vsShadingNormal(vt), d);
                }
                if (excludeDirect && depth <= 2) continue;
                value *= geo;
                value *= miWeight(em, se, s, t, lightImage, sampleDirect);
                if (vt.type == VSensorSample && !samplePositionOf(vs.p - vt.p, sx, sy)) continue;
                if (t < 2) { list.more.push_back({sx, sy, value}); }
                else { list.value += value; }
                list.luminance += oracle::luminance(value);
            }
        }
        list.nRays = (int) rays;
    }

private:
    static V3<F> vsShadingNormal(const PVertex<F> &v) { return v.type == VSurface ? v.shFrame.n : v.n; }
};

// pssmlt_utils.h:58-63: sensor = emitter = (depth + 2) * 3 rounded up to even, direct = 1
inline int findMaxDimensionsMMLT(int depth) {
    int maxDim = (depth + 2) * 3;
    if (maxDim % 2 == 1) ++maxDim;
    return maxDim;
}

// Components of the direct sampler under technique=bdpt with directSampling=true. The reference gives it maxDepth
// (pssmlt_utils.h:75) although every s = 1 / t = 1 connection draws two (pathsampler.cpp:424-452, vertex.cpp:1304-1305) and a
// sample makes up to (maxDepth - 1) t = 1 and maxDepth s = 1 connections: its primarySample then reads past the vector
// ("Exceeded maximum dimension", drmlt_sampler.cpp:256-258). Sized here to what the strategies can consume.
inline int findDirectDimensionsBDPT(int maxDepth) { return 2 * (2 * maxDepth - 1); }

// pssmlt_utils.h:69-75: sensor = emitter = (maxDepth + 2) * (2 + RR) rounded up to even; direct = 0 without direct sampling
inline int findMaxDimensionsBDPT(int maxDepth, int rrDepth) {
    int maxDim = (maxDepth + 2) * (2 + (rrDepth < maxDepth ? 1 : 0));
    if (maxDim % 2 == 1) ++maxDim;
    return maxDim;
}

// The sensor / emitter / direct samplers of one chain (drmlt_proc.cpp:84-141). They share the chain's
// Random; here each gets its own draw-index range inside a (tag, mutation) stream:
// sensor [0, 2D), emitter [2D, 4D), direct [4D, ...), D = findMaxDimensionsMMLT(maxDepth) for every chain.
template <typename F> struct MMLTSamplers {
    DRMLTSampler<F> sensor, emitter, direct;
    int depth = -1;
    uint32_t dmax;
    bool bdpt; // technique=bdpt: same triple, dimensions independent of the seed; the direct sampler is an ordinary third
               // sampler when directSampling is on (no identity stages: those are mmlt's, drmlt_proc.cpp:133-141) and unused otherwise
    uint32_t ddirect = 0;
    template <typename Cfg>
    MMLTSamplers(const Cfg &cfg, Random *r)
        : sensor(cfg, r), emitter(cfg, r), direct(cfg, r), bdpt(cfg.technique == DRMLT_TECH_BDPT) {
        dmax = (uint32_t) (bdpt ? findMaxDimensionsBDPT(cfg.maxDepth, cfg.rrDepth) : findMaxDimensionsMMLT(cfg.maxDepth));
        if (bdpt && cfg.directSampling) ddirect = (uint32_t) findDirectDimensionsBDPT(cfg.maxDepth);
        if (!bdpt) {
            direct.setStagesToIdentity();                      // the strategy stays fixed in small steps (:133-135)
            if (cfg.fixEmitterPath) emitter.handleLightTracing(); // :136-140
        }
    }
    void setMaxDim(size_t) {}
    void configureForSeed(int d) { // :452-464
        depth = d;
        size_t D = bdpt ? (size_t) dmax : (size_t) findMaxDimensionsMMLT(d);
        sensor.setMaxDim(D); emitter.setMaxDim(D); direct.setMaxDim(bdpt ? ddirect : 1);
        sensor.setDrawBase(0); emitter.setDrawBase(2 * dmax); direct.setDrawBase(4 * dmax);
    }
    void reset() { emitter.reset(); sensor.reset(); direct.reset(); }
    void setRandom(Random *r) { sensor.setRandom(r); emitter.setRandom(r); direct.setRandom(r); }
    void setReplay(bool v) { sensor.setReplay(v); emitter.setReplay(v); direct.setReplay(v); }
    void setMutation(uint32_t m) { sensor.setMutation(m); emitter.setMutation(m); direct.setMutation(m); }
    void setLargeStep(bool v) { sensor.setLargeStep(v); emitter.setLargeStep(v); direct.setLargeStep(v); if (bdpt && ddirect) direct.prime(); } // called after setMutation: the first-stage proposal is defined
    void setReverse(bool v) { sensor.setReverse(v); emitter.setReverse(v); direct.setReverse(v); }
    void nextStage(bool lightTracing) { sensor.nextStage(); direct.nextStage(); emitter.nextStage(lightTracing); if (bdpt && ddirect) direct.prime(); }
    void accept(bool first) { sensor.accept(first); emitter.accept(first); direct.accept(first); }
    void reject() { sensor.reject(); emitter.reject(); direct.reject(); }
    void fillReplay() { sensor.fillReplay(); emitter.fillReplay(); direct.fillReplay(); } // :506-509
    F getTransitionRatio() const { return sensor.getTransitionRatio() * emitter.getTransitionRatio() * direct.getTransitionRatio(); }
    std::vector<F> stateVector() const { // [sensor | emitter | direct]; an unused emitter state is empty
        std::vector<F> u = sensor.uCurrent;
        u.insert(u.end(), emitter.uCurrent.begin(), emitter.uCurrent.end());
        u.insert(u.end(), direct.uCurrent.begin(), direct.uCurrent.end());
        return u;
    }
};

// Evaluator over a scene: PathSampler::sampleSplats(EMMLT)
template <typename F> struct MMLTEvaluator {
    const Scene<F> *scene;
    int maxDepth;
    bool excludeDirect, lightImage;
    void operator()(MMLTSamplers<F> &set, SplatList<F> &list, Stats *st) const {
        run(set.emitter, set.sensor, set.direct, set.depth, list, st);
    }
    // bootstrap: one replayable stream plays all three roles (drmlt.cpp:514-516, pathsampler.cpp:864)
    void operator()(ReplayableSampler<F> &s, SplatList<F> &list, Stats *st) const { run(s, s, s, s.depth, list, st); }
    void run(Sampler<F> &emitter, Sampler<F> &sensor, Sampler<F> &direct, int depth, SplatList<F> &list, Stats *st) const {
        Bidir<F> bd(*scene);
        int s_, t_;
        const bool shared = &emitter == &sensor;
        auto consumed = [&]() { return shared ? sensor.sampleIndex : emitter.sampleIndex + sensor.sampleIndex + direct.sampleIndex; };
        size_t before = consumed();
        bd.sampleSplatsMMLT(emitter, sensor, direct, depth, maxDepth, excludeDirect, lightImage, list, s_, t_);
        list.nDims = (int) (consumed() - before);
        if (st) { st->path_evals++; st->rays += (uint64_t) list.nRays; }
    }
    int width() const { return scene->width; }
    int height() const { return scene->height; }
};

// Evaluator over a scene: PathSampler::sampleSplats(EBidirectional)
template <typename F> struct BDPTEvaluator {
    const Scene<F> *scene;
    int maxDepth, rrDepth;
    bool excludeDirect, lightImage;
    bool directSampling = false;
    void operator()(MMLTSamplers<F> &set, SplatList<F> &list, Stats *st) const { run(set.emitter, set.sensor, set.direct, list, st); }
    // bootstrap: one replayable stream plays all three roles, in call order (emitter walk, sensor walk, direct draws)
    void operator()(ReplayableSampler<F> &s, SplatList<F> &list, Stats *st) const { run(s, s, s, list, st); }
    void run(Sampler<F> &emitter, Sampler<F> &sensor, Sampler<F> &direct, SplatList<F> &list, Stats *st) const {
        Bidir<F> bd(*scene);
        const bool shared = &emitter == &sensor;
        auto consumed = [&]() { return shared ? sensor.sampleIndex : emitter.sampleIndex + sensor.sampleIndex + (directSampling ? direct.sampleIndex : 0); };
        size_t before = consumed();
        bd.sampleSplatsBDPT(emitter, sensor, maxDepth, rrDepth, excludeDirect, lightImage, list, directSampling ? &direct : nullptr);
        list.nDims = (int) (consumed() - before);
        if (st) { st->path_evals++; st->rays += (uint64_t) list.nRays; }
    }
    int width() const { return scene->width; }
    int height() const { return scene->height; }
};

} // namespace oracle
