// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
//
// Minimal scene + the reference's unidirectional estimator:
//   shapes      src/shapes/rectangle.cpp:125-168,210-216; src/shapes/sphere.cpp:163-255;
//               include/mitsuba/render/triaccel.h:37-158; skdtree.h:340-429
//   ray query   src/librender/skdtree.cpp:112-142 (closest), :207-226 (shadow)
//   emitters    src/emitters/area.cpp:111-189; src/librender/shape.cpp:102-127;
//               src/librender/scene.cpp:879-904,1057-1060; core/pmf.h:109-188
//   bsdfs       src/bsdfs/diffuse.cpp:110-149; src/bsdfs/dielectric.cpp:228-333
//   sensor      src/sensors/perspective.cpp:126-180,271-300
//   estimator   src/integrators/path/path.cpp:123-321
//   path eval   src/libbidir/pathsampler.cpp:529-567 (EUnidirectional)
// The reference intersects through a SAH kd-tree; for the closest hit the
// result is the same as testing every primitive, which is what the oracle does.
#pragma once
#include "../include/drmlt_abi.h"
#include "oracle_microfacet.hpp"
#include "oracle_sampler.hpp"
#include <string>

namespace oracle {

template <typename F> struct Ray {
    V3<F> o, d;
    F mint, maxt;
};

template <typename F> struct Bsdf {
    int type;
    V3<F> rgb;
    F eta = 1, invEta = 1; // dielectric: intIOR/extIOR
    bool ggx = false;      // rough conductor
    F alpha = F(0.1);
    V3<F> cEta, cK;
    RoughConductor<F> rough() const { return RoughConductor<F>{Microfacet<F>(ggx, alpha), cEta, cK, rgb}; }
    bool smooth() const { return type == DRMLT_BSDF_DIFFUSE || type == DRMLT_BSDF_ROUGHCONDUCTOR; }
    // DirectSamplingRecord(its): refN is zeroed for transmissive / two-sided BSDFs
    bool transmissiveOrBackside() const { return type == DRMLT_BSDF_DIELECTRIC; }
};

template <typename F> struct Shape {
    int type, bsdf, emitter;
    // triangle
    V3<F> p0, p1, p2;
    int k = 3;
    F n_u, n_v, n_d, a_u, a_v, b_nu, b_nv, c_nu, c_nv; // Wald precomputation
    // rectangle: objectToWorld (o2w) / worldToObject (w2o) as 3x4 affine
    F o2w[12], w2o[12];
    V3<F> dpdu, dpdv;
    Frame<F> frame; // geometric frame of flat shapes
    F invArea = 0;
    // sphere
    V3<F> center;
    F radius = 0;
};

template <typename F> struct Emitter {
    int shape;
    V3<F> radiance;
    F weight;
};

template <typename F> struct Intersection {
    bool valid = false;
    F t = std::numeric_limits<F>::infinity();
    int shape = -1;
    V3<F> p, dpdu, wi;
    Frame<F> geoFrame, shFrame;
    V3<F> toLocal(const V3<F> &v) const { return shFrame.toLocal(v); }
    V3<F> toWorld(const V3<F> &v) const { return shFrame.toWorld(v); }
};

template <typename F> inline V3<F> xfPoint(const F *m, const V3<F> &p) {
    return V3<F>(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
                 m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
template <typename F> inline V3<F> xfVector(const F *m, const V3<F> &v) {
    return V3<F>(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
                 m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// inverse of a 3x4 affine (double internally)
inline bool invertAffine(const double *m, double *o) {
    double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (det == 0) return false;
    double id = 1.0 / det;
    o[0] = (e * i - f * h) * id; o[1] = (c * h - b * i) * id; o[2] = (b * f - c * e) * id;
    o[4] = (f * g - d * i) * id; o[5] = (a * i - c * g) * id; o[6] = (c * d - a * f) * id;
    o[8] = (d * h - e * g) * id; o[9] = (b * g - a * h) * id; o[10] = (a * e - b * d) * id;
    for (int r = 0; r < 3; ++r)
        o[r * 4 + 3] = -(o[r * 4 + 0] * m[3] + o[r * 4 + 1] * m[7] + o[r * 4 + 2] * m[11]);
    return true;
}

template <typename F> class Scene {
public:
    std::vector<Shape<F>> shapes;
    std::vector<Bsdf<F>> bsdfs;
    std::vector<Emitter<F>> emitters;
    std::vector<F> emitterCdf; // DiscreteDistribution m_cdf (size n+1)
    V3<F> aabbMin, aabbMax;
    // camera
    F camToWorld[16];
    F tanHalfFov, aspect, nearClip, farClip;
    int width, height;
    int filterType;
    F filterParam;

    std::string load(const drmlt_scene &s) {
        width = s.camera.width; height = s.camera.height;
        if (width <= 0 || height <= 0) return "film size must be positive";
        for (int i = 0; i < 16; ++i) camToWorld[i] = (F) s.camera.to_world[i];
        tanHalfFov = (F) std::tan(0.5 * (double) s.camera.fov_x_deg * kPi / 180.0);
        aspect = (F) width / (F) height;
        nearClip = s.camera.near_clip; farClip = s.camera.far_clip;
        filterType = s.camera.filter; filterParam = s.camera.filter_param;
        for (int i = 0; i < s.n_bsdfs; ++i) {
            const drmlt_bsdf &b = s.bsdfs[i];
            Bsdf<F> o;
            o.type = b.type;
            o.rgb = V3<F>(b.rgb[0], b.rgb[1], b.rgb[2]);
            if (b.type == DRMLT_BSDF_DIELECTRIC) {
                o.eta = (F) b.p[0] / (F) b.p[1];
                o.invEta = 1 / o.eta;
            } else if (b.type == DRMLT_BSDF_ROUGHCONDUCTOR) {
                o.alpha = b.p[0];
                o.cEta = V3<F>(b.p[1], b.p[2], b.p[3]);
                o.cK = V3<F>(b.p[4], b.p[5], b.p[6]);
                o.ggx = b.p[7] != 0.f;
            } else if (b.type != DRMLT_BSDF_DIFFUSE) {
                return "oracle: unsupported bsdf type";
            }
            bsdfs.push_back(o);
        }
        aabbMin = V3<F>(std::numeric_limits<F>::infinity());
        aabbMax = V3<F>(-std::numeric_limits<F>::infinity());
        auto grow = [&](const V3<F> &p) {
            for (int a = 0; a < 3; ++a) { aabbMin[a] = std::min(aabbMin[a], p[a]); aabbMax[a] = std::max(aabbMax[a], p[a]); }
        };
        for (int i = 0; i < s.n_shapes; ++i) {
            const drmlt_shape &in = s.shapes[i];
            Shape<F> sh;
            sh.type = in.type; sh.bsdf = in.bsdf; sh.emitter = in.emitter;
            if (in.bsdf < 0 || in.bsdf >= s.n_bsdfs) return "shape references invalid bsdf";
            if (in.type == DRMLT_SHAPE_TRIANGLE) {
                sh.p0 = V3<F>(in.data[0], in.data[1], in.data[2]);
                sh.p1 = V3<F>(in.data[3], in.data[4], in.data[5]);
                sh.p2 = V3<F>(in.data[6], in.data[7], in.data[8]);
                loadTriAccel(sh);
                V3<F> side1 = sh.p1 - sh.p0, side2 = sh.p2 - sh.p0;
                V3<F> n = cross(side1, side2);
                F len = n.length();
                if (len == 0) return "degenerate triangle";
                sh.invArea = F(1) / (F(0.5) * len);
                n /= len;
                sh.dpdu = side1; sh.dpdv = side2;
                sh.frame = Frame<F>(n);
                grow(sh.p0); grow(sh.p1); grow(sh.p2);
            } else if (in.type == DRMLT_SHAPE_RECTANGLE) {
                double m[12], inv[12];
                for (int k = 0; k < 12; ++k) m[k] = in.data[k];
                if (!invertAffine(m, inv)) return "singular rectangle transform";
                for (int k = 0; k < 12; ++k) { sh.o2w[k] = (F) m[k]; sh.w2o[k] = (F) inv[k]; }
                sh.dpdu = xfVector(sh.o2w, V3<F>(2, 0, 0));
                sh.dpdv = xfVector(sh.o2w, V3<F>(0, 2, 0));
                // normal transforms with the inverse transpose
                V3<F> n(sh.w2o[8], sh.w2o[9], sh.w2o[10]);
                n = normalize(n);
                sh.frame = Frame<F>(normalize(sh.dpdu), normalize(sh.dpdv), n);
                if (std::abs(dot(normalize(sh.dpdu), normalize(sh.dpdv))) > Consts<F>::Epsilon * 100)
                    return "rectangle toWorld contains shear";
                sh.invArea = F(1) / (sh.dpdu.length() * sh.dpdv.length());
                for (int sx = -1; sx <= 1; sx += 2)
                    for (int sy = -1; sy <= 1; sy += 2) grow(xfPoint(sh.o2w, V3<F>((F) sx, (F) sy, 0)));
            } else if (in.type == DRMLT_SHAPE_SPHERE) {
                sh.center = V3<F>(in.data[0], in.data[1], in.data[2]);
                sh.radius = in.data[3];
                sh.invArea = F(1) / (F(4 * kPi) * sh.radius * sh.radius);
                grow(sh.center - V3<F>(sh.radius)); grow(sh.center + V3<F>(sh.radius));
            } else {
                return "unknown shape type";
            }
            shapes.push_back(sh);
        }
        emitterCdf.push_back(0);
        for (int i = 0; i < s.n_emitters; ++i) {
            const drmlt_emitter &e = s.emitters[i];
            if (e.type != DRMLT_EMITTER_AREA) return "unsupported emitter";
            if (e.shape < 0 || e.shape >= s.n_shapes || shapes[e.shape].emitter != i) return "emitter/shape link mismatch";
            emitters.push_back({e.shape, V3<F>(e.radiance[0], e.radiance[1], e.radiance[2]), (F) e.sampling_weight});
            emitterCdf.push_back(emitterCdf.back() + (F) e.sampling_weight);
        }
        if (emitters.empty()) return "scene has no emitters";
        // the kd-tree's box is slightly enlarged after construction (gkdtree.h:1213-1220,
        // MTS_KD_AABB_EPSILON = 1e-3): needed when geometry lies ON the tight box, as walls do
        {
            const F eps = F(1e-3);
            V3<F> ext = aabbMax - aabbMin;
            aabbMin = aabbMin - (ext * eps + V3<F>(eps));
            ext = aabbMax - aabbMin;
            aabbMax = aabbMax + (ext * eps + V3<F>(eps));
        }
        F sum = emitterCdf.back();
        for (size_t i = 1; i < emitterCdf.size(); ++i) emitterCdf[i] *= F(1) / sum; // pmf.h:109-121
        emitterCdf.back() = 1;
        return "";
    }

    // ---- ray queries ---------------------------------------------------
    bool aabbClip(const Ray<F> &ray, F &nearT, F &farT) const { // AABB::rayIntersect
        nearT = -std::numeric_limits<F>::infinity();
        farT = std::numeric_limits<F>::infinity();
        for (int a = 0; a < 3; ++a) {
            F o = ray.o[a], d = ray.d[a];
            if (d == 0) {
                if (o < aabbMin[a] || o > aabbMax[a]) return false;
            } else {
                F t1 = (aabbMin[a] - o) / d, t2 = (aabbMax[a] - o) / d;
                if (t1 > t2) std::swap(t1, t2);
                nearT = std::max(t1, nearT);
                farT = std::min(t2, farT);
                if (!(nearT <= farT)) return false;
            }
        }
        return true;
    }

    bool intersectShape(const Shape<F> &sh, const Ray<F> &ray, F mint, F maxt, F &t, F &u, F &v) const {
        if (sh.type == DRMLT_SHAPE_TRIANGLE) {
            // triaccel.h:93-158 (Wald's projection test)
            static const int mod3[5] = {0, 1, 2, 0, 1};
            if (sh.k > 2) return false;
            int ku = mod3[sh.k + 1], kv = mod3[sh.k + 2];
            F o_u = ray.o[ku], o_v = ray.o[kv], o_k = ray.o[sh.k];
            F d_u = ray.d[ku], d_v = ray.d[kv], d_k = ray.d[sh.k];
            t = (sh.n_d - o_u * sh.n_u - o_v * sh.n_v - o_k) / (d_u * sh.n_u + d_v * sh.n_v + d_k);
            if (t < mint || t > maxt) return false;
            F hu = o_u + t * d_u - sh.a_u, hv = o_v + t * d_v - sh.a_v;
            u = hv * sh.b_nu + hu * sh.b_nv;
            v = hu * sh.c_nu + hv * sh.c_nv;
            return u >= 0 && v >= 0 && u + v <= F(1);
        } else if (sh.type == DRMLT_SHAPE_RECTANGLE) {
            // rectangle.cpp:125-148: transform to object space, hit the z=0 plane
            V3<F> o = xfPoint(sh.w2o, ray.o), d = xfVector(sh.w2o, ray.d);
            F hit = -o.z / d.z;
            if (!(hit >= mint && hit <= maxt)) return false;
            F lx = o.x + hit * d.x, ly = o.y + hit * d.y;
            if (std::abs(lx) <= 1 && std::abs(ly) <= 1) { t = hit; u = lx; v = ly; return true; }
            return false;
        } else {
            // sphere.cpp:163-188 (always solved in double)
            double ox = (double) ray.o.x - (double) sh.center.x, oy = (double) ray.o.y - (double) sh.center.y,
                   oz = (double) ray.o.z - (double) sh.center.z;
            double dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
            double A = dx * dx + dy * dy + dz * dz, B = 2 * (ox * dx + oy * dy + oz * dz),
                   C = ox * ox + oy * oy + oz * oz - (double) sh.radius * (double) sh.radius;
            double disc = B * B - 4 * A * C;
            if (A == 0 || disc < 0) return false;
            double sq = std::sqrt(disc), temp = (B < 0) ? -0.5 * (B - sq) : -0.5 * (B + sq);
            double x0 = temp / A, x1 = C / temp;
            if (x0 > x1) std::swap(x0, x1);
            if (!(x0 <= maxt && x1 >= mint)) return false;
            if (x0 < mint) {
                if (x1 > maxt) return false;
                t = (F) x1;
            } else {
                t = (F) x0;
            }
            u = v = 0;
            return true;
        }
    }

    // skdtree.cpp:112-142 + skdtree.h:340-429
    bool rayIntersect(const Ray<F> &ray, Intersection<F> &its, uint64_t *rayCounter = nullptr) const {
        if (rayCounter) ++*rayCounter;
        its.valid = false;
        its.t = std::numeric_limits<F>::infinity();
        F mint, maxt;
        if (!aabbClip(ray, mint, maxt)) return false;
        F rayMinT = ray.mint;
        if (rayMinT == Consts<F>::Epsilon)
            rayMinT *= std::max(std::max(std::max(std::abs(ray.o.x), std::abs(ray.o.y)), std::abs(ray.o.z)),
                                Consts<F>::Epsilon);
        if (rayMinT > mint) mint = rayMinT;
        if (ray.maxt < maxt) maxt = ray.maxt;
        if (!(maxt > mint)) return false;
        int best = -1;
        F bu = 0, bv = 0;
        for (size_t i = 0; i < shapes.size(); ++i) {
            F t, u, v;
            if (intersectShape(shapes[i], ray, mint, maxt, t, u, v)) {
                maxt = t; best = (int) i; bu = u; bv = v;
            }
        }
        if (best < 0) return false;
        its.t = maxt;
        fillIntersection(ray, best, bu, bv, its);
        return true;
    }

    // skdtree.cpp:207-226 (any hit)
    bool rayOccluded(const Ray<F> &ray, uint64_t *rayCounter = nullptr) const {
        if (rayCounter) ++*rayCounter;
        F mint, maxt;
        if (!aabbClip(ray, mint, maxt)) return false;
        F rayMinT = ray.mint;
        if (rayMinT == Consts<F>::Epsilon)
            rayMinT *= std::max(std::max(std::abs(ray.o.x), std::abs(ray.o.y)), std::abs(ray.o.z));
        if (rayMinT > mint) mint = rayMinT;
        if (ray.maxt < maxt) maxt = ray.maxt;
        if (!(maxt > mint)) return false;
        for (const Shape<F> &sh : shapes) {
            F t, u, v;
            if (intersectShape(sh, ray, mint, maxt, t, u, v)) return true;
        }
        return false;
    }

    void fillIntersection(const Ray<F> &ray, int idx, F u, F v, Intersection<F> &its) const {
        const Shape<F> &sh = shapes[idx];
        its.valid = true;
        its.shape = idx;
        if (sh.type == DRMLT_SHAPE_TRIANGLE) {
            // barycentric position (fillIntersectionRecord<true>), face normal as shading normal
            its.p = sh.p0 * (1 - u - v) + sh.p1 * u + sh.p2 * v;
            its.geoFrame = sh.frame;
            its.dpdu = sh.dpdu;
            its.shFrame = shadingFrame(sh.frame.n, its.dpdu);
        } else if (sh.type == DRMLT_SHAPE_RECTANGLE) {
            its.p = ray.o + ray.d * its.t;
            its.geoFrame = sh.frame;
            its.dpdu = sh.dpdu;
            its.shFrame = shadingFrame(sh.frame.n, its.dpdu);
        } else {
            // sphere.cpp:207-255
            its.p = ray.o + ray.d * its.t;
            if (sizeof(F) == 4) its.p = sh.center + normalize(its.p - sh.center) * sh.radius;
            V3<F> local = its.p - sh.center;
            its.dpdu = V3<F>(-local.y, local.x, 0) * F(2 * kPi);
            V3<F> n = normalize(its.p - sh.center);
            F zrad = std::sqrt(local.x * local.x + local.y * local.y);
            its.geoFrame.n = n;
            if (zrad > 0) {
                F theta = safe_acos(local.z / sh.radius);
                F cosPhi = local.x / zrad, sinPhi = local.y / zrad;
                V3<F> dpdv = V3<F>(local.z * cosPhi, local.z * sinPhi, -std::sin(theta) * sh.radius) * F(kPi);
                its.geoFrame.s = normalize(its.dpdu);
                its.geoFrame.t = normalize(dpdv);
                its.shFrame = shadingFrame(n, its.dpdu);
            } else {
                coordinateSystem(n, its.geoFrame.s, its.geoFrame.t);
                // dpdu is zero at the poles; the reference's shading frame is NaN there
                its.shFrame = its.geoFrame;
            }
        }
        its.wi = its.toLocal(-ray.d);
    }

    // ---- emitters ------------------------------------------------------
    struct DirectSample {
        V3<F> ref, refN, p, n, d;
        F dist = 0, pdf = 0;
        int emitter = -1;
    };

    // pmf.h:124-139 sample + :164-170 sampleReuse
    size_t sampleEmitterIndex(F &sampleValue, F &pdf) const {
        const std::vector<F> &cdf = emitterCdf;
        auto entry = std::lower_bound(cdf.begin(), cdf.end(), sampleValue);
        size_t index = (size_t) std::max((ptrdiff_t) 0, (ptrdiff_t) (entry - cdf.begin()) - 1);
        index = std::min(cdf.size() - 2, index);
        while (cdf[index + 1] - cdf[index] == 0 && index < cdf.size() - 1) ++index;
        pdf = cdf[index + 1] - cdf[index];
        sampleValue = (sampleValue - cdf[index]) / (cdf[index + 1] - cdf[index]);
        return index;
    }

    void samplePosition(const Shape<F> &sh, F sx, F sy, V3<F> &p, V3<F> &n, F &pdf) const {
        if (sh.type == DRMLT_SHAPE_RECTANGLE) { // rectangle.cpp:210-216
            p = xfPoint(sh.o2w, V3<F>(sx * 2 - 1, sy * 2 - 1, 0));
            n = sh.frame.n;
        } else if (sh.type == DRMLT_SHAPE_SPHERE) { // sphere.cpp:257-268, warp.cpp:25-31
            F z = 1 - 2 * sy, r = safe_sqrt(1 - z * z);
            V3<F> v(r * std::cos(F(2 * kPi) * sx), r * std::sin(F(2 * kPi) * sx), z);
            p = sh.center + v * sh.radius;
            n = v;
        } else { // single triangle: Triangle::sample (squareToUniformTriangle)
            F a = safe_sqrt(F(1) - sx);
            F bx = 1 - a, by = a * sy;
            p = sh.p0 + (sh.p1 - sh.p0) * bx + (sh.p2 - sh.p0) * by;
            n = sh.frame.n;
        }
        pdf = sh.invArea;
    }

    // util.cpp:447-485
    static bool solveQuadratic(F a, F b, F c, F &x0, F &x1) {
        if (a == 0) { if (b != 0) { x0 = x1 = -c / b; return true; } return false; }
        F discrim = b * b - 4 * a * c;
        if (discrim < 0) return false;
        F sqrtDiscrim = std::sqrt(discrim);
        F temp = b < 0 ? F(-0.5) * (b - sqrtDiscrim) : F(-0.5) * (b + sqrtDiscrim);
        x0 = temp / a; x1 = c / temp;
        if (x0 > x1) std::swap(x0, x1);
        return true;
    }

    // Shape::sampleDirect (shape.cpp:102-116) or, for a sphere, the cone sampling of sphere.cpp:286-355.
    // Fills p, n, d, dist and the solid-angle pdf of dRec.
    void shapeSampleDirect(const Shape<F> &sh, DirectSample &dRec, F sx, F sy) const {
        if (sh.type != DRMLT_SHAPE_SPHERE) {
            samplePosition(sh, sx, sy, dRec.p, dRec.n, dRec.pdf);
            dRec.d = dRec.p - dRec.ref;
            F distSquared = dRec.d.lengthSquared();
            dRec.dist = std::sqrt(distSquared);
            dRec.d /= dRec.dist;
            F dp = absDot(dRec.d, dRec.n);
            dRec.pdf *= dp != 0 ? (distSquared / dp) : F(0);
            return;
        }
        const V3<F> refToCenter = sh.center - dRec.ref;
        const F refDist2 = refToCenter.lengthSquared();
        const F invRefDist = F(1) / std::sqrt(refDist2);
        const F sinAlpha = sh.radius * invRefDist;
        if (sinAlpha < 1 - Consts<F>::Epsilon) { // outside: sample the cone subtended by the sphere
            F cosAlpha = safe_sqrt(1 - sinAlpha * sinAlpha);
            F cosTheta = (1 - sx) + sx * cosAlpha, sinTheta = safe_sqrt(1 - cosTheta * cosTheta);
            V3<F> local(std::cos(F(2 * kPi) * sy) * sinTheta, std::sin(F(2 * kPi) * sy) * sinTheta, cosTheta);
            dRec.d = Frame<F>(refToCenter * invRefDist).toWorld(local);
            dRec.pdf = F(0.5 * kInvPi) / (1 - cosAlpha);
            const F projDist = dot(refToCenter, dRec.d);
            const F baseT = refDist2 / projDist;
            const V3<F> query = dRec.ref + dRec.d * baseT;
            const V3<F> queryToCenter = sh.center - query;
            const F queryDist2 = queryToCenter.lengthSquared(), queryProjDist = dot(queryToCenter, dRec.d);
            F nearT, farT;
            if (!solveQuadratic(F(1), -2 * queryProjDist, queryDist2 - sh.radius * sh.radius, nearT, farT)) nearT = queryProjDist;
            dRec.dist = baseT + nearT;
            dRec.n = normalize(dRec.d * nearT - queryToCenter);
            dRec.p = sh.center + dRec.n * sh.radius;
        } else { // inside: uniform on the sphere
            F pdfPos;
            samplePosition(sh, sx, sy, dRec.p, dRec.n, pdfPos);
            dRec.d = dRec.p - dRec.ref;
            F dist2 = dRec.d.lengthSquared();
            dRec.dist = std::sqrt(dist2);
            dRec.d /= dRec.dist;
            dRec.pdf = sh.invArea * dist2 / absDot(dRec.d, dRec.n);
        }
    }
    // Shape::pdfDirect (shape.cpp:118-127) / sphere.cpp:357-385, solid-angle measure
    F shapePdfDirect(const Shape<F> &sh, const DirectSample &dRec) const {
        if (sh.type == DRMLT_SHAPE_SPHERE) {
            const V3<F> refToCenter = sh.center - dRec.ref;
            const F sinAlpha = sh.radius / refToCenter.length();
            if (sinAlpha < 1 - Consts<F>::Epsilon) return F(0.5 * kInvPi) / (1 - safe_sqrt(1 - sinAlpha * sinAlpha));
        }
        return sh.invArea * (dRec.dist * dRec.dist) / absDot(dRec.d, dRec.n);
    }

    // scene.cpp:879-904; testVisibility = false is what PathVertex::sampleDirect asks for (vertex.cpp:1307: the connection
    // edge tests visibility itself)
    V3<F> sampleEmitterDirect(DirectSample &dRec, F sx, F sy, uint64_t *rayCounter, bool testVisibility = true) const {
        F emPdf;
        size_t index = sampleEmitterIndex(sx, emPdf);
        const Emitter<F> &em = emitters[index];
        const Shape<F> &sh = shapes[em.shape];
        shapeSampleDirect(sh, dRec, sx, sy);
        // AreaLight::sampleDirect, area.cpp:164-178
        V3<F> value(0);
        if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0 && dRec.pdf != 0) {
            value = em.radiance / dRec.pdf;
        } else {
            dRec.pdf = 0;
        }
        if (dRec.pdf != 0) {
            if (testVisibility) {
                Ray<F> ray{dRec.ref, dRec.d, Consts<F>::Epsilon, dRec.dist * (1 - Consts<F>::ShadowEpsilon)};
                if (rayOccluded(ray, rayCounter)) return V3<F>(0);
            }
            dRec.emitter = (int) index;
            dRec.pdf *= emPdf;
            value /= emPdf;
            return value;
        }
        return V3<F>(0);
    }

    // scene.cpp:1057-1060 with area.cpp:180-189, shape.cpp:118-127
    F pdfEmitterDirect(const DirectSample &dRec) const {
        const Emitter<F> &em = emitters[dRec.emitter];
        F pdf = 0;
        if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0)
            pdf = shapePdfDirect(shapes[em.shape], dRec);
        F discrete = emitterCdf[dRec.emitter + 1] - emitterCdf[dRec.emitter];
        return pdf * discrete;
    }

    // The same density under the AREA measure (Scene::pdfEmitterDirect with dRec.measure = EArea: what
    // PathVertex::evalPdfDirect asks for in Path::miWeight, path.cpp:944-952; shape.cpp:118-127, sphere.cpp:356-385)
    F pdfEmitterDirectArea(const DirectSample &dRec) const {
        const Emitter<F> &em = emitters[dRec.emitter];
        const Shape<F> &sh = shapes[em.shape];
        F pdf = 0;
        if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0) {
            pdf = sh.invArea;
            if (sh.type == DRMLT_SHAPE_SPHERE) {
                const V3<F> refToCenter = sh.center - dRec.ref;
                const F sinAlpha = sh.radius / refToCenter.length();
                if (sinAlpha < 1 - Consts<F>::Epsilon)
                    pdf = F(0.5 * kInvPi) / (1 - safe_sqrt(1 - sinAlpha * sinAlpha)) * absDot(dRec.d, dRec.n) / (dRec.dist * dRec.dist);
            }
        }
        return pdf * (emitterCdf[dRec.emitter + 1] - emitterCdf[dRec.emitter]);
    }

    // area.cpp:111-116
    V3<F> emitterEval(int emitter, const V3<F> &n, const V3<F> &d) const {
        if (dot(n, d) <= 0) return V3<F>(0);
        return emitters[emitter].radiance;
    }

    // ---- BSDFs (local frame, ERadiance mode) -----------------------------
    V3<F> bsdfEval(const Bsdf<F> &b, const V3<F> &wi, const V3<F> &wo) const {
        if (b.type == DRMLT_BSDF_DIFFUSE) {
            if (wi.z <= 0 || wo.z <= 0) return V3<F>(0);
            return b.rgb * (F(kInvPi) * wo.z);
        }
        if (b.type == DRMLT_BSDF_ROUGHCONDUCTOR) return b.rough().eval(wi, wo);
        return V3<F>(0); // delta BSDFs evaluate to zero under the solid-angle measure
    }
    F bsdfPdf(const Bsdf<F> &b, const V3<F> &wi, const V3<F> &wo) const {
        if (b.type == DRMLT_BSDF_DIFFUSE) {
            if (wi.z <= 0 || wo.z <= 0) return 0;
            return squareToCosineHemispherePdf(wo);
        }
        if (b.type == DRMLT_BSDF_ROUGHCONDUCTOR) return b.rough().pdf(wi, wo);
        return 0;
    }
    // returns weight = f*cos/pdf; `delta`: sampled a Dirac component
    // `mode`: 1 = ERadiance (default), 0 = EImportance -- only the dielectric's radiance scaling depends on it
    V3<F> bsdfSample(const Bsdf<F> &b, const V3<F> &wi, F sx, F sy, V3<F> &wo, F &pdf, F &eta, bool &delta,
                     int mode = 1) const {
        eta = 1; delta = false;
        if (b.type == DRMLT_BSDF_DIFFUSE) {
            if (wi.z <= 0) return V3<F>(0);
            wo = squareToCosineHemisphere(sx, sy);
            pdf = squareToCosineHemispherePdf(wo);
            return b.rgb;
        }
        if (b.type == DRMLT_BSDF_ROUGHCONDUCTOR) return b.rough().sample(wi, sx, sy, wo, pdf);
        // dielectric.cpp:270-333, both components enabled
        delta = true;
        F cosThetaT;
        F Fr = fresnelDielectricExt(wi.z, cosThetaT, b.eta);
        if (sx <= Fr) {
            wo = V3<F>(-wi.x, -wi.y, wi.z);
            pdf = Fr;
            return V3<F>(1);
        }
        F scale = -(cosThetaT < 0 ? b.invEta : b.eta);
        wo = V3<F>(scale * wi.x, scale * wi.y, cosThetaT);
        eta = cosThetaT < 0 ? b.eta : b.invEta;
        pdf = 1 - Fr;
        F factor = mode == 1 ? (cosThetaT < 0 ? b.invEta : b.eta) : F(1); // radiance scaling across the interface
        return V3<F>(factor * factor);
    }
    // dielectric.cpp:227-276, discrete measure (used by the bidirectional vertices for the reverse quantities)
    static constexpr F DeltaEpsilon = F(1e-3);
    bool dielectricMatch(const Bsdf<F> &b, const V3<F> &wi, const V3<F> &wo, F &Fr, F &cosThetaT, bool &reflection) const {
        Fr = fresnelDielectricExt(wi.z, cosThetaT, b.eta);
        if (wi.z * wo.z >= 0) {
            reflection = true;
            return !(std::abs(dot(V3<F>(-wi.x, -wi.y, wi.z), wo) - 1) > DeltaEpsilon);
        }
        reflection = false;
        F scale = -(cosThetaT < 0 ? b.invEta : b.eta);
        return !(std::abs(dot(V3<F>(scale * wi.x, scale * wi.y, cosThetaT), wo) - 1) > DeltaEpsilon);
    }
    F bsdfPdfDelta(const Bsdf<F> &b, const V3<F> &wi, const V3<F> &wo) const {
        if (b.type != DRMLT_BSDF_DIELECTRIC) return 0;
        F Fr, cosThetaT;
        bool refl;
        if (!dielectricMatch(b, wi, wo, Fr, cosThetaT, refl)) return 0;
        return refl ? Fr : 1 - Fr;
    }
    V3<F> bsdfEvalDelta(const Bsdf<F> &b, const V3<F> &wi, const V3<F> &wo, int mode) const {
        if (b.type != DRMLT_BSDF_DIELECTRIC) return V3<F>(0);
        F Fr, cosThetaT;
        bool refl;
        if (!dielectricMatch(b, wi, wo, Fr, cosThetaT, refl)) return V3<F>(0);
        if (refl) return V3<F>(Fr);
        F factor = mode == 1 ? (cosThetaT < 0 ? b.invEta : b.eta) : F(1);
        return V3<F>(factor * factor * (1 - Fr));
    }

    // ---- sensor: perspective.cpp:271-300 ---------------------------------
    Ray<F> sampleRay(F px, F py) const {
        // sampleToCamera: inverse of scale(-1/2,-aspect/2)*translate(-1,-1/aspect)*perspective(fov)
        F sx = px / (F) width, sy = py / (F) height;
        V3<F> nearP((1 - 2 * sx) * tanHalfFov * nearClip, (1 - 2 * sy) * tanHalfFov / aspect * nearClip, nearClip);
        V3<F> d = normalize(nearP);
        F invZ = 1 / d.z;
        Ray<F> ray;
        ray.mint = nearClip * invZ;
        ray.maxt = farClip * invZ;
        ray.o = V3<F>(camToWorld[3], camToWorld[7], camToWorld[11]);
        ray.d = V3<F>(camToWorld[0] * d.x + camToWorld[1] * d.y + camToWorld[2] * d.z,
                      camToWorld[4] * d.x + camToWorld[5] * d.y + camToWorld[6] * d.z,
                      camToWorld[8] * d.x + camToWorld[9] * d.y + camToWorld[10] * d.z);
        return ray;
    }

private:
    void loadTriAccel(Shape<F> &sh) { // triaccel.h:58-91
        static const int mod3[5] = {0, 1, 2, 0, 1};
        V3<F> A = sh.p0, b = sh.p2 - sh.p0, c = sh.p1 - sh.p0, N = cross(c, b);
        int k = 0;
        for (int j = 0; j < 3; ++j)
            if (std::abs(N[j]) > std::abs(N[k])) k = j;
        int u = mod3[k + 1], v = mod3[k + 2];
        F n_k = N[k], denom = b[u] * c[v] - b[v] * c[u];
        if (denom == 0) { sh.k = 3; return; }
        sh.k = k;
        sh.n_u = N[u] / n_k; sh.n_v = N[v] / n_k; sh.n_d = dot(A, N) / n_k;
        sh.b_nu = b[u] / denom; sh.b_nv = -b[v] / denom;
        sh.a_u = A[u]; sh.a_v = A[v];
        sh.c_nu = c[v] / denom; sh.c_nv = -c[u] / denom;
    }
};

// One evaluated PSS point: a SplatList (include/mitsuba/bidir/pathsampler.h:317-360). technique=path and mmlt
// produce exactly one splat (px, py, value); technique=bdpt adds the light-image splats of the t = 1 strategies
// in `more` (pathsampler.cpp:514-519), `hasMain` tells whether the main splat exists (:357-361).
template <typename F> struct SplatList {
    struct Splat { F px, py; V3<F> value; };
    F px = 0, py = 0;
    V3<F> value;
    std::vector<Splat> more;
    bool hasMain = true;
    F luminance = 0;
    int nDims = 0, nRays = 0;
    int s = 0, t = 0; // technique=mmlt: SplatList::setStrategy (pathsampler.cpp:129)
    // pathsampler.cpp:1001-1028. `importance` (two-stage MLT): W x H luminance image, row-major, or null
    void normalize(const float *importance = nullptr, int w = 0, int h = 0) {
        if (importance) {
            luminance = 0;
            auto weigh = [&](F x, F y, V3<F> &v) {
                if (v.isZero()) return;
                int ix = std::min(std::max(0, (int) x), w - 1), iy = std::min(std::max(0, (int) y), h - 1);
                v /= (F) importance[ix + iy * w];
                luminance += oracle::luminance(v);
            };
            if (hasMain) weigh(px, py, value);
            for (Splat &sp : more) weigh(sp.px, sp.py, sp.value);
        }
        if (luminance > 0) {
            F inv = F(1) / luminance;
            value *= inv;
            for (Splat &sp : more) sp.value *= inv;
        }
    }
};

// path.cpp:123-315 (strictNormals=false, hideEmitters=false, minDepth=0, no env map)
template <typename F>
inline V3<F> pathLi(const Scene<F> &scene, Ray<F> ray, Sampler<F> &sampler, int maxDepth, int rrDepth,
                    bool excludeDirect, uint64_t *rays) {
    enum { EEmitted = 1, EDirect = 4, EIndirect = 8 };
    int type = excludeDirect ? EIndirect : (EEmitted | EDirect | EIndirect); // pathsampler.cpp:558-561
    Intersection<F> its;
    V3<F> Li(0), throughput(1);
    F eta = 1;
    bool non_specular = false;
    int depth = 1;
    scene.rayIntersect(ray, its, rays);
    while (depth <= maxDepth || maxDepth < 0) {
        if (!its.valid) break; // no environment emitter
        const Shape<F> &sh = scene.shapes[its.shape];
        const Bsdf<F> &bsdf = scene.bsdfs[sh.bsdf];
        // emitted radiance on a direct hit: requires a prior non-specular scatter (:162-165)
        if (sh.emitter >= 0 && (type & EEmitted) && non_specular)
            Li += throughput * scene.emitterEval(sh.emitter, its.shFrame.n, -ray.d);
        if (depth >= maxDepth && maxDepth > 0) break;

        typename Scene<F>::DirectSample dRec;
        dRec.ref = its.p;
        dRec.refN = bsdf.transmissiveOrBackside() ? V3<F>(0) : its.shFrame.n;

        if ((type & EDirect) && bsdf.smooth()) {
            F sx, sy;
            sampler.next2D(sx, sy);
            V3<F> value = scene.sampleEmitterDirect(dRec, sx, sy, rays);
            if (!value.isZero()) {
                V3<F> wo = its.toLocal(dRec.d);
                V3<F> bsdfVal = scene.bsdfEval(bsdf, its.wi, wo);
                if (!bsdfVal.isZero()) {
                    F bsdfPdf = scene.bsdfPdf(bsdf, its.wi, wo);
                    F a = dRec.pdf * dRec.pdf, b = bsdfPdf * bsdfPdf;
                    Li += throughput * value * bsdfVal * (a / (a + b));
                }
            }
        }

        F bx, by;
        sampler.next2D(bx, by);
        V3<F> woLocal;
        F bsdfPdf, bEta;
        bool delta;
        V3<F> bsdfWeight = scene.bsdfSample(bsdf, its.wi, bx, by, woLocal, bsdfPdf, bEta, delta);
        if (bsdfWeight.isZero()) break;
        non_specular |= !delta;

        V3<F> wo = its.toWorld(woLocal);
        bool hitEmitter = false;
        V3<F> value(0);
        ray = Ray<F>{its.p, wo, Consts<F>::Epsilon, std::numeric_limits<F>::infinity()};
        if (scene.rayIntersect(ray, its, rays)) {
            const Shape<F> &hs = scene.shapes[its.shape];
            if (hs.emitter >= 0) {
                value = scene.emitterEval(hs.emitter, its.shFrame.n, -ray.d);
                dRec.p = its.p; dRec.n = its.shFrame.n; dRec.d = ray.d; dRec.dist = its.t; dRec.emitter = hs.emitter;
                hitEmitter = true;
            }
        } else {
            break;
        }
        throughput *= bsdfWeight;
        eta *= bEta;
        if (hitEmitter && (type & EDirect)) {
            F lumPdf = !delta ? scene.pdfEmitterDirect(dRec) : F(0);
            if (non_specular) {
                F a = bsdfPdf * bsdfPdf, b = lumPdf * lumPdf;
                Li += throughput * value * (a / (a + b));
            }
        }
        if (!(type & EIndirect)) break;
        type = EDirect | EIndirect; // ERadianceNoEmission
        if (depth++ >= rrDepth) {
            F q = std::min(throughput.max() * eta * eta, F(0.95));
            if (sampler.next1D() >= q) break;
            throughput /= q;
        }
    }
    return Li;
}

// pathsampler.cpp:529-567
template <typename F>
inline void sampleSplats(const Scene<F> &scene, Sampler<F> &sampler, int maxDepth, int rrDepth, bool excludeDirect,
                         SplatList<F> &list) {
    size_t start = sampler.sampleIndex;
    F sx, sy;
    sampler.next2D(sx, sy);
    list.px = sx * (F) scene.width;
    list.py = sy * (F) scene.height;
    Ray<F> ray = scene.sampleRay(list.px, list.py);
    uint64_t rays = 0;
    list.value = pathLi(scene, ray, sampler, maxDepth, rrDepth, excludeDirect, &rays);
    list.luminance = luminance(list.value);
    list.nDims = (int) (sampler.sampleIndex - start);
    list.nRays = (int) rays;
}

} // namespace oracle
