// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
//
// Isotropic microfacet distribution with visible-normal sampling and the rough conductor BSDF:
//   src/bsdfs/microfacet.h:191-235 (eval), :421-466 (sampleVisible / pdfVisible),
//   :477-514 (smithG1), :573-691 (sampleVisible11); src/libcore/math.cpp:25-72 (erfinv, erf)
//   src/bsdfs/roughconductor.cpp:258-409 (eval / pdf / sample), util.cpp:723-745 (Fresnel)
#pragma once
#include "oracle_math.hpp"

namespace oracle {

template <typename F> inline F mts_erfinv(F x) { // Giles' approximation, math.cpp:25-53
    F w = -std::log((F(1) - x) * (F(1) + x));
    F p;
    if (w < F(5)) {
        w = w - F(2.5);
        p = F(2.81022636e-08);
        p = F(3.43273939e-07) + p * w; p = F(-3.5233877e-06) + p * w; p = F(-4.39150654e-06) + p * w;
        p = F(0.00021858087) + p * w;  p = F(-0.00125372503) + p * w; p = F(-0.00417768164) + p * w;
        p = F(0.246640727) + p * w;    p = F(1.50140941) + p * w;
    } else {
        w = std::sqrt(w) - F(3);
        p = F(-0.000200214257);
        p = F(0.000100950558) + p * w; p = F(0.00134934322) + p * w; p = F(-0.00367342844) + p * w;
        p = F(0.00573950773) + p * w;  p = F(-0.0076224613) + p * w; p = F(0.00943887047) + p * w;
        p = F(1.00167406) + p * w;     p = F(2.83297682) + p * w;
    }
    return p * x;
}
template <typename F> inline F mts_erf(F x) { // A&S 7.1.26, math.cpp:55-72
    const F a1 = F(0.254829592), a2 = F(-0.284496736), a3 = F(1.421413741), a4 = F(-1.453152027), a5 = F(1.061405429),
            p = F(0.3275911);
    F sign = x < 0 ? F(-1) : (x > 0 ? F(1) : F(0));
    x = std::abs(x);
    F t = F(1) / (F(1) + p * x);
    F y = F(1) - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * std::exp(-x * x);
    return sign * y;
}

template <typename F> struct Microfacet {
    bool ggx;
    F alpha;
    Microfacet(bool g, F a) : ggx(g), alpha(std::max(a, F(1e-4))) {}

    F eval(const V3<F> &m) const {
        if (m.z <= 0) return 0;
        F cos2 = m.z * m.z;
        F be = ((m.x * m.x + m.y * m.y) / (alpha * alpha)) / cos2;
        F result;
        if (!ggx) result = std::exp(-be) / (F(kPi) * alpha * alpha * cos2 * cos2);
        else { F root = (F(1) + be) * cos2; result = F(1) / (F(kPi) * alpha * alpha * root * root); }
        if (result * m.z < F(1e-20)) result = 0;
        return result;
    }
    F smithG1(const V3<F> &v, const V3<F> &m) const {
        if (dot(v, m) * v.z <= 0) return 0;
        F temp = 1 - v.z * v.z;
        F tanTheta = temp <= 0 ? F(0) : std::abs(std::sqrt(temp) / v.z);
        if (tanTheta == 0) return 1;
        if (!ggx) {
            F a = F(1) / (alpha * tanTheta);
            if (a >= F(1.6)) return 1;
            F aSqr = a * a;
            return (F(3.535) * a + F(2.181) * aSqr) / (F(1) + F(2.276) * a + F(2.577) * aSqr);
        }
        F root = alpha * tanTheta;
        return F(2) / (F(1) + std::hypot(F(1), root));
    }
    F G(const V3<F> &wi, const V3<F> &wo, const V3<F> &m) const { return smithG1(wi, m) * smithG1(wo, m); }
    F pdfVisible(const V3<F> &wi, const V3<F> &m) const {
        if (wi.z == 0) return 0;
        return smithG1(wi, m) * absDot(wi, m) * eval(m) / std::abs(wi.z);
    }
    void sampleVisible11(F thetaI, F sx, F sy, F &slx, F &sly) const {
        const F SQRT_PI_INV = F(1) / std::sqrt(F(kPi));
        if (!ggx) {
            if (thetaI < F(1e-4)) {
                F r = std::sqrt(-std::log(F(1) - sx));
                slx = r * std::cos(F(2 * kPi) * sy); sly = r * std::sin(F(2 * kPi) * sy);
                return;
            }
            F tanThetaI = std::tan(thetaI), cotThetaI = 1 / tanThetaI;
            F a = -1, c = mts_erf(cotThetaI);
            F sample_x = std::max(sx, F(1e-6));
            F fit = 1 + thetaI * (F(-0.876) + thetaI * (F(0.4265) - F(0.0594) * thetaI));
            F b = c - (1 + c) * std::pow(1 - sample_x, fit);
            F normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * std::exp(-cotThetaI * cotThetaI));
            int it = 0;
            while (++it < 10) {
                if (!(b >= a && b <= c)) b = F(0.5) * (a + c);
                F invErf = mts_erfinv(b);
                F value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * std::exp(-invErf * invErf)) - sample_x;
                F derivative = normalization * (1 - invErf * tanThetaI);
                if (std::abs(value) < F(1e-5)) break;
                if (value > 0) c = b; else a = b;
                b -= value / derivative;
            }
            slx = mts_erfinv(b);
            sly = mts_erfinv(F(2) * std::max(sy, F(1e-6)) - F(1));
            return;
        }
        if (thetaI < F(1e-4)) {
            F r = safe_sqrt(sx / (1 - sx));
            slx = r * std::cos(F(2 * kPi) * sy); sly = r * std::sin(F(2 * kPi) * sy);
            return;
        }
        F tanThetaI = std::tan(thetaI);
        F a = 1 / tanThetaI;
        F G1 = F(2) / (F(1) + safe_sqrt(F(1) + F(1) / (a * a)));
        F A = F(2) * sx / G1 - F(1);
        if (std::abs(A) == 1) A -= (A < 0 ? F(-1) : F(1)) * Consts<F>::Epsilon;
        F tmp = F(1) / (A * A - F(1));
        F B = tanThetaI;
        F D = safe_sqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
        F s1 = B * tmp - D, s2 = B * tmp + D;
        slx = (A < 0 || s2 > F(1) / tanThetaI) ? s1 : s2;
        F S;
        if (sy > F(0.5)) { S = 1; sy = F(2) * (sy - F(0.5)); } else { S = -1; sy = F(2) * (F(0.5) - sy); }
        F z = (sy * (sy * (sy * F(-0.365728915865723) + F(0.790235037209296)) - F(0.424965825137544)) + F(0.000152998850436920)) /
              (sy * (sy * (sy * (sy * F(0.169507819808272) - F(0.397203533833404)) - F(0.232500544458471)) + F(1)) - F(0.539825872510702));
        sly = S * z * std::sqrt(F(1) + slx * slx);
    }
    V3<F> sampleVisible(const V3<F> &_wi, F sx, F sy) const {
        V3<F> wi = normalize(V3<F>(alpha * _wi.x, alpha * _wi.y, _wi.z));
        F theta = 0, phi = 0;
        if (wi.z < F(0.99999)) { theta = std::acos(wi.z); phi = std::atan2(wi.y, wi.x); }
        F sinPhi = std::sin(phi), cosPhi = std::cos(phi);
        F slx, sly;
        sampleVisible11(theta, sx, sy, slx, sly);
        F rx = cosPhi * slx - sinPhi * sly, ry = sinPhi * slx + cosPhi * sly;
        rx *= alpha; ry *= alpha;
        F nrm = F(1) / std::sqrt(rx * rx + ry * ry + F(1));
        return V3<F>(-rx * nrm, -ry * nrm, nrm);
    }
};

template <typename F> struct RoughConductor {
    Microfacet<F> distr;
    V3<F> eta, k, refl;
    V3<F> fresnel(F c) const {
        return V3<F>(fresnelConductorExact(c, eta.x, k.x), fresnelConductorExact(c, eta.y, k.y), fresnelConductorExact(c, eta.z, k.z)) * refl;
    }
    V3<F> eval(const V3<F> &wi, const V3<F> &wo) const { // f * cos(theta_o)
        if (wi.z <= 0 || wo.z <= 0) return V3<F>(0);
        V3<F> H = normalize(wo + wi);
        F D = distr.eval(H);
        if (D == 0) return V3<F>(0);
        return fresnel(dot(wi, H)) * (D * distr.G(wi, wo, H) / (F(4) * wi.z));
    }
    F pdf(const V3<F> &wi, const V3<F> &wo) const {
        if (wi.z <= 0 || wo.z <= 0) return 0;
        V3<F> H = normalize(wo + wi);
        return distr.eval(H) * distr.smithG1(wi, H) / (F(4) * wi.z);
    }
    V3<F> sample(const V3<F> &wi, F sx, F sy, V3<F> &wo, F &pdf) const {
        if (wi.z < 0) return V3<F>(0);
        V3<F> m = distr.sampleVisible(wi, sx, sy);
        F pm = distr.pdfVisible(wi, m);
        if (pm == 0) return V3<F>(0);
        wo = m * (F(2) * dot(wi, m)) - wi;
        if (wo.z <= 0) return V3<F>(0);
        F weight = distr.smithG1(wo, m);
        if (weight > 0) {
            pdf = pm / (F(4) * dot(wo, m));
            return fresnel(dot(wi, m)) * weight;
        }
        return V3<F>(0);
    }
};

} // namespace oracle
