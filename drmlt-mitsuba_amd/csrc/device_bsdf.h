// Rough conductor on the device: isotropic Beckmann / GGX microfacet distribution with
// visible-normal sampling (the reference's defaults).
//   src/bsdfs/microfacet.h:191-235 (eval), :421-466 (sampleVisible / pdfVisible), :477-514 (smithG1),
//   :573-691 (sampleVisible11); src/libcore/math.cpp:25-72 (erfinv / erf approximations);
//   src/bsdfs/roughconductor.cpp:258-409; src/libcore/util.cpp:723-745 (fresnelConductorExact)
#pragma once
#include "device_math.h"

// exp / log on the hardware's base-2 units (~2e-6 relative over the ranges used here, against 1e-7 for libm's):
// the microfacet sampler's Newton loop evaluates them up to ten times per sample
DEV float rc_exp(float x) { return fast_exp2(x * 1.4426950408889634f); }
DEV float rc_log(float x) { return fast_log2(x) * 0.6931471805599453f; }
DEV float mts_erfinv(float x) {
    float w = -rc_log((1.f - x) * (1.f + x));
    float p;
    if (w < 5.f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = fmaf(p, w, 3.43273939e-07f); p = fmaf(p, w, -3.5233877e-06f); p = fmaf(p, w, -4.39150654e-06f);
        p = fmaf(p, w, 0.00021858087f);  p = fmaf(p, w, -0.00125372503f); p = fmaf(p, w, -0.00417768164f);
        p = fmaf(p, w, 0.246640727f);    p = fmaf(p, w, 1.50140941f);
    } else {
        w = sqrtf(w) - 3.f;
        p = -0.000200214257f;
        p = fmaf(p, w, 0.000100950558f); p = fmaf(p, w, 0.00134934322f); p = fmaf(p, w, -0.00367342844f);
        p = fmaf(p, w, 0.00573950773f);  p = fmaf(p, w, -0.0076224613f); p = fmaf(p, w, 0.00943887047f);
        p = fmaf(p, w, 1.00167406f);     p = fmaf(p, w, 2.83297682f);
    }
    return p * x;
}
DEV float mts_erf(float x) {
    const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
    float sign = x < 0.f ? -1.f : (x > 0.f ? 1.f : 0.f);
    x = fabsf(x);
    float t = 1.f / (1.f + p * x);
    float y = 1.f - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * rc_exp(-x * x);
    return sign * y;
}

struct DMicrofacet {
    bool ggx;
    float alpha;
    DEV float eval(f3 m) const {
        if (m.z <= 0.f) return 0.f;
        float cos2 = m.z * m.z;
        float be = ((m.x * m.x + m.y * m.y) / (alpha * alpha)) / cos2;
        float result;
        if (!ggx) result = rc_exp(-be) / (PI_F * alpha * alpha * cos2 * cos2);
        else { float root = (1.f + be) * cos2; result = 1.f / (PI_F * alpha * alpha * root * root); }
        if (result * m.z < 1e-20f) result = 0.f;
        return result;
    }
    DEV float smithG1(f3 v, f3 m) const {
        if (dot3(v, m) * v.z <= 0.f) return 0.f;
        float temp = 1.f - v.z * v.z;
        float tanTheta = temp <= 0.f ? 0.f : fabsf(sqrtf(temp) / v.z);
        if (tanTheta == 0.f) return 1.f;
        if (!ggx) {
            float a = 1.f / (alpha * tanTheta);
            if (a >= 1.6f) return 1.f;
            float aSqr = a * a;
            return (3.535f * a + 2.181f * aSqr) / (1.f + 2.276f * a + 2.577f * aSqr);
        }
        float root = alpha * tanTheta;
        return 2.f / (1.f + sqrtf(1.f + root * root));
    }
    DEV float pdfVisible(f3 wi, f3 m) const {
        if (wi.z == 0.f) return 0.f;
        return smithG1(wi, m) * fabsf(dot3(wi, m)) * eval(m) / fabsf(wi.z);
    }
    // thetaI with its tangent supplied by the caller: tan = sin / cos from the direction's components is the same
    // number as tanf(acosf(z)) without the two libm calls (tanf's range reduction alone is ~100 instructions)
    DEV void sampleVisible11(float thetaI, float tanThetaI, float sx, float sy, float &slx, float &sly) const {
        const float SQRT_PI_INV = 0.5641895835477563f;
        if (!ggx) {
            if (thetaI < 1e-4f) {
                float r = sqrtf(-rc_log(1.f - sx));
                slx = r * cos_rev(sy); sly = r * sin_rev(sy);
                return;
            }
            float cotThetaI = 1.f / tanThetaI;
            float a = -1.f, c = mts_erf(cotThetaI);
            float sample_x = fmaxf(sx, 1e-6f);
            float fit = 1.f + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
            float b = c - (1.f + c) * fast_exp2(fit * fast_log2(1.f - sample_x)); // pow(): only the Newton start value
            float normalization = 1.f / (1.f + c + SQRT_PI_INV * tanThetaI * rc_exp(-cotThetaI * cotThetaI));
            int it = 0;
            while (++it < 10) {
                if (!(b >= a && b <= c)) b = 0.5f * (a + c);
                float invErf = mts_erfinv(b);
                float value = normalization * (1.f + b + SQRT_PI_INV * tanThetaI * rc_exp(-invErf * invErf)) - sample_x;
                float derivative = normalization * (1.f - invErf * tanThetaI);
                if (fabsf(value) < 1e-5f) break;
                if (value > 0.f) c = b; else a = b;
                b -= value / derivative;
            }
            slx = mts_erfinv(b);
            sly = mts_erfinv(2.f * fmaxf(sy, 1e-6f) - 1.f);
            return;
        }
        if (thetaI < 1e-4f) {
            float r = sqrtf(fmaxf(0.f, sx / (1.f - sx)));
            slx = r * cos_rev(sy); sly = r * sin_rev(sy);
            return;
        }
        float a = 1.f / tanThetaI;
        float G1 = 2.f / (1.f + sqrtf(fmaxf(0.f, 1.f + 1.f / (a * a))));
        float A = 2.f * sx / G1 - 1.f;
        if (fabsf(A) == 1.f) A -= (A < 0.f ? -1.f : 1.f) * EPSILON_F;
        float tmp = 1.f / (A * A - 1.f);
        float B = tanThetaI;
        float D = sqrtf(fmaxf(0.f, B * B * tmp * tmp - (A * A - B * B) * tmp));
        float s1 = B * tmp - D, s2 = B * tmp + D;
        slx = (A < 0.f || s2 > 1.f / tanThetaI) ? s1 : s2;
        float S;
        if (sy > 0.5f) { S = 1.f; sy = 2.f * (sy - 0.5f); } else { S = -1.f; sy = 2.f * (0.5f - sy); }
        float z = (sy * (sy * (sy * -0.365728915865723f + 0.790235037209296f) - 0.424965825137544f) + 0.000152998850436920f) /
                  (sy * (sy * (sy * (sy * 0.169507819808272f - 0.397203533833404f) - 0.232500544458471f) + 1.f) - 0.539825872510702f);
        sly = S * z * sqrtf(1.f + slx * slx);
    }
    DEV f3 sampleVisible(f3 _wi, float sx, float sy) const {
        f3 wi = normalize3(mk3(alpha * _wi.x, alpha * _wi.y, _wi.z));
        // (theta, phi) of wi (microfacet.h:428-437). Only tan(theta), cos(phi), sin(phi) are used -- read them off the
        // components -- and theta itself by the Beckmann fit polynomial and the theta < 1e-4 test.
        float theta = 0.f, tanTheta = 0.f, sinPhi = 0.f, cosPhi = 1.f;
        if (wi.z < 0.99999f) {
            const float r2 = wi.x * wi.x + wi.y * wi.y, inv = rsqrtf(r2);
            theta = ggx ? 1.f : acosf(wi.z);             // GGX only compares it with 1e-4
            tanTheta = r2 * inv / wi.z;
            cosPhi = wi.x * inv; sinPhi = wi.y * inv;
        }
        float slx, sly;
        sampleVisible11(theta, tanTheta, sx, sy, slx, sly);
        float rx = (cosPhi * slx - sinPhi * sly) * alpha, ry = (sinPhi * slx + cosPhi * sly) * alpha;
        float nrm = rsqrtf(rx * rx + ry * ry + 1.f);
        return mk3(-rx * nrm, -ry * nrm, nrm);
    }
};

DEV float fresnel_conductor_exact(float cosThetaI, float eta, float k) {
    float cosThetaI2 = cosThetaI * cosThetaI, sinThetaI2 = 1.f - cosThetaI2, sinThetaI4 = sinThetaI2 * sinThetaI2;
    float temp1 = eta * eta - k * k - sinThetaI2;
    float a2pb2 = sqrtf(fmaxf(0.f, temp1 * temp1 + 4.f * k * k * eta * eta));
    float a = sqrtf(fmaxf(0.f, 0.5f * (a2pb2 + temp1)));
    float term1 = a2pb2 + cosThetaI2, term2 = 2.f * a * cosThetaI;
    float Rs2 = (term1 - term2) / (term1 + term2);
    float term3 = a2pb2 * cosThetaI2 + sinThetaI4, term4 = term2 * sinThetaI2;
    float Rp2 = Rs2 * (term3 - term4) / (term3 + term4);
    return 0.5f * (Rp2 + Rs2);
}

// rgb = specularReflectance, p[0] = alpha, p[1..3] = eta, p[4..6] = k, p[7] != 0: GGX
struct DRoughConductor {
    DMicrofacet distr;
    f3 eta, k, refl;
    DEV f3 fresnel(float c) const {
        return mk3(fresnel_conductor_exact(c, eta.x, k.x), fresnel_conductor_exact(c, eta.y, k.y), fresnel_conductor_exact(c, eta.z, k.z)) * refl;
    }
    DEV f3 eval(f3 wi, f3 wo) const { // f * cos(theta_o), roughconductor.cpp:258-295
        if (wi.z <= 0.f || wo.z <= 0.f) return mk3(0.f, 0.f, 0.f);
        f3 H = normalize3(wo + wi);
        float D = distr.eval(H);
        if (D == 0.f) return mk3(0.f, 0.f, 0.f);
        return fresnel(dot3(wi, H)) * (D * distr.smithG1(wi, H) * distr.smithG1(wo, H) / (4.f * wi.z));
    }
    DEV float pdf(f3 wi, f3 wo) const { // :297-323 (sampleVisible)
        if (wi.z <= 0.f || wo.z <= 0.f) return 0.f;
        f3 H = normalize3(wo + wi);
        return distr.eval(H) * distr.smithG1(wi, H) / (4.f * wi.z);
    }
    DEV f3 sample(f3 wi, float sx, float sy, f3 &wo, float &pdf) const { // :371-409
        if (wi.z < 0.f) return mk3(0.f, 0.f, 0.f);
        f3 m = distr.sampleVisible(wi, sx, sy);
        float pm = distr.pdfVisible(wi, m);
        if (pm == 0.f) return mk3(0.f, 0.f, 0.f);
        wo = m * (2.f * dot3(wi, m)) - wi;
        if (wo.z <= 0.f) return mk3(0.f, 0.f, 0.f);
        float weight = distr.smithG1(wo, m);
        if (weight > 0.f) {
            pdf = pm / (4.f * dot3(wo, m));
            return fresnel(dot3(wi, m)) * weight;
        }
        return mk3(0.f, 0.f, 0.f);
    }
};
