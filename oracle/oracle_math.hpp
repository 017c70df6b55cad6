// ORACLE -- TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// CPU restatement of the reference's DRMLT hot path (joeylitalien/drmlt-mitsuba),
// written from the reference's behaviour, templated on Float = float | double.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// Parity status: PARITY UNPINNED. The reference ships no test, scene, golden vector or image for drmlt / pssmlt /
// PathSampler, and none of its sources on this path builds here without stand-ins for Mitsuba's headers
// (Boost, Xerces, OpenEXR are absent): there is no reference build and no reference-generated fixture. What holds the
// restatement instead: line-by-line citations; Philox against Random123's published known answers; every sampling
// routine chi^2-tested against its own pdf (the reference's test_chisquare.cpp strategy); detailed balance of every
// acceptance rule on an analytic target; the estimators against analytic form factors and against each other; and a
// literal replay of the reference's order of operations for the acceptance map (tests/test_oracle_process.py).
// tests/golden/transition_kat.json holds numbers produced in round 1 by compiling the reference's tools/transition.h
// behind a 15-line stand-in for Float / Random / math::* (`make -C oracle ref`): a cross-check of four formulas, NOT a
// reference build by the project's rules, and not claimed as pinning.
//
// This file: small vector math, constants, frames, warps, Fresnel.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>

namespace oracle {

// include/mitsuba/core/constants.h:24-31 (double build vs single build)
template <typename F> struct Consts;
template <> struct Consts<double> {
    static constexpr double Epsilon = 1e-7, ShadowEpsilon = 1e-5, DeltaEpsilon = 1e-7;
};
template <> struct Consts<float> {
    static constexpr float Epsilon = 1e-4f, ShadowEpsilon = 1e-3f, DeltaEpsilon = 1e-3f;
};

constexpr double kPi = 3.14159265358979323846;
constexpr double kInvPi = 0.31830988618379067154;
constexpr double kSqrt1_2Pi = 0.39894228040143267794; // transition.h:6

template <typename F> struct V3 {
    F x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(F a, F b, F c) : x(a), y(b), z(c) {}
    explicit V3(F a) : x(a), y(a), z(a) {}
    F operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    F &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    V3 operator+(const V3 &o) const { return V3(x + o.x, y + o.y, z + o.z); }
    V3 operator-(const V3 &o) const { return V3(x - o.x, y - o.y, z - o.z); }
    V3 operator*(const V3 &o) const { return V3(x * o.x, y * o.y, z * o.z); }
    V3 operator*(F s) const { return V3(x * s, y * s, z * s); }
    V3 operator/(F s) const { return V3(x / s, y / s, z / s); }
    V3 operator-() const { return V3(-x, -y, -z); }
    V3 &operator+=(const V3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    V3 &operator*=(const V3 &o) { x *= o.x; y *= o.y; z *= o.z; return *this; }
    V3 &operator*=(F s) { x *= s; y *= s; z *= s; return *this; }
    V3 &operator/=(F s) { x /= s; y /= s; z /= s; return *this; }
    F lengthSquared() const { return x * x + y * y + z * z; }
    F length() const { return std::sqrt(lengthSquared()); }
    bool isZero() const { return x == 0 && y == 0 && z == 0; }
    F max() const { return std::max(x, std::max(y, z)); }
};
template <typename F> inline F dot(const V3<F> &a, const V3<F> &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename F> inline F absDot(const V3<F> &a, const V3<F> &b) { return std::abs(dot(a, b)); }
template <typename F> inline V3<F> cross(const V3<F> &a, const V3<F> &b) {
    return V3<F>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename F> inline V3<F> normalize(const V3<F> &a) { return a / a.length(); }

// Spectrum = RGB (SPECTRUM_SAMPLES=3); luminance weights spectrum.h:734-736
template <typename F> inline F luminance(const V3<F> &c) {
    return c.x * F(0.212671) + c.y * F(0.715160) + c.z * F(0.072169);
}
// Spectrum::isValid: all finite and non-negative
template <typename F> inline bool spectrumValid(const V3<F> &c) {
    for (int i = 0; i < 3; ++i)
        if (!std::isfinite(c[i]) || c[i] < 0) return false;
    return true;
}

template <typename F> inline F safe_acos(F v) { return std::acos(std::min(F(1), std::max(F(-1), v))); }
template <typename F> inline F safe_sqrt(F v) { return std::sqrt(std::max(F(0), v)); }

// src/libcore/util.cpp:606-616  coordinateSystem(a, b, c)
template <typename F> inline void coordinateSystem(const V3<F> &a, V3<F> &b, V3<F> &c) {
    if (std::abs(a.x) > std::abs(a.y)) {
        F invLen = F(1) / std::sqrt(a.x * a.x + a.z * a.z);
        c = V3<F>(a.z * invLen, 0, -a.x * invLen);
    } else {
        F invLen = F(1) / std::sqrt(a.y * a.y + a.z * a.z);
        c = V3<F>(0, a.z * invLen, -a.y * invLen);
    }
    b = cross(c, a);
}

template <typename F> struct Frame {
    V3<F> s, t, n;
    Frame() {}
    explicit Frame(const V3<F> &nn) : n(nn) { coordinateSystem(n, s, t); }
    Frame(const V3<F> &ss, const V3<F> &tt, const V3<F> &nn) : s(ss), t(tt), n(nn) {}
    V3<F> toLocal(const V3<F> &v) const { return V3<F>(dot(v, s), dot(v, t), dot(v, n)); }
    V3<F> toWorld(const V3<F> &v) const { return s * v.x + t * v.y + n * v.z; }
    static F cosTheta(const V3<F> &v) { return v.z; }
};

// src/librender/shape.cpp computeShadingFrame: s = normalize(dpdu - n (n.dpdu)), t = n x s
template <typename F> inline Frame<F> shadingFrame(const V3<F> &n, const V3<F> &dpdu) {
    Frame<F> fr;
    fr.n = n;
    fr.s = normalize(dpdu - n * dot(n, dpdu));
    fr.t = cross(fr.n, fr.s);
    return fr;
}

// src/libcore/warp.cpp:81-102 (Cline's concentric map) and :43-52
template <typename F> inline void squareToUniformDiskConcentric(F sx, F sy, F &ox, F &oy) {
    F r1 = F(2) * sx - F(1), r2 = F(2) * sy - F(1);
    F phi, r;
    if (r1 == 0 && r2 == 0) {
        r = phi = 0;
    } else if (r1 * r1 > r2 * r2) {
        r = r1;
        phi = F(kPi / 4) * (r2 / r1);
    } else {
        r = r2;
        phi = F(kPi / 2) - (r1 / r2) * F(kPi / 4);
    }
    ox = r * std::cos(phi);
    oy = r * std::sin(phi);
}
template <typename F> inline V3<F> squareToCosineHemisphere(F sx, F sy) {
    F px, py;
    squareToUniformDiskConcentric(sx, sy, px, py);
    F z = safe_sqrt(F(1) - px * px - py * py);
    if (z == 0) z = F(1e-10);
    return V3<F>(px, py, z);
}
template <typename F> inline F squareToCosineHemispherePdf(const V3<F> &d) { return F(kInvPi) * d.z; }

// src/libcore/warp.cpp:30-41 squareToUniformSphere
template <typename F> inline V3<F> squareToUniformSphere(F sx, F sy) {
    F z = F(1) - F(2) * sy;
    F r = safe_sqrt(F(1) - z * z);
    F phi = F(2 * kPi) * sx;
    return V3<F>(r * std::cos(phi), r * std::sin(phi), z);
}

// src/libcore/util.cpp:659-689 fresnelDielectricExt
template <typename F> inline F fresnelDielectricExt(F cosThetaI_, F &cosThetaT_, F eta) {
    if (eta == 1) {
        cosThetaT_ = -cosThetaI_;
        return 0;
    }
    F scale = (cosThetaI_ > 0) ? 1 / eta : eta;
    F cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0) {
        cosThetaT_ = 0;
        return 1;
    }
    F cosThetaI = std::abs(cosThetaI_);
    F cosThetaT = std::sqrt(cosThetaTSqr);
    F Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    F Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return F(0.5) * (Rs * Rs + Rp * Rp);
}

// src/libcore/util.cpp fresnelConductorExact (used by conductor / roughconductor)
template <typename F> inline F fresnelConductorExact(F cosThetaI, F eta, F k) {
    F cosThetaI2 = cosThetaI * cosThetaI, sinThetaI2 = 1 - cosThetaI2, sinThetaI4 = sinThetaI2 * sinThetaI2;
    F temp1 = eta * eta - k * k - sinThetaI2;
    F a2pb2 = safe_sqrt(temp1 * temp1 + 4 * k * k * eta * eta);
    F a = safe_sqrt(F(0.5) * (a2pb2 + temp1));
    F term1 = a2pb2 + cosThetaI2, term2 = 2 * a * cosThetaI;
    F Rs2 = (term1 - term2) / (term1 + term2);
    F term3 = a2pb2 * cosThetaI2 + sinThetaI4, term4 = term2 * sinThetaI2;
    F Rp2 = Rs2 * (term3 - term4) / (term3 + term4);
    return F(0.5) * (Rp2 + Rs2);
}

} // namespace oracle
