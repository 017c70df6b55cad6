// Small device math layer for the DRMLT kernels (gfx950): float3 helpers, Philox4x32-10,
// hardware transcendental wrappers. v_sin_f32 / v_cos_f32 take their argument in revolutions,
// which is exactly what every use on this path wants (2*pi*u).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEV __device__ __forceinline__

#define EPSILON_F 1e-4f        // Mitsuba single-precision Epsilon (core/constants.h:28)
#define SHADOW_EPSILON_F 1e-3f // ShadowEpsilon (core/constants.h:29)
#define INV_PI_F 0.31830988618379067154f
#define PI_F 3.14159265358979323846f

struct f3 {
    float x, y, z;
};
DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
DEV f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
DEV f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
DEV f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
DEV float dot3(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
DEV f3 cross3(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DEV f3 fma3(f3 a, float s, f3 b) { return f3{fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)}; }
DEV float max3(f3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
DEV bool is_zero3(f3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }
DEV f3 normalize3(f3 a) { return a * rsqrtf(dot3(a, a)); }
DEV f3 ld3(const float *p) { return f3{p[0], p[1], p[2]}; }
DEV float luminance3(f3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; } // spectrum.h:734-736

// sin/cos of 2*pi*rev. |rev| <= 256 (hardware domain); callers pass values in [-1, 2].
DEV float sin_rev(float rev) { return __builtin_amdgcn_sinf(rev); }
DEV float cos_rev(float rev) { return __builtin_amdgcn_cosf(rev); }
DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
DEV float fast_log2(float x) { return __builtin_amdgcn_logf(x); } // v_log_f32 = log2
DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// ---- Philox4x32-10 (Salmon et al. 2011), counter = (c0,c1,c2,c3), key = (k0,k1) ------------
struct u4 {
    uint32_t x, y, z, w;
};
DEV u4 philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a quarter-rate mul_hi / mul_lo pair
        const unsigned long long p0 = (unsigned long long) 0xD2511F53u * c0, p1 = (unsigned long long) 0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t) (p0 >> 32), lo0 = (uint32_t) p0, hi1 = (uint32_t) (p1 >> 32), lo1 = (uint32_t) p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u4{c0, c1, c2, c3};
}
// 24-bit uniform in [0,1): same value the oracle's Random::nextFloat produces
DEV float u32_to_unit(uint32_t w) { return (float) (w >> 8) * (1.0f / 16777216.0f); }
DEV float pick4(u4 b, uint32_t i) {
    uint32_t w = (i & 2u) ? ((i & 1u) ? b.w : b.z) : ((i & 1u) ? b.y : b.x);
    return u32_to_unit(w);
}

// stream tags: counter word 3 (DESIGN.md "RNG addressing"; same numbering as oracle_sampler.hpp)
enum : uint32_t { TAG_BOOT = 0, TAG_SEEDSEL = 1, TAG_COIN = 2, TAG_S1 = 3, TAG_S2 = 4, TAG_PT = 5 };
