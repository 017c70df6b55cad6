// Helpers shared by the chain kernels of all techniques (kernels.hip, kernels_mmlt.hip, kernels_bdpt.hip): film splats,
// splat normalisation, wave reductions. The decision logic they share is device_mh.h.
#pragma once
#include "device_path.h"
#include "device_mh.h"

#define CHAIN_BLOCK 64 // one wave per workgroup: no barriers anywhere on the chain path

// ImageBlock::put (imageblock.h:150-216) through the 32-entry filter table (rfilter.h:76-77).
// The reference's work-unit blocks carry a border that is dropped when merged into m_accum;
// clamping the footprint to the film gives the same sums.
DEV void film_put(const DParams &P, float px, float py, f3 v) {
    if (P.debug & 1) return;
    if (!(isfinite(v.x) && isfinite(v.y) && isfinite(v.z)) || v.x < 0.f || v.y < 0.f || v.z < 0.f) return;
    float posx = px - 0.5f, posy = py - 0.5f;
    int minx = max((int) ceilf(posx - P.filter_radius), 0), miny = max((int) ceilf(posy - P.filter_radius), 0);
    int maxx = min((int) floorf(posx + P.filter_radius), P.width - 1), maxy = min((int) floorf(posy + P.filter_radius), P.height - 1);
    const bool box = P.box_weight > 0.f; // box table = one constant in entries 0..30 and 0 in entry 31
    for (int y = miny; y <= maxy; ++y) {
        const int iy = min((int) fabsf(((float) y - posy) * P.filter_scale), 31);
        float wy = box ? (iy < 31 ? P.box_weight : 0.f) : P.filter_lut[iy];
        for (int x = minx; x <= maxx; ++x) {
            const int ix = min((int) fabsf(((float) x - posx) * P.filter_scale), 31);
            float w = (box ? (ix < 31 ? P.box_weight : 0.f) : P.filter_lut[ix]) * wy;
            float *dst = P.film + ((size_t) y * P.width + x) * 3;
            atomicAdd(dst + 0, w * v.x);
            atomicAdd(dst + 1, w * v.y);
            atomicAdd(dst + 2, w * v.z);
        }
    }
}

// per-field select: a reference/pointer select between two structs would force them into scratch memory
DEV DSplat select_splat(bool c, const DSplat &a, const DSplat &b) {
    DSplat r;
    r.lum = c ? a.lum : b.lum; r.px = c ? a.px : b.px; r.py = c ? a.py : b.py;
    r.r = c ? a.r : b.r; r.g = c ? a.g : b.g; r.b = c ? a.b : b.b;
    return r;
}

// SplatList::normalize, pathsampler.cpp:1001-1028. Two-stage MLT (`importance` != NULL): the splat is first divided
// by the luminance image at its pixel and the list luminance recomputed, so chains sample f / importance.
DEV void normalize_splat(DSplat &s, const DParams &P) {
    if (P.importance) {
        float lum = 0.f;
        if (!(s.r == 0.f && s.g == 0.f && s.b == 0.f)) {
            const int ix = min(max(0, (int) s.px), P.width - 1), iy = min(max(0, (int) s.py), P.height - 1);
            const float lv = P.importance[ix + iy * P.width];
            s.r /= lv; s.g /= lv; s.b /= lv;
            lum = luminance3(mk3(s.r, s.g, s.b));
        }
        s.lum = lum;
    }
    if (s.lum > 0.f) {
        float inv = 1.f / s.lum;
        s.r *= inv; s.g *= inv; s.b *= inv;
    }
}

DEV unsigned long long wave_sum(uint32_t v) {
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

// wave-reduce the event counters of a launch, one atomic per counter per wave (layout of DParams::stats)
DEV void flush_counters(const DParams &P, const Counters &ct, uint32_t lane) {
    unsigned long long v[9];
    v[0] = wave_sum(ct.large_acc1l & 0xffffu); v[1] = wave_sum(ct.large_acc1l >> 16);
    v[2] = wave_sum(ct.acc1b_secl & 0xffffu);  v[3] = wave_sum(ct.acc1b_secl >> 16);
    v[4] = wave_sum(ct.secb_acc2l & 0xffffu);  v[5] = wave_sum(ct.secb_acc2l >> 16);
    v[6] = wave_sum(ct.acc2b_rev & 0xffffu);   v[7] = wave_sum(ct.acc2b_rev >> 16);
    v[8] = wave_sum(ct.rays);
    if (lane == 0)
        for (int i = 0; i < 9; ++i) if (v[i]) atomicAdd(P.stats + i, v[i]);
}

// field-by-field copy of the parameter block out of the kernarg segment (constant address space: scalar loads)
typedef const DParams __attribute__((address_space(4))) *KArgPtr;
DEV void load_params(DParams &dst, KArgPtr src) {
    static_assert(sizeof(DParams) % 8 == 0, "DParams is copied in 8-byte words");
    typedef const unsigned long long __attribute__((address_space(4))) *KWords;
    const KWords q = (KWords) src;
    unsigned long long *d = reinterpret_cast<unsigned long long *>(&dst);
#pragma unroll
    for (unsigned i = 0; i < sizeof(DParams) / 8u; ++i) d[i] = q[i];
}
// A section's own copy of the parameter block, read through a kernarg pointer the compiler cannot see through: the fields the
// section uses are scalar loads at its head (scalar-cache hits) and dead at its end, instead of scalar registers that stay live --
// and spill into vector lanes -- across the whole chain loop (k_mutate_v4: 200 -> 85 spilled SGPRs). The kernel's first parameter
// must be the DParams block.
#define SECTION_PARAMS_OF_KERNEL(name)                                                          \
    KArgPtr name##_q = (KArgPtr) __builtin_amdgcn_kernarg_segment_ptr();                        \
    asm volatile("" : "+s"(name##_q));                                                          \
    DParams name;                                                                               \
    load_params(name, name##_q)
