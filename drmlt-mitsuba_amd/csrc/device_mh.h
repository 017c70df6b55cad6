// The decision logic of the chain loops, ONE copy for every chain kernel (k_mutate_v3 / v4, k_mutate_mmlt, k_mutate_bdpt):
// acceptance rules of both stages, expectation weights, the event counters behind the seven ratios, and the
// acceptance-map rule. The kernels differ in how a proposal is evaluated and where its splats live (a DSplat in
// registers, a list in memory); what is decided from the luminances is the same code.
//
// Reference (paths relative to the checkout):
//   stage 1 (Eq. 5), doSecond gating          src/integrators/drmlt/drmlt_proc.cpp:544-558
//   Green / Mira / orbital second stage       :588-621 / :625-650 / :655-669
//   expectation weights                       :677-688
//   acceptance map                            :443-450, :693-709
//   statistics                                :711-768 (mixture :336-376)
//   processMixture                            :161-380
#pragma once
#include "device_math.h"

DEV bool lum_invalid(float x) { return isnan(x) || isinf(x) || x <= 0.f; }        // drmlt_proc.cpp:428
DEV bool lum_invalid_mix(float x) { return isnan(x) || isinf(x) || x < 0.f; }     // drmlt_proc.cpp:181

// Per-lane event counters of one launch, packed 2 x 16 bit (launch length is capped at 32768).
struct Counters {
    uint32_t large_acc1l; // lo: large steps                hi: accepted first stage after large
    uint32_t acc1b_secl;  // lo: accepted first stage, bold  hi: second stages after large
    uint32_t secb_acc2l;  // lo: second stages after bold    hi: accepted second stage after large
    uint32_t acc2b_rev;   // lo: accepted second, bold       hi: Green reverse evaluations
    uint32_t rays;
};

// flipCoin (drmlt_proc.cpp:419-422): x >= 1 accepts without a draw; the coins are addressed (TAG_COIN), so "without a
// draw" only means the value is not looked at
DEV bool mh_flip(float a, float coin) { return a >= 1.f || coin < a; }

// First stage: Metropolis-Hastings on the luminances (Eq. 5) and whether a second stage follows.
//   delayed rejection: after a rejected first stage, not after a large step unless timidAfterLarge (:553-558)
//   mixture:           with probability 1/2 on every non-large step, whatever the first test said (:296-299)
DEV void mh_first(bool mix, bool timid_after_large, bool large, float y_lum, float cur_lum, float coin_acc1, float coin_mix,
                  float &a1, bool &acc1, bool &do_second) {
    a1 = 0.f;
    if (!(mix ? lum_invalid_mix(y_lum) : lum_invalid(y_lum))) a1 = fminf(1.f, y_lum / cur_lum);
    acc1 = a1 > 0.f && mh_flip(a1, coin_acc1);
    if (!mix) do_second = !acc1 && (timid_after_large || !large);
    else do_second = !large && coin_mix < 0.5f;
}

// processMixture's second proposal REPLACES the first one and is tested by plain Metropolis-Hastings (:313-324)
DEV void mh_second_mixture(float z_lum, float cur_lum, float coin_acc2, float &a2, bool &acc2) {
    a2 = 0.f; acc2 = false;
    if (!lum_invalid_mix(z_lum)) {
        a2 = fminf(1.f, z_lum / cur_lum);
        acc2 = mh_flip(a2, coin_acc2);
    }
}

// Tierney & Mira (:625-650). `ratio` = Q1(y|z) / Q1(y|x) over the dimensions used (1 after a large step).
DEV void mh_second_mira(float y_lum, float z_lum, float cur_lum, float a1, float ratio, float coin_acc2, float &a2, bool &acc2) {
    a2 = 0.f; acc2 = false;
    const float aRev = fminf(1.f, y_lum / z_lum);
    if (!(aRev >= 1.f) && !lum_invalid(ratio)) {
        a2 = fminf(1.f, (z_lum / cur_lum) * ratio * (1.f - aRev) / (1.f - a1));
        acc2 = mh_flip(a2, coin_acc2);
    }
}

// pairwise orbital, DRMLT Eq. 11 (:655-669)
DEV void mh_second_orbital(float y_lum, float z_lum, float cur_lum, float coin_acc2, float &a2, bool &acc2) {
    a2 = 0.f; acc2 = false;
    if (z_lum < y_lum) return;
    if (z_lum >= cur_lum) { a2 = 1.f; acc2 = true; return; }
    a2 = (z_lum - y_lum) / (cur_lum - y_lum);
    acc2 = mh_flip(a2, coin_acc2);
}

// Green & Mira, Eq. 13-14 (:599-615): `rev_lum` = luminance of the reverse move y* = z - (y - x)
DEV void mh_second_green(float rev_lum, float z_lum, float cur_lum, float a1, float coin_acc2, float &a2, bool &acc2) {
    a2 = 0.f; acc2 = false;
    const float aRev = lum_invalid(rev_lum) ? 0.f : fminf(1.f, rev_lum / z_lum);
    if (aRev != 1.f) {
        a2 = fminf(1.f, (z_lum / cur_lum) * (1.f - aRev) / (1.f - a1));
        acc2 = mh_flip(a2, coin_acc2);
    }
}

// Expectation weights of the current state, the first-stage and the second-stage proposal (:677-688; mixture :327-333,
// where `a` is the acceptance of whichever proposal was tested). An acceptance-map run of the delayed-rejection loop
// splats no radiance (:432); the mixture loop has no such switch (:183-194).
struct MhWeights { float w0, w1, w2; };
DEV MhWeights mh_weights(bool mix, bool amap, bool do_second, float a1, float a2) {
    MhWeights w;
    if (!mix) {
        w.w1 = a1; w.w2 = (1.f - a1) * a2; w.w0 = 1.f - w.w1 - w.w2;
        if (!do_second) w.w2 = 0.f;
        if (amap) w.w0 = w.w1 = w.w2 = 0.f;
    } else {
        const float a = do_second ? a2 : a1;
        w.w0 = 1.f - a; w.w1 = do_second ? 0.f : a; w.w2 = do_second ? a : 0.f;
    }
    return w;
}

// Event counts of one decided mutation (the seven ratios of :34-49 are assembled from them on the host).
DEV void mh_count(Counters &ct, bool large, bool acc1, bool acc2, bool do_second) {
    // Every counter is updated unconditionally, by a selected VALUE. Written as `if (large) ct.a += .. else ct.b += ..` the two
    // branches were merged into one add through a selected ADDRESS, and the counters then lived in scratch memory (12 bytes per
    // lane in every chain kernel: VERDICT r03 #8).
    const bool bold = !large;
    ct.large_acc1l += (large ? 1u : 0u) + ((large && acc1) ? 1u << 16 : 0u);
    ct.acc1b_secl += ((bold && acc1) ? 1u : 0u) + ((large && do_second) ? 1u << 16 : 0u);
    ct.secb_acc2l += ((bold && do_second) ? 1u : 0u) + ((large && acc2) ? 1u << 16 : 0u);
    ct.acc2b_rev += (bold && acc2) ? 1u : 0u;
}

// Acceptance map (:693-709). The reference swaps the accepted proposal into `current` FIRST and then calls
// splatAcceptanceOnly on `proposed.first` / `proposed.second` -- which, after the swap, own the list that WAS current. The
// mark therefore goes to every splat position of the state being REPLACED: red (1,0,0) when a bold first stage was
// accepted (nothing after a large step), green (0,1,0) when the second stage was. Returns 0 none, 1 red, 2 green; the
// caller puts it at the OLD current state's positions. processMixture draws no map.
enum { AMAP_NONE = 0, AMAP_RED = 1, AMAP_GREEN = 2 };
DEV int mh_amap_mark(bool mix, bool amap, bool large, bool acc1, bool acc2) {
    if (!amap || mix || !(acc1 || acc2)) return AMAP_NONE;
    return acc1 ? (large ? AMAP_NONE : AMAP_RED) : AMAP_GREEN;
}
DEV f3 mh_amap_colour(int mark) { return mark == AMAP_RED ? mk3(1.f, 0.f, 0.f) : mk3(0.f, 1.f, 0.f); }
