// Host-side builder of the 2-wide BVH the kernels traverse (binned SAH, one primitive per leaf by default).
// The reference uses a SAH kd-tree (src/librender/skdtree.cpp, sahkdtree3.h); a closest-hit
// query returns the same primitive through either structure, so the builder is free to pick
// the layout that suits the GPU: one 64 B node = both child boxes, fetched as one line.
#pragma once
#include "device_types.h"

#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <vector>

struct PrimBounds {
    float lo[3], hi[3];
};

namespace bvh_detail {

struct Box {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const PrimBounds &b) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
    }
    void grow(const Box &b) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
    }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.f;
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
};

constexpr int kLeafMax = 1; // measured on the 2000-triangle soup: 1 -> 2.08e8, 2 or 3 -> 1.84e8, 4 -> 1.80e8 mutations/s (no leaf loop to diverge in)
constexpr int kLeafCap = 4; // DRMLT_BVH_LEAF may raise it to this (the leaf reference keeps 3 bits for the count)
constexpr int kBins = 16;

// inner levels of a subtree over n primitives built by median splits
inline int median_depth(int n, int leaf_max) {
    int d = 0;
    while (n > leaf_max) { n = (n + 1) / 2; d++; }
    return d;
}

struct Builder {
    const std::vector<PrimBounds> &pb;
    std::vector<int> &order; // primitive indices, leaf ranges are contiguous
    std::vector<DBvhNode> &nodes;
    int max_depth = BVH_STACK; // inner nodes on any root-to-leaf path: the traversal pushes at most one entry per level
    int median_splits = 0;
    int leaf_max = kLeafMax;

    Box range_box(int first, int count) const {
        Box b;
        for (int i = 0; i < count; ++i) b.grow(pb[order[first + i]]);
        return b;
    }

    // returns split position (number of prims going left), reorders order[first, first+count)
    int split(int first, int count, const Box &bounds) {
        Box cb;
        for (int i = 0; i < count; ++i) {
            const PrimBounds &p = pb[order[first + i]];
            for (int k = 0; k < 3; ++k) { float c = 0.5f * (p.lo[k] + p.hi[k]); cb.lo[k] = std::min(cb.lo[k], c); cb.hi[k] = std::max(cb.hi[k], c); }
        }
        int bestAxis = -1, bestBin = -1;
        float bestCost = FLT_MAX;
        for (int axis = 0; axis < 3; ++axis) {
            float ext = cb.hi[axis] - cb.lo[axis];
            if (!(ext > 0.f)) continue;
            Box bins[kBins];
            int cnt[kBins] = {0};
            float scale = kBins / ext;
            for (int i = 0; i < count; ++i) {
                const PrimBounds &p = pb[order[first + i]];
                int b = std::min(kBins - 1, (int) ((0.5f * (p.lo[axis] + p.hi[axis]) - cb.lo[axis]) * scale));
                bins[b].grow(p);
                cnt[b]++;
            }
            float rightArea[kBins];
            int rightCnt[kBins];
            Box acc;
            int c = 0;
            for (int b = kBins - 1; b > 0; --b) { acc.grow(bins[b]); c += cnt[b]; rightArea[b] = acc.area(); rightCnt[b] = c; }
            Box left;
            int lc = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                left.grow(bins[b]);
                lc += cnt[b];
                if (lc == 0 || rightCnt[b + 1] == 0) continue;
                float cost = left.area() * lc + rightArea[b + 1] * rightCnt[b + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = b; }
            }
        }
        if (bestAxis < 0) return count / 2; // all centroids coincide: split in the middle
        float ext = cb.hi[bestAxis] - cb.lo[bestAxis], scale = kBins / ext;
        auto mid = std::partition(order.begin() + first, order.begin() + first + count, [&](int idx) {
            const PrimBounds &p = pb[idx];
            int b = std::min(kBins - 1, (int) ((0.5f * (p.lo[bestAxis] + p.hi[bestAxis]) - cb.lo[bestAxis]) * scale));
            return b <= bestBin;
        });
        int nl = (int) (mid - (order.begin() + first));
        if (nl == 0 || nl == count) return count / 2;
        (void) bounds;
        return nl;
    }

    // median split along the widest centroid axis: bounds the depth where SAH would build a degenerate chain
    int split_median(int first, int count) {
        Box cb;
        auto centre = [&](int idx, int k) { return 0.5f * (pb[idx].lo[k] + pb[idx].hi[k]); };
        for (int i = 0; i < count; ++i)
            for (int k = 0; k < 3; ++k) { float c = centre(order[first + i], k); cb.lo[k] = std::min(cb.lo[k], c); cb.hi[k] = std::max(cb.hi[k], c); }
        int axis = 0;
        for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
        const int nl = (count + 1) / 2;
        std::nth_element(order.begin() + first, order.begin() + first + nl, order.begin() + first + count,
                         [&](int a, int b) { return centre(a, axis) < centre(b, axis); });
        return nl;
    }

    // builds the subtree over order[first, first+count) (count > kLeafMax) and returns its node index
    int build(int first, int count, int depth = 0) {
        int self = (int) nodes.size();
        nodes.emplace_back();
        Box bounds = range_box(first, count);
        int nl = split(first, count, bounds);
        if (depth + 1 + median_depth(std::max(nl, count - nl), leaf_max) > max_depth) { nl = split_median(first, count); median_splits++; }
        int nr = count - nl;
        Box bl = range_box(first, nl), br = range_box(first + nl, nr);
        int c0, c1, n0 = 0, n1 = 0;
        if (nl <= leaf_max) { c0 = ~first; n0 = nl; } else { c0 = build(first, nl, depth + 1); }
        if (nr <= leaf_max) { c1 = ~(first + nl); n1 = nr; } else { c1 = build(first + nl, nr, depth + 1); }
        DBvhNode &N = nodes[self];
        for (int k = 0; k < 3; ++k) { N.lo0[k] = bl.lo[k]; N.hi0[k] = bl.hi[k]; N.lo1[k] = br.lo[k]; N.hi1[k] = br.hi[k]; }
        N.c0 = c0; N.c1 = c1; N.n0 = n0; N.n1 = n1;
        return self;
    }
};

} // namespace bvh_detail

// nodes[0] is the root; returns the number of depth-bounded (median) splits. `order[i]` = original index of the primitive stored in slot i.
inline int build_bvh(const std::vector<PrimBounds> &pb, std::vector<DBvhNode> &nodes, std::vector<int> &order, int max_depth = BVH_STACK) {
    using namespace bvh_detail;
    const int n = (int) pb.size();
    order.resize(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    nodes.clear();
    Builder b{pb, order, nodes};
    if (const char *t = getenv("DRMLT_BVH_LEAF")) b.leaf_max = std::max(1, std::min(kLeafCap, atoi(t)));
    b.max_depth = std::max(std::min(max_depth, BVH_STACK), median_depth(n, b.leaf_max)); // never below what a balanced tree needs
    if (n <= b.leaf_max) {
        // single leaf under a root whose second child is empty
        nodes.emplace_back();
        Box box = b.range_box(0, n);
        DBvhNode &N = nodes[0];
        for (int k = 0; k < 3; ++k) { N.lo0[k] = box.lo[k]; N.hi0[k] = box.hi[k]; N.lo1[k] = FLT_MAX; N.hi1[k] = -FLT_MAX; }
        N.c0 = ~0; N.n0 = n; N.c1 = ~0; N.n1 = 0;
        return 0;
    }
    b.build(0, n);
    return b.median_splits; // nodes where the depth bound overrode the SAH split
}
