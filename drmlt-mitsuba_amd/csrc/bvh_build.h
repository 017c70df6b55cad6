// Host-side builder of the BVH the kernels traverse: a binary tree by binned SAH, collapsed into 4-wide nodes.
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

constexpr int kLeafMax = 1; // primitives per leaf (DRMLT_BVH_LEAF: 1..4). Measured on the 2000-triangle soup: 1 -> 1.12e8, 2 -> 0.97e8 mutations/s (no leaf loop to diverge in)
constexpr int kLeafCap = 4; // DRMLT_BVH_LEAF may raise it to this (the leaf reference keeps 3 bits for the count)
constexpr int kBins = 16;

// inner levels of a subtree over n primitives built by median splits
inline int median_depth(int n, int leaf_max) {
    int d = 0;
    while (n > leaf_max) { n = (n + 1) / 2; d++; }
    return d;
}

constexpr int kMaxBinaryDepth = 64; // recursion bound of the builder; the traversal stack spills to memory beyond BVH_STACK / 3 four-wide levels

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
    b.max_depth = std::max(std::min(max_depth, kMaxBinaryDepth), median_depth(n, b.leaf_max)); // never below what a balanced tree needs
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


// ---- 4-wide tree: every node takes the two children of a binary node and keeps replacing one inner child by its two
// children until it has four (or only leaves are left). `by_height`: expand the child with the tallest subtree first
// (then every root-to-leaf path loses at least two binary levels per 4-wide node: depth4 <= ceil(depth2 / 2));
// otherwise the child with the largest box (better culling). Returns the 4-wide depth (inner nodes on the longest path).
namespace bvh_detail {
struct Ref4 {
    int c, n; // c >= 0: binary inner node; c < 0: leaf ~first with n primitives
    float lo[3], hi[3];
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
};
inline void child_refs(const DBvhNode &N, Ref4 &a, Ref4 &b) {
    a.c = N.c0; a.n = N.n0; b.c = N.c1; b.n = N.n1;
    for (int k = 0; k < 3; ++k) { a.lo[k] = N.lo0[k]; a.hi[k] = N.hi0[k]; b.lo[k] = N.lo1[k]; b.hi[k] = N.hi1[k]; }
}
inline int height2(const std::vector<DBvhNode> &bin, int node, std::vector<int> &memo) {
    if (node < 0) return 0;
    if (memo[node]) return memo[node];
    return memo[node] = 1 + std::max(height2(bin, bin[node].c0, memo), height2(bin, bin[node].c1, memo));
}
inline int collapse4(const std::vector<DBvhNode> &bin, int node, std::vector<DBvh4Node> &out, bool by_height, std::vector<int> &memo, int leaf_shift) {
    std::vector<Ref4> refs(2);
    child_refs(bin[node], refs[0], refs[1]);
    while (refs.size() < 4) {
        int best = -1;
        float key = -1.f;
        for (size_t i = 0; i < refs.size(); ++i) {
            if (refs[i].c < 0) continue;
            const float k = by_height ? (float) height2(bin, refs[i].c, memo) + 1e-3f * std::min(refs[i].area(), 100.f) : refs[i].area();
            if (k > key) { key = k; best = (int) i; }
        }
        if (best < 0) break;
        Ref4 a, b;
        child_refs(bin[refs[best].c], a, b);
        refs[best] = a;
        refs.push_back(b);
    }
    const int self = (int) out.size();
    out.emplace_back();
    DBvh4Node N4{};
    int depth = 1;
    for (int i = 0; i < 4; ++i) {
        const bool used = i < (int) refs.size() && !(refs[i].c < 0 && refs[i].n == 0);
        N4.pad[i] = 0;
        if (!used) {
            // a point at +FLT_MAX on every axis: on an axis the ray moves along, both slab distances are the same infinity, so
            // near > far whatever the direction. (An INVERTED box does not do it: the slab test takes min / max of the two plane
            // distances and would see the whole line.)
            N4.lox[i] = N4.loy[i] = N4.loz[i] = FLT_MAX; N4.hix[i] = N4.hiy[i] = N4.hiz[i] = FLT_MAX;
            N4.child[i] = ~0; // never read: no ray enters the box
            continue;
        }
        const Ref4 &r = refs[i];
        N4.lox[i] = r.lo[0]; N4.hix[i] = r.hi[0]; N4.loy[i] = r.lo[1]; N4.hiy[i] = r.hi[1]; N4.loz[i] = r.lo[2]; N4.hiz[i] = r.hi[2];
        if (r.c >= 0) {
            N4.child[i] = (int) out.size(); // the node the recursive call is about to append
            depth = std::max(depth, 1 + collapse4(bin, r.c, out, by_height, memo, leaf_shift));
        } else {
            N4.child[i] = leaf_shift ? ~(((~r.c) << leaf_shift) | r.n) : ~(~r.c);
        }
    }
    out[self] = N4;
    return depth;
}
} // namespace bvh_detail

// nodes4[0] is the root. The traversal pushes at most 3 entries per inner node on the current path: 3 * depth4 entries
// bound its stack; what does not fit the LDS column (BVH_STACK) spills to memory (device_path.h: trav_run).
// `leaf_shift` (out): 0 when every leaf holds one primitive, else 3 (see DBvh4Node).
inline int build_bvh4(const std::vector<DBvhNode> &bin, std::vector<DBvh4Node> &nodes4, int *leaf_shift_out = nullptr) {
    std::vector<int> memo(bin.size(), 0);
    nodes4.clear();
    bool single = true;
    for (const DBvhNode &N : bin) single = single && (N.c0 >= 0 || N.n0 <= 1) && (N.c1 >= 0 || N.n1 <= 1);
    const int leaf_shift = single ? 0 : 3;
    if (leaf_shift_out) *leaf_shift_out = leaf_shift;
    return bvh_detail::collapse4(bin, 0, nodes4, false, memo, leaf_shift);
}
