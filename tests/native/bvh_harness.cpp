// CPU check of the host-side BVH builder (drmlt-mitsuba_amd/csrc/bvh_build.h): structure invariants and a
// closest-box traversal compared with brute force. Built and run by tests/test_bvh_build.py (g++, no GPU).
//   bvh_harness <case> <n> <max_depth>     case: soup | chain | coincident
#include "bvh_build.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

static bool box_hit(const float *lo, const float *hi, const float *o, const float *inv, float tmax) {
    float t0 = 0.f, t1 = tmax;
    for (int k = 0; k < 3; ++k) {
        float a = (lo[k] - o[k]) * inv[k], b = (hi[k] - o[k]) * inv[k];
        if (a > b) std::swap(a, b);
        t0 = std::max(t0, a); t1 = std::min(t1, b);
    }
    return t0 <= t1;
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const std::string kind = argv[1];
    const int n = atoi(argv[2]), max_depth = atoi(argv[3]);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<PrimBounds> pb(n);
    for (int i = 0; i < n; ++i) {
        float c[3], e;
        if (kind == "soup") { for (float &v : c) v = U(rng); e = 0.05f; }
        else if (kind == "chain") { const float s = std::pow(0.5f, 0.3f * i); c[0] = 0.9f * s; c[1] = c[2] = 0.f; e = 0.1f * s; }
        else { c[0] = c[1] = c[2] = 0.25f; e = 0.01f * (1 + i % 3); }
        for (int k = 0; k < 3; ++k) { pb[i].lo[k] = c[k] - e; pb[i].hi[k] = c[k] + e; }
    }
    std::vector<DBvhNode> nodes;
    std::vector<int> order;
    const int medians = build_bvh(pb, nodes, order, max_depth);

    // ---- structure: every primitive in exactly one leaf; child boxes bound their subtrees; depth within the bound
    std::vector<int> seen(n, 0);
    int depth = 0, leaves = 0;
    struct Item { int node, d; };
    std::vector<Item> todo{{0, 1}};
    bool ok = (int) order.size() == n;
    auto check_leaf = [&](int c, int cnt, const float *lo, const float *hi) {
        const int first = ~c;
        leaves++;
        for (int i = 0; i < cnt; ++i) {
            const int p = order[first + i];
            seen[p]++;
            for (int k = 0; k < 3; ++k) ok = ok && pb[p].lo[k] >= lo[k] && pb[p].hi[k] <= hi[k];
        }
    };
    std::vector<bvh_detail::Box> subtree(nodes.size());
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        depth = std::max(depth, it.d);
        const DBvhNode &N = nodes[it.node];
        if (N.c0 < 0) check_leaf(N.c0, N.n0, N.lo0, N.hi0); else todo.push_back({N.c0, it.d + 1});
        if (N.c1 < 0) check_leaf(N.c1, N.n1, N.lo1, N.hi1); else todo.push_back({N.c1, it.d + 1});
    }
    for (int i = 0; i < n; ++i) ok = ok && seen[i] == 1;
    // inner children: the stored child box must contain the child's own two boxes
    for (const DBvhNode &N : nodes)
        for (int side = 0; side < 2; ++side) {
            const int c = side ? N.c1 : N.c0;
            if (c < 0) continue;
            const float *lo = side ? N.lo1 : N.lo0, *hi = side ? N.hi1 : N.hi0;
            const DBvhNode &C = nodes[c];
            for (int k = 0; k < 3; ++k) {
                ok = ok && std::min(C.lo0[k], C.n1 || C.c1 >= 0 ? C.lo1[k] : C.lo0[k]) >= lo[k];
                ok = ok && std::max(C.hi0[k], C.n1 || C.c1 >= 0 ? C.hi1[k] : C.hi0[k]) <= hi[k];
            }
        }

    // ---- the 4-wide tree the kernels traverse: same checks (every primitive in exactly one leaf, boxes bound subtrees),
    // the 4-wide tree: its depth (3 * depth4 entries bound the traversal stack), and the same queries
    std::vector<DBvh4Node> n4;
    int shift = 0;
    const int depth4 = build_bvh4(nodes, n4, &shift);
    std::vector<int> seen4(n, 0);
    int leaves4 = 0, measured_depth4 = 0;
    {
        struct It4 { int node, d; };
        std::vector<It4> todo4{{0, 1}};
        while (!todo4.empty()) {
            const It4 it = todo4.back();
            todo4.pop_back();
            measured_depth4 = std::max(measured_depth4, it.d);
            const DBvh4Node &N = n4[it.node];
            for (int c = 0; c < 4; ++c) {
                const float lo[3] = {N.lox[c], N.loy[c], N.loz[c]}, hi[3] = {N.hix[c], N.hiy[c], N.hiz[c]};
                if (N.child[c] >= 0) {
                    todo4.push_back({N.child[c], it.d + 1});
                    const DBvh4Node &C = n4[N.child[c]];
                    for (int q = 0; q < 4; ++q) {
                        if (C.lox[q] == FLT_MAX) continue; // empty slot
                        ok = ok && C.lox[q] >= lo[0] && C.loy[q] >= lo[1] && C.loz[q] >= lo[2] && C.hix[q] <= hi[0] && C.hiy[q] <= hi[1] && C.hiz[q] <= hi[2];
                    }
                } else {
                    if (lo[0] == FLT_MAX) continue; // empty slot: point box at +FLT_MAX
                    const int first = ~N.child[c] >> shift, cnt = shift ? (~N.child[c] & 7) : 1;
                    leaves4++;
                    ok = ok && cnt <= 4;
                    for (int i = 0; i < cnt; ++i) {
                        const int p = order[first + i];
                        seen4[p]++;
                        for (int k = 0; k < 3; ++k) ok = ok && pb[p].lo[k] >= lo[k] && pb[p].hi[k] <= hi[k];
                    }
                }
            }
        }
    }
    for (int i = 0; i < n; ++i) ok = ok && seen4[i] == 1;
    ok = ok && measured_depth4 == depth4 && depth4 <= depth; // (a stack deeper than BVH_STACK spills to memory on the device)

    // ---- queries: the set of primitive boxes a ray can hit, through both trees and by brute force
    int mismatches = 0, max_stack = 0, max_stack4 = 0;
    for (int q = 0; q < 2000; ++q) {
        float o[3] = {U(rng) * 2.f, U(rng) * 2.f, U(rng) * 2.f}, d[3] = {U(rng), U(rng), U(rng)}, inv[3];
        for (int k = 0; k < 3; ++k) inv[k] = 1.f / d[k];
        long brute = 0, tree = 0;
        for (int i = 0; i < n; ++i) if (box_hit(pb[i].lo, pb[i].hi, o, inv, 1e30f)) brute += i + 1;
        std::vector<int> stack{0};
        while (!stack.empty()) {
            max_stack = std::max(max_stack, (int) stack.size());
            const DBvhNode &N = nodes[stack.back()];
            stack.pop_back();
            for (int side = 0; side < 2; ++side) {
                const int c = side ? N.c1 : N.c0, cnt = side ? N.n1 : N.n0;
                if (!box_hit(side ? N.lo1 : N.lo0, side ? N.hi1 : N.hi0, o, inv, 1e30f)) continue;
                if (c >= 0) { stack.push_back(c); continue; }
                for (int i = 0; i < cnt; ++i) { const int p = order[~c + i]; if (box_hit(pb[p].lo, pb[p].hi, o, inv, 1e30f)) tree += p + 1; }
            }
        }
        long tree4 = 0;
        std::vector<int> st4{0};
        size_t max4 = 0;
        while (!st4.empty()) {
            max4 = std::max(max4, st4.size());
            const DBvh4Node &N = n4[st4.back()];
            st4.pop_back();
            for (int c = 0; c < 4; ++c) {
                const float lo[3] = {N.lox[c], N.loy[c], N.loz[c]}, hi[3] = {N.hix[c], N.hiy[c], N.hiz[c]};
                if (!box_hit(lo, hi, o, inv, 1e30f)) continue; // empty slots must fail this test by themselves
                if (N.child[c] >= 0) { st4.push_back(N.child[c]); continue; }
                const int first = ~N.child[c] >> shift, cnt = shift ? (~N.child[c] & 7) : 1;
                for (int i = 0; i < cnt; ++i) { const int p = order[first + i]; if (box_hit(pb[p].lo, pb[p].hi, o, inv, 1e30f)) tree4 += p + 1; }
            }
        }
        max_stack4 = std::max(max_stack4, (int) max4);
        mismatches += brute != tree;
        mismatches += brute != tree4;
    }
    printf("{\"ok\": %s, \"n\": %d, \"nodes\": %zu, \"leaves\": %d, \"depth\": %d, \"median_splits\": %d, \"mismatches\": %d, \"stack\": %d, "
           "\"nodes4\": %zu, \"leaves4\": %d, \"depth4\": %d}\n",
           ok ? "true" : "false", n, nodes.size(), leaves, depth, medians, mismatches, BVH_STACK, n4.size(), leaves4, depth4);
    return 0;
}
