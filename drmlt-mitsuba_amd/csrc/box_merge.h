// Host side of the brute-force ray loop's CUBOID records (device_path.h: test_box).
//
// Scenes small enough for the wave-uniform loop over intersection records (Cornell class: <= 48 records) spend a third of the
// chain kernels' time in that loop, one parallelogram test per wall or box face. Faces that together bound a parallelepiped --
// Mitsuba's `cube` (12 triangles = 6 merged pairs), or the walls of a room (rectangles; the open side is simply missing) --
// are intersected here as ONE record: the ray is taken into the cuboid's own coordinates (the same affine map a flat record
// uses), clipped against the three slabs, and the face it enters through -- or, for a ray that starts inside, leaves through --
// is the hit, provided the scene HAS that face. A line crosses the boundary of a convex body at most twice, so "the nearest
// existing face crossing at t >= tmin" is exactly what the loop over the separate faces returns; the hit is handed on in the
// face's own (u, v) parametrisation with the face's own record, so shading points, frames, sub-triangles and emitters are those
// of the separate faces. Config 2: 18 records -> 1 + 3 cuboids.
//
// find_boxes works on the world-space parallelograms of the flat records (corner + two edges), pure geometry, no device.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

struct QuadGeo {
    double a[3], e1[3], e2[3]; // corner and the two edges the record's (u, v) run along: p = a + u e1 + v e2, u, v in [0, 1]
    bool usable;               // a parallelogram (rectangle or merged triangle pair); plain triangles / spheres are not
};

// face id = 2 * axis + side: the face in the plane b[axis] = side of the cuboid's own coordinates b in [0, 1]^3
struct BoxGeo {
    double a[3], E[3][3]; // corner and edge vectors: p = a + b0 E[0] + b1 E[1] + b2 E[2]
    int face[6];          // record index of each face, or -1: the scene has no surface there
    int code[6];          // how the face's own (u, v) follow from the in-face coordinates (p, q) = (b[j], b[k]), j < k the
                          // two other axes: bit 0 swap (u runs along k), bit 1 u = 1 - ., bit 2 v = 1 - .
    int n_faces;
};

namespace box_merge_detail {
inline double dot(const double *x, const double *y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; }
inline double len(const double *x) { return std::sqrt(dot(x, x)); }
inline double det3(const double *x, const double *y, const double *z) {
    return x[0] * (y[1] * z[2] - y[2] * z[1]) - x[1] * (y[0] * z[2] - y[2] * z[0]) + x[2] * (y[0] * z[1] - y[1] * z[0]);
}
inline bool close3(const double *x, const double *y, double tol) {
    return std::fabs(x[0] - y[0]) <= tol && std::fabs(x[1] - y[1]) <= tol && std::fabs(x[2] - y[2]) <= tol;
}
// does quad q lie on face (axis, side) of box B with its corner on a box vertex and its edges along the box's? -> code, or -1
inline int match_face(const QuadGeo &q, const BoxGeo &B, int axis, int side, double tol) {
    const int j = axis == 0 ? 1 : 0, k = axis == 2 ? 1 : 2;
    // the quad's corner in box coordinates must be (side on `axis`, 0 or 1 on j and k)
    for (int pj = 0; pj < 2; ++pj)
        for (int pk = 0; pk < 2; ++pk) {
            double c[3];
            for (int d = 0; d < 3; ++d) c[d] = B.a[d] + side * B.E[axis][d] + pj * B.E[j][d] + pk * B.E[k][d];
            if (!close3(c, q.a, tol)) continue;
            // edges: e1 = +-E[j] and e2 = +-E[k] (no swap) or e1 = +-E[k] and e2 = +-E[j] (swap), pointing into the face
            for (int swap = 0; swap < 2; ++swap) {
                const double *Eu = B.E[swap ? k : j], *Ev = B.E[swap ? j : k];
                const int cu = swap ? pk : pj, cv = swap ? pj : pk; // the corner's coordinate along the u / v axis
                double wu[3], wv[3];
                for (int d = 0; d < 3; ++d) { wu[d] = (cu ? -1.0 : 1.0) * Eu[d]; wv[d] = (cv ? -1.0 : 1.0) * Ev[d]; }
                if (close3(wu, q.e1, tol) && close3(wv, q.e2, tol)) return swap | (cu << 1) | (cv << 2);
            }
        }
    return -1;
}
} // namespace box_merge_detail

// Greedy: the parallelepiped that collects the most faces first. Only cuboids with at least `min_faces` faces are worth a
// record of their own (a cuboid test costs about three parallelogram tests). `max_boxes`: what the caller can hold.
inline std::vector<BoxGeo> find_boxes(const std::vector<QuadGeo> &quads, int min_faces = 4, int max_boxes = 16) {
    using namespace box_merge_detail;
    const int n = (int) quads.size();
    std::vector<char> used(n, 0);
    std::vector<BoxGeo> out;
    double scale = 0.0;
    for (const QuadGeo &q : quads) if (q.usable) scale = std::max(scale, std::max(len(q.e1), len(q.e2)));
    const double tol = 1e-5 * scale;
    while ((int) out.size() < max_boxes) {
        BoxGeo best;
        best.n_faces = 0;
        for (int qi = 0; qi < n; ++qi) {
            if (used[qi] || !quads[qi].usable) continue;
            const QuadGeo &Q = quads[qi];
            for (int ri = 0; ri < n; ++ri) {
                if (ri == qi || used[ri] || !quads[ri].usable) continue;
                for (int ev = 0; ev < 4; ++ev) { // the third edge: +- an edge of another face
                    double e3[3];
                    const double *src = (ev & 1) ? quads[ri].e2 : quads[ri].e1;
                    for (int d = 0; d < 3; ++d) e3[d] = (ev & 2) ? -src[d] : src[d];
                    if (std::fabs(det3(Q.e1, Q.e2, e3)) < 1e-6 * len(Q.e1) * len(Q.e2) * len(e3)) continue; // coplanar with Q
                    BoxGeo B; // Q is the face b2 = 0 (the case "Q is b2 = 1" is the same cuboid with e3 negated: ev & 2)
                    for (int d = 0; d < 3; ++d) { B.a[d] = Q.a[d]; B.E[0][d] = Q.e1[d]; B.E[1][d] = Q.e2[d]; B.E[2][d] = e3[d]; }
                    B.n_faces = 0;
                    std::vector<char> taken(n, 0);
                    for (int f = 0; f < 6; ++f) {
                        B.face[f] = -1; B.code[f] = 0;
                        for (int si = 0; si < n; ++si) {
                            if (used[si] || taken[si] || !quads[si].usable) continue;
                            const int c = match_face(quads[si], B, f >> 1, f & 1, tol);
                            if (c >= 0) { B.face[f] = si; B.code[f] = c; B.n_faces++; taken[si] = 1; break; }
                        }
                    }
                    if (B.n_faces > best.n_faces) best = B;
                }
            }
        }
        if (best.n_faces < min_faces) break;
        for (int f = 0; f < 6; ++f) if (best.face[f] >= 0) used[best.face[f]] = 1;
        out.push_back(best);
    }
    return out;
}
