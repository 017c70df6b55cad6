// Device-side data layout of the DRMLT hot path (gfx950). Plain structs shared by the
// host-side C-ABI (drmlt_capi.cpp) and the kernels (kernels.hip).
//
// HBM layout (all fp32 unless noted), n = chains on this GPU, D = PSS dims kept per chain:
//   x        [D][n]   current PSS state, SoA: lane-contiguous 256 B rows per dimension
//   cur_*    [n]      current state's luminance, splat position, normalised RGB
//   film     [H][W][3] accumulation film, float atomics (ImageBlock semantics, border dropped)
//   prims    [P] x 64 B  intersection records (wave-uniform scalar loads in the brute-force loop)
//   shade    [P] x 64 B  shading records (per-lane gathers after a hit)
//   bvh      [N] x 64 B  2-wide BVH nodes (both child boxes in one 64 B line)
#pragma once
#include <stdint.h>

#define DRMLT_MAX_LDS_PRIMS 96
#define BDPT_MAX_DEPTH 24 // technique=bdpt: one wave's LDS rows (samplers + densities, device_bdpt.h) reach the 64 KB a workgroup may have at maxDepth 26; the flag words would hold 31

// PRIM_QUAD2: two triangles (a,b,c), (a,c,d) that form a parallelogram, intersected once; the hit is
// attributed to the sub-triangle it falls in (two consecutive shading records), so f(u) is unchanged
enum { PRIM_TRIANGLE = 0, PRIM_RECTANGLE = 1, PRIM_SPHERE = 2, PRIM_QUAD2 = 3 };

// World -> primitive space affine map (rows), so that one transform serves all three
// primitive kinds: triangle -> barycentric (u,v,w); rectangle -> Mitsuba object space
// ([-1,1]^2, z=0); sphere -> unit sphere. The ray parameter t is preserved by an affine map.
struct DPrim {
    float m[12];
    int32_t type;
    int32_t shade;      // index of the (first) shading record of this primitive
    int32_t kind_shade; // type | shade << 8 (what the scalar loop loads)
    int32_t pad;
};

// The same intersection record laid out for the brute-force loop of flat-primitive scenes: rows 0 and 1 of the
// affine map interleaved column by column, so that each (u, v) column is one aligned SGPR pair feeding a packed
// fp32 FMA directly (the row-major record needs six scalar moves per primitive to build those pairs).
struct DPrimFlat {
    float c[8];        // (m0, m4), (m1, m5), (m2, m6), (m3, m7)
    float rz[4];       // row 2: m8 .. m11
    int32_t kind_shade;
    int32_t pad[3];
};

// A cuboid (box_merge.h): world -> cuboid coordinates b in [0, 1]^3, rows laid out as DPrimFlat's, and one word per axis with
// the two faces in the planes b[axis] = 0 (low half) and 1 (high half): exists | code << 1 | kind << 4 | shade << 6
// (code: how the face's own (u, v) follow from the in-face coordinates; kind: PRIM_RECTANGLE / PRIM_QUAD2; shade < 1024).
struct DPrimBox {
    float c[8];
    float rz[4];
    uint32_t fw[3];
    int32_t pad;
};

struct DShade {
    float origin[3]; // tri: p0; rect: centre; sphere: centre
    float eu[3];     // tri: p1-p0; rect: objectToWorld column 0; sphere: eu[0] = radius
    float ev[3];     // tri: p2-p0; rect: objectToWorld column 1
    float n[3];      // unit geometric (= shading) normal of flat primitives
    float inv_len_eu;
    int32_t bsdf;
    int32_t emitter; // -1: none
    float inv_area;
};

struct DBsdf {
    int32_t type;
    float rgb[3];
    float p[8]; // dielectric: p[0] = eta (int/ext), p[1] = 1/eta
};

struct DEmitter {
    float radiance[3];
    int32_t prim;
    float cdf_lo, cdf_hi; // DiscreteDistribution entries (normalised)
    float pad[2];
};

#define BVH_SPILL 12 // entries moved to / from the overflow area at a time (trees deeper than BVH_STACK / 3 levels)
#define BVH_STACK 24 // per-lane traversal stack entries (LDS): a 4-wide node pushes at most 3, so the 4-wide depth is bounded by 8 (bvh_build.h)
// 4-wide node, 128 B = one cache line: the four child boxes component by component (each row one 16 B load), then the
// children. child >= 0: inner node index; child < 0: leaf, ~child = (first primitive slot << shift) | count, with
// shift = DParams::bvh_leaf_shift (0 when every leaf holds exactly one primitive: then ~child is the slot itself);
// empty slot: a point box at +FLT_MAX (no ray enters it).
struct DBvh4Node {
    float lox[4], hix[4], loy[4], hiy[4], loz[4], hiz[4];
    int32_t child[4];
    int32_t pad[4];
};
// binary SAH tree the 4-wide one is collapsed from (host only)
struct DBvhNode {      // 64 B
    float lo0[3], hi0[3]; // child 0 box
    float lo1[3], hi1[3]; // child 1 box
    int32_t c0, c1;       // >= 0: inner node index; < 0: leaf, ~c = first prim slot
    int32_t n0, n1;       // leaf primitive counts (0 for inner children)
};

struct DParams {
    // scene
    const DPrim *prims;
    const DShade *shade;
    const DBsdf *bsdfs;
    const DEmitter *emitters;
    const DBvh4Node *bvh; // 4-wide BVH (scenes above the brute-force threshold)
    const float *filter_lut; // 32 entries (MTS_FILTER_RESOLUTION + 1)
    int32_t n_prims, n_emitters, n_bvh_nodes, use_bvh, n_bsdfs, tables_in_lds;
    int32_t n_shade; // shading records (>= n_prims: a merged pair has two)
    // sensor + film
    float cam[12]; // camera-to-world rows (3x4)
    float tan_half_fov, inv_aspect, near_clip, far_clip;
    int32_t width, height;
    float filter_radius, filter_scale, box_weight; // box_weight: the (constant) table value of the box filter
    float *film;
    // configuration
    int32_t type, max_depth, rr_depth, exclude_direct;
    int32_t acceptance_map, timid_after_large, use_mixture, max_dim, eff_dim;
    float p_large, sigma2; // sigma2 = scaleSecond * sigma
    // rng
    uint32_t key0, key1, chain_offset, boot_stream;
    // chains
    uint32_t n_chains;
    float *x;
    float *cur_lum, *cur_px, *cur_py, *cur_r, *cur_g, *cur_b;
    unsigned long long *stats; // 18 counters, layout of drmlt_stats
    int32_t *error_flag;
    int32_t debug;          // DRMLT_DEBUG bit mask (diagnostics only)
    int32_t kernel_variant; // 1: k_mutate (nested loops), 2: k_mutate_v2 (lane state machines), 3: k_mutate_v3 (2 lanes per chain), 4: k_mutate_v4 (free-running, flattened bookkeeping; default)
    int32_t features;       // bit 0 rough conductor, bit 1 dielectric, bit 2 spheres, bit 3 BVH traversal needed
    int32_t mh_batch;       // k_mutate_v2: parked lanes needed before the bookkeeping branch is taken
    // technique=mmlt (device_bidir.h)
    int32_t technique;        // DRMLT_TECH_*
    int32_t light_image;      // "lightImage"
    int32_t fix_emitter_path; // "fixEmitterPath"
    int32_t mmlt_S, mmlt_E;   // state rows of the sensor / emitter segments: 2 (maxDepth + 1), 2 maxDepth
    int32_t mmlt_dmax;        // findMaxDimensions of the deepest chain: draw bases 2 dmax (emitter), 4 dmax (direct)
    int32_t bd_Dd;            // technique=bdpt, directSampling=true: state rows of the direct sampler, 2 (2 maxDepth - 1); else 0
    int32_t *chain_depth;     // [n] path depth of each chain (fixed by its seed)
    // k_mutate_v4, run-ahead (drmlt_run with more than one launch): per-chain count of decided mutations, the launch's counter
    // of waves that still have a chain under the launch's target, and the count no chain may exceed (the render's total)
    uint32_t *chain_done;     // [n] or NULL (then every chain runs exactly n_mut mutations from mutation index mut_base)
    uint32_t *waves_left;
    uint32_t run_limit;
    const uint32_t *exec_order; // technique=mmlt: chain run by lane `slot` of the grid, deepest chains first (rounded up to whole waves, padded with n), or NULL
    int32_t *cur_t;           // [n] sensor-subpath length t of the current state (light tracing: t == 1)
    const float *importance;  // [H][W] two-stage MLT luminance image, or NULL (pathsampler.cpp:1001-1020)
    // technique=bdpt (device_bdpt.h)
    float *bd_verts;          // [BV_FIELDS][2 maxDepth + 1][n_chains_alloc] stored subpath vertices
    float *bd_lists;          // [3][BL rows][n_chains_alloc] splat lists: 0 current, 1 first-stage, 2 second-stage proposal
    uint32_t n_chains_alloc;  // column stride of the two buffers above
    // flat-primitive fast path of the brute-force ray loop (device_path.h: trace_flat)
    const DPrimFlat *prims_flat; // n_flat records + 2 sentinels that no ray can hit, or NULL (BVH scenes)
    int32_t has_plain_tri;       // any PRIM_TRIANGLE record (needs the u + v <= 1 test)
    // algo=pssmlt (kernels.hip: k_mutate_pssmlt)
    int32_t kelemen_weights, kelemen_mutation; // "kelemenStyleWeights" (pssmlt_proc.cpp:197-203), Kelemen (1) or Gaussian (0) mutation
    float pss_sigma, luminance_b;              // Gaussian mutation size; b of the Kelemen weights
    int32_t n_flat;              // records [0, n_flat) of `prims` are flat (and mirrored in prims_flat), [n_flat, n_prims) are spheres
    int32_t bvh_leaf_shift;      // 0: one primitive per leaf, ~child = slot; 3: ~child = slot << 3 | count
    int32_t bvh_stack16;         // every stack entry fits a short: k_mutate_v4 runs its 16-bit-stack variant
    int32_t trace_yield;         // k_mutate_v4 on BVH scenes: a traversal slice ends once this many lanes have finished their ray
    int32_t *bvh_overflow;       // [entry][lane of the grid]: where a traversal stack that outgrows its LDS column puts its oldest entries, or NULL
    uint32_t bvh_ovf_lanes;      // column count of that area (>= lanes of the launch)
    int32_t trace_vote;          // traversal: the wave tests nodes when 16 * (lanes at a leaf) <= trace_vote * (lanes at a node), leaves otherwise
    int32_t pool_refill;         // k_mutate_v5: idle lanes take pending rays off the queue inside a trace phase once this many lanes have run dry
    const DPrimBox *prims_box;   // cuboid records of the brute-force loop (n_box of them + one sentinel), tested before prims_flat
    int32_t n_box;
    int32_t n_flat_rec;          // records in prims_flat (flat records that are no cuboid's face) -- n_flat counts the flat records of `prims`
    int32_t small_tables_lds;    // k_mutate_v5 on traversed scenes: BSDF / emitter records and the emitters' shape records are staged in LDS (they fit beside the pool)
    int32_t pad_tables;
    float *rows;                 // k_mutate_v5 with its proposal rows in device memory ([dim][chain], as x), or NULL: rows in LDS
    int32_t boot_weighted;       // bootstrap kernels: also write each sample's luminance under the importance map, to lum_out[n + i] (two-stage MLT: seeds drawn from the chains' own target, drmlt_capi.cpp)
};

// result of one PSS evaluation, SoA-friendly
struct DSplat {
    float lum, px, py, r, g, b;
};
