/*
 * drmlt_abi.h -- C-ABI of the MI355X-native DRMLT hot path.
 *
 * This is the drop-in boundary: a thin Mitsuba `Integrator` adaptor (see
 * INTEGRATION.md and drmlt-mitsuba_amd/host/) fills the POD structs below
 * from the public Scene/Sensor/Film accessors and calls these entry points
 * from `DRMLT::render()`. Nothing here uses C++ or torch types.
 *
 * Each entry point cites the reference interface it replaces
 * (paths relative to the reference checkout):
 *
 *   drmlt_create      DRMLT::DRMLT(props) + DRMLTRenderer::prepare
 *                     src/integrators/drmlt/drmlt.cpp:178-351,
 *                     src/integrators/drmlt/drmlt_proc.cpp:84-154
 *   drmlt_seed        PathSampler::generateSeeds + seed replay
 *                     src/libbidir/pathsampler.cpp:859-960,
 *                     src/integrators/drmlt/drmlt.cpp:498-546,
 *                     src/integrators/drmlt/drmlt_proc.cpp:467-514
 *   drmlt_run         DRMLTRenderer::process / processMixture (the chain loop)
 *                     src/integrators/drmlt/drmlt_proc.cpp:161-380,386-771
 *   drmlt_develop     DRMLTProcess::develop
 *                     src/integrators/drmlt/drmlt_proc.cpp:813-854
 *   drmlt_stats       the StatsCounter block
 *                     src/integrators/drmlt/drmlt_proc.cpp:34-49
 *   drmlt_eval_paths  PathSampler::sampleSplats (EUnidirectional; EMMLT:
 *                     src/libbidir/pathsampler.cpp:84-320)
 *                     include/mitsuba/bidir/pathsampler.h:124,
 *                     src/libbidir/pathsampler.cpp:529-567
 *   drmlt_film_*      ImageBlock accumulation  (m_accum)
 *                     src/integrators/drmlt/drmlt_proc.cpp:856-867
 *   drmlt_destroy     ~DRMLT / ref<> release
 *
 * The same structs are consumed by the CPU oracle (oracle/, test
 * infrastructure only) so that parity tests feed both sides identical
 * inputs.
 */
#ifndef DRMLT_ABI_H
#define DRMLT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRMLT_ABI_VERSION 4

/* ---- enums (values are ABI) ------------------------------------------- */

/* PathSampler::ETechnique, include/mitsuba/bidir/pathsampler.h */
enum { DRMLT_TECH_PATH = 0, DRMLT_TECH_BDPT = 1, DRMLT_TECH_MMLT = 2 };

/* DRMLTConfiguration::EType, src/integrators/drmlt/drmlt.h:70-74 */
enum { DRMLT_TYPE_GREEN = 0, DRMLT_TYPE_MIRA = 1, DRMLT_TYPE_ORBITAL = 2 };

/* which chain loop: drmlt_proc.cpp (0) or pssmlt_proc.cpp (1, oracle only) */
enum { DRMLT_ALGO_DRMLT = 0, DRMLT_ALGO_PSSMLT = 1 };

enum { DRMLT_SHAPE_TRIANGLE = 0, DRMLT_SHAPE_RECTANGLE = 1, DRMLT_SHAPE_SPHERE = 2 };

enum {
    DRMLT_BSDF_DIFFUSE = 0,        /* src/bsdfs/diffuse.cpp        */
    DRMLT_BSDF_DIELECTRIC = 1,     /* src/bsdfs/dielectric.cpp     */
    DRMLT_BSDF_ROUGHCONDUCTOR = 2  /* src/bsdfs/roughconductor.cpp */
};

enum { DRMLT_EMITTER_AREA = 0 };   /* src/emitters/area.cpp */

enum { DRMLT_FILTER_BOX = 0, DRMLT_FILTER_GAUSSIAN = 1 };

/* Two-stage MLT: what the chain seeds are resampled in proportion to (drmlt_config.seed_rule; adaptor property
 * "firstStageSeeding" = "target" | "reference").
 *   TARGET    (default) the luminance of f / importance -- the chains' own target (they sample the splat list AFTER
 *             SplatList::normalize(importanceMap), pathsampler.cpp:1001-1020), so every chain starts in its stationary
 *             distribution. Upstream Mitsuba's rule; a device's tens of thousands of short chains need it (dev. 19).
 *   REFERENCE the luminance of f itself: this fork's PathSampler::generateSeeds takes it BEFORE the division
 *             (pathsampler.cpp:901-905). Harmless over the reference's 1e5-mutation work units, a start-up bias over short chains.
 * Without an importance map the two rules are the same rule. b is the mean of f under both. */
enum { DRMLT_SEED_TARGET = 0, DRMLT_SEED_REFERENCE = 1 };

/* workUnits = -1 ("derived", drmlt_config.work_units_rule; adaptor property "workUnitsRule" = "device" | "reference").
 *   DEVICE    (default) a device-filling chain count: 196 608 (path: three waves per SIMD of its pool kernel), 131 072 (bdpt), 262 144 (mmlt; 1 048 576 from 2^35 mutations up), pssmlt 65 536; at least 64
 *             mutations per chain.
 *   REFERENCE ceil(budget / 200 000) (path) or / 100 000 (bdpt, mmlt): drmlt.cpp:434-444 -- sized for a CPU scheduler. */
enum { DRMLT_WORK_UNITS_DEVICE = 0, DRMLT_WORK_UNITS_REFERENCE = 1 };

/* error codes (0 = ok, negative = failure; message in the err buffer /
 * drmlt_last_error) */
enum {
    DRMLT_OK = 0,
    DRMLT_E_INVALID = -1,     /* bad argument / unsupported configuration     */
    DRMLT_E_DEVICE = -2,      /* HIP error                                    */
    DRMLT_E_STATE = -3,       /* call order violated (e.g. run before seed)   */
    DRMLT_E_ZERO_LUM = -4,    /* "average image luminance appears to be zero" */
    DRMLT_E_REPLAY = -5,      /* seed replay luminance mismatch               */
    DRMLT_E_CANCELLED = -6    /* *stop became non-zero                        */
};

/* ---- configuration: names/defaults follow drmlt.cpp:193-349 ------------ */

typedef struct drmlt_config {
    uint32_t struct_size;        /* = sizeof(drmlt_config)                         */
    int32_t  algo;               /* DRMLT_ALGO_*                                   */
    int32_t  technique;          /* "technique"        (required)                  */
    int32_t  type;               /* "type"             (required)                  */
    int32_t  max_depth;          /* "maxDepth"         default -1 (path: finite)   */
    int32_t  rr_depth;           /* "rrDepth"          default 5                   */
    int32_t  direct_samples;     /* "directSamples"    default 16; <0: MLT does it */
    int32_t  luminance_samples;  /* "luminanceSamples" default 100000              */
    int32_t  work_units;         /* "workUnits"        default -1: derived -- a device-filling chain count (see DRMLT_WORK_UNITS_DEVICE),
                                  * at least 64 mutations per chain; the reference derives ~budget / 200 000 */
    int32_t  sample_count;       /* sensor sampler's sampleCount = mutations/pixel */
    float    p_large;            /* "pLarge"           default 0.3                 */
    float    sigma;              /* "sigma"            default 1/64                */
    float    scale_second;       /* "scaleSecond"      default 0.1 (error if > 1)  */
    float    average_luminance;  /* "averageLuminance" default -1                  */
    int32_t  acceptance_map;     /* "acceptanceMap"    default 0                   */
    int32_t  timid_after_large;  /* "timidAfterLarge"  default 0                   */
    int32_t  fix_emitter_path;   /* "fixEmitterPath"   default 0 (mmlt only)       */
    int32_t  use_mixture;        /* "useMixture"       default 0                   */
    int32_t  kelemen_style_weights;  /* pssmlt: Kelemen weights (default 1)        */
    int32_t  kelemen_style_mutation; /* pssmlt: Kelemen (1) or Gaussian (0)        */
    int32_t  no_light_image;     /* 1 = "lightImage" false (mmlt; default: true)   */
    int32_t  timeout_s;          /* "timeout" default 0: stop drmlt_run after this many seconds (equal-time runs) */
    int32_t  no_direct_sampling; /* 1 = "directSampling" false (default true; bdpt only, forced false for mmlt) */
    int32_t  seed_rule;          /* DRMLT_SEED_*: two-stage MLT seeding, default TARGET (see the enum)            */
    int32_t  work_units_rule;    /* DRMLT_WORK_UNITS_*: what work_units = -1 derives, default DEVICE            */
    int32_t  reserved[3];
} drmlt_config;

/* ---- flat scene description -------------------------------------------- */

/* One primitive. `data` by type:
 *   TRIANGLE   p0.xyz p1.xyz p2.xyz            (face normal = (p1-p0)x(p2-p0))
 *   RECTANGLE  row-major 3x4 objectToWorld of Mitsuba's rectangle
 *              (local [-1,1]^2 in the z=0 plane, normal +z; rectangle.cpp:80-112)
 *   SPHERE     center.xyz radius
 */
typedef struct drmlt_shape {
    int32_t type;
    int32_t bsdf;        /* index into bsdfs                         */
    int32_t emitter;     /* index into emitters, or -1               */
    int32_t reserved;
    float   data[12];
} drmlt_shape;

/* DIFFUSE: rgb = reflectance.
 * DIELECTRIC: p[0]=intIOR p[1]=extIOR, rgb unused (specular refl/trans = 1).
 * ROUGHCONDUCTOR: rgb = specularReflectance, p[0]=alpha, p[1..3]=eta rgb,
 *                 p[4..6]=k rgb, p[7]: 0=beckmann 1=ggx.
 * Any other BSDF plugin (smooth conductor, plastic, ...) is refused by
 * drmlt_create, never approximated. */
typedef struct drmlt_bsdf {
    int32_t type;
    float   rgb[3];
    float   p[8];
} drmlt_bsdf;

typedef struct drmlt_emitter {
    int32_t type;        /* DRMLT_EMITTER_AREA                        */
    int32_t shape;       /* index of the shape carrying this emitter */
    float   radiance[3];
    float   sampling_weight; /* Emitter::getSamplingWeight, default 1 */
} drmlt_emitter;

/* perspective pinhole (src/sensors/perspective.cpp) + hdrfilm size/filter */
typedef struct drmlt_camera {
    float   to_world[16];   /* row-major 4x4 camera-to-world (no scale)       */
    float   fov_x_deg;      /* horizontal field of view in degrees            */
    float   near_clip;      /* default 1e-2                                   */
    float   far_clip;       /* default 1e4                                    */
    int32_t width, height;  /* film (= crop) size                             */
    int32_t filter;         /* DRMLT_FILTER_*                                 */
    float   filter_param;   /* box: radius (0.5); gaussian: stddev (0.5)      */
} drmlt_camera;

typedef struct drmlt_scene {
    uint32_t struct_size;       /* = sizeof(drmlt_scene) */
    int32_t  n_shapes;
    int32_t  n_bsdfs;
    int32_t  n_emitters;
    const drmlt_shape   *shapes;
    const drmlt_bsdf    *bsdfs;
    const drmlt_emitter *emitters;
    drmlt_camera camera;
} drmlt_scene;

/* ---- statistics: numerators / denominators of drmlt_proc.cpp:34-49 ----- */

typedef struct drmlt_stats {
    uint64_t first_acc,  first_base;        /* "Accepted 1st-stage mutations"            */
    uint64_t large_acc,  large_base;        /* "... large mutations in the 1st stage"    */
    uint64_t bold_acc,   bold_base;         /* "... bold mutation in the 1st stage"      */
    uint64_t second_acc, second_base;       /* "Accepted 2nd-stage mutations"            */
    uint64_t second_large_acc, second_large_base; /* "... after large mutation"          */
    uint64_t second_bold_acc,  second_bold_base;  /* "... after bold mutation"           */
    uint64_t overall_acc, overall_base;     /* "Overall acceptance rate"                 */
    uint64_t mutations;          /* chain-loop iterations (++mutationCtr, :541)          */
    uint64_t path_evals;         /* sampleSplats calls (stage 1 + stage 2 + reverse)     */
    uint64_t rays;               /* closest-hit + shadow rays traced                     */
    uint64_t accepted;           /* iterations that changed the chain state              */
    double   kernel_ms;          /* device time inside the chain kernels (HIP events)    */
    double   seed_ms;            /* device+host time of drmlt_seed                       */
    uint32_t n_chains;
    uint32_t max_dim;            /* findMaxDimensions(...).sensor                        */
    uint64_t launches;           /* chain-kernel launches so far                         */
    uint64_t bvh_node_visits;    /* 4-wide BVH nodes fetched (128 B each); 0 for brute-force scenes */
    uint64_t bvh_prim_tests;     /* primitive records fetched in BVH leaves (64 B each)           */
    uint64_t bvh_node_iterations; /* wave-level traversal iterations that advanced lanes holding a node ...       */
    uint64_t bvh_leaf_iterations; /* ... and a leaf: bvh_node_visits / (64 x bvh_node_iterations) = share of the
                                   * wave's lanes that ADVANCE per node iteration (the exec mask of the straight-line
                                   * traversal blocks also counts lanes that compute on zeros and keep nothing)      */
} drmlt_stats;

/* one evaluated PSS point: SplatList of pathsampler.cpp:529-567 (one splat) */
typedef struct drmlt_splat {
    float luminance;
    float x, y;          /* sample position in fractional pixel coordinates */
    float rgb[3];        /* un-normalised contribution                      */
    int32_t n_dims;      /* PSS components consumed                         */
    int32_t n_rays;
} drmlt_splat;

typedef struct drmlt_ctx drmlt_ctx;

typedef void (*drmlt_progress_cb)(uint64_t done_mutations, uint64_t total_mutations, void *user);

/* Create a context on HIP device `device`. Validates config + scene the way
 * the DRMLT ctor / PathSampler ctor do and fails (NULL, message in err) on
 * anything unsupported -- never silently approximates. */
drmlt_ctx *drmlt_create(const drmlt_config *cfg, const drmlt_scene *scene,
                        int device, char *err, size_t errlen);

/* Bootstrap: luminance samples, b = mean luminance, luminance-proportional
 * seed resampling, replay of every seed into its chain (with the luminance
 * sanity check). chain ids are [chain_offset, chain_offset + work_units):
 * ranks of a multi-GPU job pass disjoint offsets. */
int drmlt_seed(drmlt_ctx *ctx, uint64_t seed, uint32_t chain_offset, double *b_out);

/* Run `total_mutations` chain-loop iterations spread evenly over the
 * chains (total / work_units each, as drmlt.cpp:475-476). `stop` is polled
 * between kernel launches (may be NULL). */
int drmlt_run(drmlt_ctx *ctx, uint64_t total_mutations, volatile int *stop,
              drmlt_progress_cb cb, void *user);

/* out_rgb (host, W*H*3 floats) = accum * (b / mean_lum(accum)) + direct. */
int drmlt_develop(drmlt_ctx *ctx, const float *direct_rgb_or_null, float *out_rgb);

int drmlt_stats_get(drmlt_ctx *ctx, drmlt_stats *out);

/* f(u): evaluate n PSS points (row-major n x dim floats in [0,1], host).
 * technique=mmlt: a point is [sensor S | emitter E | direct | depth] with
 * S = 2 (maxDepth + 1), E = 2 maxDepth (the components the three samplers can
 * hand to a path), depth as a float; n_dims of the result carries the
 * strategy as well: dims | s << 8 | t << 16. */
int drmlt_eval_paths(drmlt_ctx *ctx, const float *u, uint32_t n, uint32_t dim,
                     drmlt_splat *out);

/* Raw accumulated film (W*H*3 floats, un-normalised), host copy / reset. */
int drmlt_film_read(drmlt_ctx *ctx, float *out_rgb);
int drmlt_film_clear(drmlt_ctx *ctx);
/* Device pointer of the film (W*H*3 fp32) so the caller can reduce it across
 * GPUs (RCCL) without a host round trip; and the override of b used by
 * develop once the per-rank estimates have been averaged (drmlt.cpp:544). */
void *drmlt_film_device_ptr(drmlt_ctx *ctx);
int drmlt_set_luminance(drmlt_ctx *ctx, double b);

/* Two-stage MLT ("twoStage", drmlt.cpp:278,406-418). The adaptor renders the
 * first stage with a second context on a film reduced by
 * firstStageSizeReduction (sample_count multiplied by it, gaussian filter, no
 * direct image: BidirectionalUtils::mltLuminancePass, src/libbidir/util.cpp:96-199),
 * turns its developed image into the full-size luminance image with
 * drmlt_luminance_map (util.cpp:179-196 + core/rfilter.h:123-290) and hands
 * it to the second-stage context BEFORE drmlt_seed. From then on every splat
 * list is weighted by 1 / map[pixel] (SplatList::normalize,
 * pathsampler.cpp:1001-1020) and drmlt_develop multiplies it back
 * (drmlt_proc.cpp:824-845). lum_map: W*H floats, positive; NULL clears it. */
int drmlt_set_importance_map(drmlt_ctx *ctx, const float *lum_map_or_null);
int drmlt_luminance_map(const float *rgb_small, int w, int h, int W, int H, float *out_lum);
/* Launch all kernels of this context on a caller-owned hipStream_t. */
int drmlt_set_stream(drmlt_ctx *ctx, void *hip_stream);
/* Timing of the dominant kernel for bench.py's roofline (HIP events on the
 * launch stream): average ms per launch over the launches since last reset. */
int drmlt_kernel_time(drmlt_ctx *ctx, double *avg_ms, uint64_t *launches, int reset);

/* Plain independent-sample path tracing of the same integrand
 * (test utility: equal-expectation reference for the MLT image). */
int drmlt_render_pt(drmlt_ctx *ctx, uint32_t spp, uint64_t seed, float *out_rgb);

/* Bootstrap inspection (test utilities for the seed-selection parity test): the luminance samples of
 * generateSeeds' first loop (pathsampler.cpp:879-920) for bootstrap stream `stream`, and the sample indices the last
 * drmlt_seed / drmlt_seed_pool picked for this context's chains (sorted, n_chains values). */
int drmlt_bootstrap_luminances(drmlt_ctx *ctx, uint64_t seed, uint32_t stream, uint32_t n, float *out);
int drmlt_seed_indices(drmlt_ctx *ctx, uint32_t *out);

/* technique=bdpt: f(u) is a splat LIST (one sensor-side splat accumulating all
 * t >= 2 strategies + one light-image splat per t = 1 strategy,
 * pathsampler.cpp:357-361,514-519). A point is [sensor S | emitter E]
 * (drmlt_stats.max_dim / 2 each at most); a result row is `stride` floats:
 * [lum, hasMain, px, py, r, g, b, nMore, nDims, nRays, nMore x (px, py, r, g, b)]. */
int drmlt_eval_lists(drmlt_ctx *ctx, const float *u, uint32_t n, uint32_t dim,
                     float *out, uint32_t stride);

/* Chain state dump for parity tests: lum/x/y/rgb of chain's current state and
 * the first `dim` PSS components (u is n_chains x dim, may be NULL).
 * technique=mmlt: components are [sensor S | emitter E | direct]; n_dims = the
 * chain's path depth, n_rays = t of the current state. technique=bdpt:
 * components are [sensor S | emitter E]; cur = main splat of the current list
 * (normalised), n_dims = 1 if it exists, n_rays = number of light-image splats. */
int drmlt_chain_state(drmlt_ctx *ctx, drmlt_splat *cur, float *u, uint32_t dim);

/* ---- several GPUs --------------------------------------------------------
 * Chains are independent given their seeds (drmlt_proc.cpp:869-883): the chains
 * of a render are partitioned over the GPUs, every GPU accumulates a full-frame
 * film, and the films are summed once at the end, where the reference merges
 * its work units' ImageBlocks (DRMLTProcess::processResult, drmlt_proc.cpp:856-867):
 * ncclReduceScatter(sum) leaves rank r with rows [r * ceil(H / N), ...) of the
 * summed film, a two-element ncclAllReduce shares the film's total luminance
 * (and the ranks' b), every rank develops its own tile (develop, :813-854).
 * RCCL is called from C++ inside the library (loaded at run time). */

/* Seeds from ONE pool for the whole job: the bootstrap is sized for
 * `pool_chains` chains, `pool_chains` seeds are drawn (sorted), and this
 * context takes seeds and chain ids [first_chain, first_chain + work_units).
 * Every rank computes the same list and the same b; a job split over several
 * contexts runs exactly the chains of one context with pool_chains work units. */
int drmlt_seed_pool(drmlt_ctx *ctx, uint64_t seed, uint32_t first_chain,
                    uint32_t pool_chains, double *b_out);

/* (a) one process per GPU (any launcher). Rank 0 calls drmlt_comm_unique_id and
 * hands the id to the other ranks (it is ncclUniqueId); every rank then calls
 * drmlt_comm_init on its context. drmlt_exchange_tiled runs the film exchange
 * on the context's stream: *b_inout = this rank's b in, the ranks' mean out;
 * the developed tile (rows [*row_lo, *row_hi), W * 3 floats each) is copied to
 * tile_host_or_null when given. The local film is left untouched. With
 * tile_host_or_null, row_lo and row_hi all NULL nothing waits for the host: the
 * exchange (tile developed with device-resident sums) is only enqueued on the
 * context's stream behind the chain kernels, and *b_inout is not updated. */
#define DRMLT_COMM_ID_BYTES 128
int drmlt_comm_unique_id(char id[DRMLT_COMM_ID_BYTES]);
int drmlt_comm_init(drmlt_ctx *ctx, const char id[DRMLT_COMM_ID_BYTES], int rank, int world);
int drmlt_exchange_tiled(drmlt_ctx *ctx, double *b_inout, float *tile_host_or_null,
                         int *row_lo, int *row_hi);
/* What the communicator itself reports (ncclCommCount / ncclCommUserRank), not
 * what the caller asked for: a launcher checks it against its own world size. */
int drmlt_comm_info(drmlt_ctx *ctx, int *nranks, int *rank);
/* The row partition of the exchange as pure arithmetic (no device needed):
 * rank `rank` of `world` owns rows [*row_lo, *row_hi) of an H-row film, the
 * reduce-scatter moves *rows_per_rank = ceil(H / world) rows per rank (the film
 * allocation carries the zero rows that pads H to world * rows_per_rank).
 * DRMLT_E_INVALID when world is not a valid partition (world > 16, rank out
 * of range). Replaces the row bookkeeping a caller of the reference's
 * processResult (drmlt_proc.cpp:856-867) never needed: it merged whole frames. */
int drmlt_film_tile(int height, int rank, int world, int *row_lo, int *row_hi, int *rows_per_rank);

/* (b) one process drives the GPUs of `device_mask` (bit d = HIP device d): what
 * the Mitsuba plugin uses, so that `-D integrator=drmlt` renders on the whole
 * node. cfg->work_units is PER DEVICE. Same call sequence as a context:
 * create -> [set_importance_map] -> seed -> run -> develop (-> stats). */
typedef struct drmlt_node drmlt_node;
drmlt_node *drmlt_node_create(const drmlt_config *cfg, const drmlt_scene *scene,
                              uint32_t device_mask, char *err, size_t errlen);
int drmlt_node_seed(drmlt_node *node, uint64_t seed, double *b_out);
int drmlt_node_run(drmlt_node *node, uint64_t total_mutations, volatile int *stop,
                   drmlt_progress_cb cb, void *user);
int drmlt_node_develop(drmlt_node *node, const float *direct_rgb_or_null, float *out_rgb);
int drmlt_node_stats_get(drmlt_node *node, drmlt_stats *out);
int drmlt_node_set_importance_map(drmlt_node *node, const float *lum_map_or_null);
int drmlt_node_device_count(drmlt_node *node);
drmlt_ctx *drmlt_node_context(drmlt_node *node, int rank); /* borrowed; for film / chain inspection */
const char *drmlt_node_last_error(drmlt_node *node);
void drmlt_node_destroy(drmlt_node *node);

const char *drmlt_last_error(drmlt_ctx *ctx);
uint32_t drmlt_abi_version(void);
void drmlt_destroy(drmlt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* DRMLT_ABI_H */
