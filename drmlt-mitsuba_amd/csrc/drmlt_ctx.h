// Internal (not ABI): the context behind `drmlt_ctx`, shared by drmlt_capi.cpp (one device) and drmlt_node.cpp
// (several devices of one node, RCCL).
#pragma once
#include "../../include/drmlt_abi.h"
#include "device_types.h"
#include "film_tiles.h" // FILM_PAD_ROWS, film_tile(): the row partition of the tiled exchange

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void) hipFree(p); }
    hipError_t alloc(size_t n) {
        if (p) { (void) hipFree(p); p = nullptr; }
        bytes = n;
        return n ? hipMalloc(&p, n) : hipSuccess;
    }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

// RAII pair of HIP events (drmlt_seed / drmlt_run time their kernels with them; an early return must not leak them)
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    EventPair() = default;
    EventPair(const EventPair &) = delete;
    EventPair &operator=(const EventPair &) = delete;
    EventPair(EventPair &&o) noexcept : a(o.a), b(o.b) { o.a = o.b = nullptr; }
    ~EventPair() {
        if (a) (void) hipEventDestroy(a);
        if (b) (void) hipEventDestroy(b);
    }
    hipError_t create() {
        hipError_t e = hipEventCreate(&a);
        return e != hipSuccess ? e : hipEventCreate(&b);
    }
    float elapsed_ms() const {
        float ms = 0.f;
        (void) hipEventElapsedTime(&ms, a, b);
        return ms;
    }
};

struct drmlt_comm; // RCCL communicator + tile buffers of one rank (drmlt_node.cpp)

struct drmlt_ctx {
    drmlt_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DParams P{};
    std::string error;

    int bvh_depth = 0;
    int ovf_entries = 0;   // capacity per lane of the traversal stacks' overflow area (0: every stack fits its LDS column)
    size_t ovf_lanes = 0;  // columns allocated in d_ovf
    DevBuf d_prims, d_shade, d_bsdfs, d_emitters, d_bvh, d_lut, d_film, d_x, d_cur, d_stats, d_err, d_scratch, d_chain_i, d_importance, d_bd_verts, d_bd_lists, d_prims_flat, d_prims_box, d_regroup, d_ovf, d_order, d_done, d_rows;
    std::vector<DPrim> prims;
    std::vector<DShade> shade;
    std::vector<uint32_t> seed_indices; // bootstrap sample index of every chain's seed (last drmlt_seed)
    size_t film_floats = 0; // W * H * 3 (the allocation carries FILM_PAD_ROWS more rows of zeros for the tiled reduce)

    uint32_t n_chains = 0, mutation_base = 0, chain_offset = 0;
    bool seeded = false;
    double b = 0.0;
    // accounting
    uint64_t mutations = 0, launches = 0, accepted_dummy = 0;
    double kernel_ms = 0.0, seed_ms = 0.0;
    double kt_ms = 0.0; uint64_t kt_launches = 0; // drmlt_kernel_time window
    uint64_t host_counters[9] = {0};
    unsigned regroup_checks = 0; // DRMLT_REGROUP_CHECK (test hook): device permutations compared with the host's so far
    bool regrouped = false; // the bidirectional kernels' execution order has been regrouped by work at least once since the last seed (drmlt_capi.cpp: regroup_chains)
    int slice = 1024; // mutations per chain per launch (<= 32768: the per-lane event counters are 16 bit)
    drmlt_comm *comm = nullptr; // set by drmlt_comm_init / drmlt_node_create

    ~drmlt_ctx();
    int fail(int code, const char *fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        error = buf;
        return code;
    }
};


#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return (ctx)->fail(DRMLT_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

void drmlt_comm_release(drmlt_comm *c); // drmlt_node.cpp
void launch_lum_sum(const float *film, const float *importance, uint32_t n_pixels, double *sum, hipStream_t st);
void launch_develop(const float *film, const float *direct, const float *importance, float factor, uint32_t n, float *out, hipStream_t st);
void launch_develop_dev(const float *film, const float *importance, const double *scal, float inv_world, float inv_pixels, int acceptance_map, uint32_t n,
                        float *out, hipStream_t st);
void launch_set2(double *p, double a, double b, hipStream_t st);
void launch_set_u32(uint32_t *p, uint32_t v, hipStream_t st);
