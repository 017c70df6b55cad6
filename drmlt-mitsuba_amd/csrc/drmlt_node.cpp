// Multi-GPU layer of the C-ABI (include/drmlt_abi.h, "one node, several GPUs").
//
// Reference merge point: every work unit's ImageBlock is summed into one accumulation buffer under a mutex
// (DRMLTProcess::processResult, src/integrators/drmlt/drmlt_proc.cpp:856-867) and normalised once (develop, :813-854).
// Chains are independent given their seeds (:869-883), so the chains of a render are partitioned over the GPUs with no
// data-path collective; the only exchange is the film: ncclReduceScatter(sum) leaves rank r with rows
// [r * ceil(H / N), ...) of the summed film ("the image tiled across the GPUs" is realised at the reduction), one
// two-element ncclAllReduce shares the summed film's luminance (and the ranks' b estimates) so that every rank applies
// the same develop factor to its tile (:824-839).
//
// Two ways in:
//   * one process per GPU (bench.py under torchrun, any launcher): drmlt_comm_unique_id / drmlt_comm_init /
//     drmlt_exchange_tiled on an ordinary drmlt_ctx;
//   * one process, N GPUs (the Mitsuba plugin: `-D integrator=drmlt` drives the whole node): drmlt_node_*.
// RCCL is loaded at run time (dlopen) the first time a communicator is needed: a single-GPU host without RCCL can
// still use the library.
#include "drmlt_ctx.h"
#include "film_tiles.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

void launch_accumulate(float *dst, const float *src, size_t n, hipStream_t st); // kernels.hip

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// nullptr + message when RCCL cannot be loaded
const RcclApi *rccl(std::string &err) {
    static RcclApi api;
    static std::string load_error;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *forced = getenv("DRMLT_RCCL_LIB"); // an explicit choice is final: no fall-back to the system's library
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (forced && *forced) api.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        else
            for (const char *n : names) {
                api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (api.lib) break;
            }
        if (!api.lib) {
            const char *why = dlerror(); // ONE call: dlerror() hands the message out once and then returns NULL
            load_error = std::string("RCCL is not available: ") + (why ? why : "librccl.so.1 not found");
            return;
        }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(api.lib, name);
            if (!p && load_error.empty()) load_error = std::string("RCCL lacks the symbol ") + name;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
        api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
        api.ReduceScatter = reinterpret_cast<decltype(api.ReduceScatter)>(sym("ncclReduceScatter"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    if (!load_error.empty()) { err = load_error; return nullptr; }
    return &api;
}

} // namespace

// One rank of the film exchange. `comm` == nullptr: the loopback transport of a node whose ranks share a device (tests on
// a one-GPU box; RCCL refuses two ranks on one device) -- the node sums the films with a device kernel instead.
struct drmlt_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    int tile_rows = 0; // ceil(H / world)
    DevBuf tile, scal, out;
};

void drmlt_comm_release(drmlt_comm *c) {
    if (!c) return;
    std::string e;
    if (c->comm) if (const RcclApi *R = rccl(e)) (void) R->CommDestroy(c->comm);
    delete c;
}

#define NCCL_TRY(ctx, R, expr)                                                                         \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) return (ctx)->fail(DRMLT_E_DEVICE, "%s: %s", #expr, (R)->GetErrorString(r_)); \
    } while (0)

namespace {

int comm_attach(drmlt_ctx *ctx, ncclComm_t comm, int rank, int world) {
    if (ctx->comm) { drmlt_comm_release(ctx->comm); ctx->comm = nullptr; }
    std::unique_ptr<drmlt_comm> c(new drmlt_comm());
    c->comm = comm; c->rank = rank; c->world = world;
    FilmTile ft;
    if (!film_tile(ctx->P.height, rank, world, ft)) return ctx->fail(DRMLT_E_INVALID, "world size %d is too large for the film's row padding", world);
    c->tile_rows = ft.rows_per_rank;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t) c->tile_rows * ctx->P.width * 3;
    HIP_TRY(ctx, c->tile.alloc(n * sizeof(float)));
    HIP_TRY(ctx, c->out.alloc(n * sizeof(float)));
    HIP_TRY(ctx, c->scal.alloc(2 * sizeof(double)));
    ctx->comm = c.release();
    return DRMLT_OK;
}

// rows [lo, hi) of the film that rank `rank` of `world` owns after the reduce-scatter (hi clipped to the film)
void tile_range(const drmlt_ctx *ctx, int rank, int world, int &lo, int &hi) {
    FilmTile ft{0, 0, 0};
    (void) film_tile(ctx->P.height, rank, world, ft); // validated when the communicator was attached
    lo = ft.lo; hi = ft.hi;
}

// Steps 2-5 of the exchange, once the summed tile sits in comm->tile: tile luminance, scalar all-reduce, develop.
// `aborted` (node exchange only): set by a rank whose collective could not be enqueued, after it has called ncclCommAbort on every
// communicator of the node -- checked before the RCCL call so that no rank hands an aborted communicator to the library.
int finish_tile(drmlt_ctx *ctx, const RcclApi *R, double *b_inout, const float *direct_tile_or_null, float *tile_host_or_null, bool want_results = true,
                const std::atomic<bool> *aborted = nullptr) {
    drmlt_comm *c = ctx->comm;
    int lo, hi;
    tile_range(ctx, c->rank, c->world, lo, hi);
    const uint32_t npix = (uint32_t) (hi - lo) * ctx->P.width, n = npix * 3;
    const float *imp = ctx->P.importance ? ctx->P.importance + (size_t) lo * ctx->P.width : nullptr;
    double host[2] = {0.0, b_inout ? *b_inout : ctx->b};
    launch_set2(c->scal.as<double>(), host[0], host[1], ctx->stream); // (a kernel, not a copy from pageable memory: nothing here waits for the host)
    if (npix) launch_lum_sum(c->tile.as<float>(), imp, npix, c->scal.as<double>(), ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    if (c->comm) {
        if (aborted && aborted->load(std::memory_order_acquire)) return ctx->fail(DRMLT_E_DEVICE, "the film exchange was aborted by another rank");
        NCCL_TRY(ctx, R, R->AllReduce(c->scal.p, c->scal.p, 2, ncclDouble, ncclSum, c->comm, ctx->stream));
    }
    if (!want_results) { // nobody reads the tile or the mean b now: develop with the device-resident sums, no host round trip
        if (n) launch_develop_dev(c->tile.as<float>(), imp, c->scal.as<double>(), c->comm ? 1.f / (float) c->world : 1.f,
                                  1.f / ((float) ctx->P.width * (float) ctx->P.height), ctx->cfg.acceptance_map, n, c->out.as<float>(), ctx->stream);
        HIP_TRY(ctx, hipGetLastError());
        return DRMLT_OK;
    }
    HIP_TRY(ctx, hipMemcpyAsync(host, c->scal.p, sizeof host, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const double b_mean = c->comm ? host[1] / c->world : host[1]; // loopback: the node hands in the job's b
    const double avg = host[0] / ((double) ctx->P.width * ctx->P.height);
    const double factor = ctx->cfg.acceptance_map ? 1.0 : b_mean / avg; // drmlt_proc.cpp:834-839
    DevBuf d_direct;
    if (direct_tile_or_null && n) {
        HIP_TRY(ctx, d_direct.alloc((size_t) n * sizeof(float)));
        HIP_TRY(ctx, hipMemcpyAsync(d_direct.p, direct_tile_or_null, (size_t) n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    }
    if (n) launch_develop(c->tile.as<float>(), d_direct.as<float>(), imp, (float) factor, n, c->out.as<float>(), ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    if (tile_host_or_null && n) HIP_TRY(ctx, hipMemcpyAsync(tile_host_or_null, c->out.p, (size_t) n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (b_inout) *b_inout = b_mean;
    return DRMLT_OK;
}

} // namespace

extern "C" {

int drmlt_comm_unique_id(char id[DRMLT_COMM_ID_BYTES]) {
    if (!id) return DRMLT_E_INVALID;
    static_assert(sizeof(ncclUniqueId) <= DRMLT_COMM_ID_BYTES, "unique id does not fit the ABI buffer");
    std::string err;
    const RcclApi *R = rccl(err);
    if (!R) return DRMLT_E_DEVICE;
    ncclUniqueId u;
    if (R->GetUniqueId(&u) != ncclSuccess) return DRMLT_E_DEVICE;
    memset(id, 0, DRMLT_COMM_ID_BYTES);
    memcpy(id, &u, sizeof u);
    return DRMLT_OK;
}

int drmlt_comm_init(drmlt_ctx *ctx, const char id[DRMLT_COMM_ID_BYTES], int rank, int world) {
    if (!ctx || !id) return DRMLT_E_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return ctx->fail(DRMLT_E_INVALID, "bad rank %d of %d", rank, world);
    std::string err;
    const RcclApi *R = rccl(err);
    if (!R) return ctx->fail(DRMLT_E_DEVICE, "%s", err.c_str());
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    NCCL_TRY(ctx, R, R->CommInitRank(&comm, world, u, rank));
    const int rc = comm_attach(ctx, comm, rank, world);
    if (rc != DRMLT_OK) (void) R->CommDestroy(comm);
    return rc;
}

// Pure arithmetic (no device, no context): the rows of an H-row film that `rank` of `world` owns after the exchange and
// the per-rank row count of the reduce-scatter -- film_tiles.h, the copy every path of the library uses.
int drmlt_film_tile(int height, int rank, int world, int *row_lo, int *row_hi, int *rows_per_rank) {
    FilmTile ft;
    if (!film_tile(height, rank, world, ft)) return DRMLT_E_INVALID;
    if (row_lo) *row_lo = ft.lo;
    if (row_hi) *row_hi = ft.hi;
    if (rows_per_rank) *rows_per_rank = ft.rows_per_rank;
    return DRMLT_OK;
}

// What the communicator itself says (ncclCommCount / ncclCommUserRank), not what the caller asked for: bench.py prints it
// and fails when it differs from --gpus.
int drmlt_comm_info(drmlt_ctx *ctx, int *nranks, int *rank) {
    if (!ctx) return DRMLT_E_INVALID;
    if (!ctx->comm) return ctx->fail(DRMLT_E_STATE, "drmlt_comm_info needs a communicator (drmlt_comm_init / drmlt_node_create)");
    int n = ctx->comm->world, r = ctx->comm->rank; // loopback transport: the node's own bookkeeping
    if (ctx->comm->comm) {
        std::string err;
        const RcclApi *R = rccl(err);
        if (!R) return ctx->fail(DRMLT_E_DEVICE, "%s", err.c_str());
        NCCL_TRY(ctx, R, R->CommCount(ctx->comm->comm, &n));
        NCCL_TRY(ctx, R, R->CommUserRank(ctx->comm->comm, &r));
    }
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    return DRMLT_OK;
}

int drmlt_exchange_tiled(drmlt_ctx *ctx, double *b_inout, float *tile_host_or_null, int *row_lo, int *row_hi) {
    if (!ctx) return DRMLT_E_INVALID;
    if (!ctx->comm || !ctx->comm->comm) return ctx->fail(DRMLT_E_STATE, "drmlt_exchange_tiled needs drmlt_comm_init first");
    std::string err;
    const RcclApi *R = rccl(err);
    if (!R) return ctx->fail(DRMLT_E_DEVICE, "%s", err.c_str());
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    drmlt_comm *c = ctx->comm;
    const size_t count = (size_t) c->tile_rows * ctx->P.width * 3; // the film allocation is padded with zero rows up to world * count
    NCCL_TRY(ctx, R, R->ReduceScatter(ctx->d_film.p, c->tile.p, count, ncclFloat, ncclSum, c->comm, ctx->stream));
    int lo, hi;
    tile_range(ctx, c->rank, c->world, lo, hi);
    if (row_lo) *row_lo = lo;
    if (row_hi) *row_hi = hi;
    // tile_host_or_null == NULL and row_lo == NULL: fire and forget (a render step's exchange): everything is enqueued on the
    // context's stream, *b_inout is read but not updated, the developed tile stays on the device
    return finish_tile(ctx, R, b_inout, nullptr, tile_host_or_null, tile_host_or_null != nullptr || row_lo != nullptr);
}

} // extern "C"

// ------------------------------------------------------------------------------------------ one process, N GPUs
struct drmlt_node {
    std::vector<drmlt_ctx *> subs;
    std::vector<int> devices;
    std::string error;
    bool loopback = false; // ranks share a device: films summed by a device kernel (tests on a one-GPU box)
    bool seeded = false;
    double b = 0.0;
    ~drmlt_node() { for (drmlt_ctx *c : subs) drmlt_destroy(c); }
    int fail(int code, const std::string &msg) { error = msg; return code; }
};

namespace {

// run f(rank) on one host thread per device; first failure wins
template <class F> int for_each_rank(drmlt_node *node, F f) {
    const int n = (int) node->subs.size();
    std::vector<int> rc(n, DRMLT_OK);
    if (n == 1) rc[0] = f(0);
    else {
        std::vector<std::thread> th;
        for (int r = 0; r < n; ++r) th.emplace_back([&, r] { rc[r] = f(r); });
        for (auto &t : th) t.join();
    }
    for (int r = 0; r < n; ++r)
        if (rc[r] != DRMLT_OK) return node->fail(rc[r], "device " + std::to_string(node->devices[r]) + ": " + drmlt_last_error(node->subs[r]));
    return DRMLT_OK;
}

} // namespace

extern "C" {

drmlt_node *drmlt_node_create(const drmlt_config *cfg, const drmlt_scene *scene, uint32_t device_mask, char *err, size_t errlen) {
    auto bail = [&](drmlt_node *n, const std::string &msg) -> drmlt_node * {
        if (err && errlen) snprintf(err, errlen, "%s", msg.c_str());
        delete n;
        return nullptr;
    };
    std::vector<int> devs;
    // Test hook, honoured only together with DRMLT_TEST_HOOKS=1 (a stale variable must not redirect a production render):
    // an explicit rank -> device list that may repeat a device ("0,0": two ranks on one GPU, loopback transport).
    const char *hook = getenv("DRMLT_NODE_DEVICES");
    const char *hooks_on = getenv("DRMLT_TEST_HOOKS");
    if (hook && *hook && hooks_on && atoi(hooks_on) == 1) {
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess) n_dev = 0;
        for (const char *p = hook; *p;) {
            char *end = nullptr;
            const long d = strtol(p, &end, 10);
            if (end == p || (*end && *end != ',') || d < 0 || d >= n_dev)
                return bail(nullptr, std::string("DRMLT_NODE_DEVICES: bad entry in \"") + hook + "\" (" + std::to_string(n_dev) + " devices visible)");
            devs.push_back((int) d);
            p = *end ? end + 1 : end;
        }
        fprintf(stderr, "[drmlt] test hook: DRMLT_NODE_DEVICES=%s overrides device mask 0x%x\n", hook, device_mask);
    } else {
        for (int d = 0; d < 32; ++d) if (device_mask & (1u << d)) devs.push_back(d);
    }
    if (devs.empty()) return bail(nullptr, "empty device mask");
    if (devs.size() > FILM_PAD_ROWS) return bail(nullptr, "at most 16 devices per node");
    std::unique_ptr<drmlt_node> node(new drmlt_node());
    node->devices = devs;
    for (int d : devs) {
        char msg[512] = {0};
        drmlt_ctx *c = drmlt_create(cfg, scene, d, msg, sizeof msg);
        if (!c) return bail(node.release(), "device " + std::to_string(d) + ": " + msg);
        node->subs.push_back(c);
    }
    const int n = (int) devs.size();
    std::vector<int> sorted = devs;
    std::sort(sorted.begin(), sorted.end());
    node->loopback = std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end();
    std::vector<ncclComm_t> comms(n, nullptr);
    if (n > 1 && !node->loopback) {
        std::string e;
        const RcclApi *R = rccl(e);
        if (!R) return bail(node.release(), e);
        ncclResult_t r = R->CommInitAll(comms.data(), n, devs.data());
        if (r != ncclSuccess) return bail(node.release(), std::string("ncclCommInitAll: ") + R->GetErrorString(r));
    }
    for (int r = 0; r < n; ++r)
        if (comm_attach(node->subs[r], comms[r], r, n) != DRMLT_OK) return bail(node.release(), drmlt_last_error(node->subs[r]));
    return node.release();
}

void drmlt_node_destroy(drmlt_node *node) { delete node; }
const char *drmlt_node_last_error(drmlt_node *node) { return node ? node->error.c_str() : "null node"; }
int drmlt_node_device_count(drmlt_node *node) { return node ? (int) node->subs.size() : 0; }
drmlt_ctx *drmlt_node_context(drmlt_node *node, int rank) { return (node && rank >= 0 && rank < (int) node->subs.size()) ? node->subs[rank] : nullptr; }

// Seeds from ONE pool for the whole job: rank r runs chains [r n, (r + 1) n) of the N n chains a single
// context with N n work units would run; every rank finds the same b.
int drmlt_node_seed(drmlt_node *node, uint64_t seed, double *b_out) {
    if (!node) return DRMLT_E_INVALID;
    const uint32_t n = node->subs[0]->n_chains, total = n * (uint32_t) node->subs.size();
    std::vector<double> b(node->subs.size(), 0.0);
    int rc = for_each_rank(node, [&](int r) { return drmlt_seed_pool(node->subs[r], seed, (uint32_t) r * n, total, &b[r]); });
    if (rc != DRMLT_OK) return rc;
    node->b = b[0];
    node->seeded = true;
    if (b_out) *b_out = node->b;
    return DRMLT_OK;
}

// total_mutations over ALL chains of the node; the callback reports rank 0's progress scaled to the job
int drmlt_node_run(drmlt_node *node, uint64_t total_mutations, volatile int *stop, drmlt_progress_cb cb, void *user) {
    if (!node) return DRMLT_E_INVALID;
    if (!node->seeded) return node->fail(DRMLT_E_STATE, "drmlt_node_run called before drmlt_node_seed");
    const uint64_t n = node->subs.size(), share = total_mutations / n;
    struct Fwd { drmlt_progress_cb cb; void *user; uint64_t n; } fwd{cb, user, n};
    auto thunk = [](uint64_t done, uint64_t total, void *u) { Fwd *f = static_cast<Fwd *>(u); f->cb(done * f->n, total * f->n, f->user); };
    return for_each_rank(node, [&](int r) { return drmlt_run(node->subs[r], share, stop, (cb && r == 0) ? (drmlt_progress_cb) thunk : nullptr, &fwd); });
}

int drmlt_node_develop(drmlt_node *node, const float *direct_rgb_or_null, float *out_rgb) {
    if (!node || !out_rgb) return DRMLT_E_INVALID;
    if (!node->seeded) return node->fail(DRMLT_E_STATE, "drmlt_node_develop called before drmlt_node_seed");
    const int n = (int) node->subs.size();
    const int W = node->subs[0]->P.width;
    std::string e;
    const RcclApi *R = (n > 1 && !node->loopback) ? rccl(e) : nullptr;
    if (n > 1 && !node->loopback && !R) return node->fail(DRMLT_E_DEVICE, e);
    if (node->loopback || n == 1) {
        // ranks on one device: tile r = sum over ranks of rows [lo_r, hi_r) -- the arithmetic of the reduce-scatter
        for (int r = 0; r < n; ++r) {
            drmlt_ctx *c = node->subs[r];
            int lo, hi;
            tile_range(c, r, n, lo, hi);
            const size_t cnt = (size_t) (hi - lo) * W * 3, off = (size_t) lo * W * 3;
            if (hipSetDevice(c->device) != hipSuccess) return node->fail(DRMLT_E_DEVICE, "hipSetDevice failed");
            if (hipMemsetAsync(c->comm->tile.p, 0, c->comm->tile.bytes, c->stream) != hipSuccess) return node->fail(DRMLT_E_DEVICE, "hipMemsetAsync failed");
            for (int s = 0; s < n && cnt; ++s) {
                (void) hipStreamSynchronize(node->subs[s]->stream);
                launch_accumulate(c->comm->tile.as<float>(), node->subs[s]->d_film.as<float>() + off, cnt, c->stream);
            }
            if (hipStreamSynchronize(c->stream) != hipSuccess) return node->fail(DRMLT_E_DEVICE, "film accumulation failed");
        }
        // the tiles' luminances are summed on the host here (what the scalar all-reduce does between devices)
        std::vector<double> lum(n, 0.0);
        for (int r = 0; r < n; ++r) {
            drmlt_ctx *c = node->subs[r];
            int lo, hi;
            tile_range(c, r, n, lo, hi);
            const uint32_t npix = (uint32_t) (hi - lo) * W;
            double zero = 0.0;
            (void) hipMemcpyAsync(c->comm->scal.p, &zero, sizeof zero, hipMemcpyHostToDevice, c->stream);
            const float *imp = c->P.importance ? c->P.importance + (size_t) lo * W : nullptr;
            if (npix) launch_lum_sum(c->comm->tile.as<float>(), imp, npix, c->comm->scal.as<double>(), c->stream);
            (void) hipMemcpyAsync(&lum[r], c->comm->scal.p, sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (hipStreamSynchronize(c->stream) != hipSuccess) return node->fail(DRMLT_E_DEVICE, "tile luminance failed");
        }
        double total = 0.0;
        for (double v : lum) total += v;
        const double avg = total / ((double) W * node->subs[0]->P.height);
        const double factor = node->subs[0]->cfg.acceptance_map ? 1.0 : node->b / avg;
        for (int r = 0; r < n; ++r) {
            drmlt_ctx *c = node->subs[r];
            int lo, hi;
            tile_range(c, r, n, lo, hi);
            const uint32_t cnt = (uint32_t) (hi - lo) * W * 3;
            if (!cnt) continue;
            const size_t off = (size_t) lo * W * 3;
            DevBuf d_direct;
            if (direct_rgb_or_null) {
                if (d_direct.alloc((size_t) cnt * sizeof(float)) != hipSuccess) return node->fail(DRMLT_E_DEVICE, "allocation failed");
                (void) hipMemcpyAsync(d_direct.p, direct_rgb_or_null + off, (size_t) cnt * sizeof(float), hipMemcpyHostToDevice, c->stream);
            }
            const float *imp = c->P.importance ? c->P.importance + (size_t) lo * W : nullptr;
            launch_develop(c->comm->tile.as<float>(), d_direct.as<float>(), imp, (float) factor, cnt, c->comm->out.as<float>(), c->stream);
            (void) hipMemcpyAsync(out_rgb + off, c->comm->out.p, (size_t) cnt * sizeof(float), hipMemcpyDeviceToHost, c->stream);
            if (hipStreamSynchronize(c->stream) != hipSuccess) return node->fail(DRMLT_E_DEVICE, "tile develop failed");
        }
        return DRMLT_OK;
    }
    // RCCL: one host thread per device, each issues its rank's reduce-scatter + all-reduce and develops its tile. A rank
    // that fails BEFORE its collective is enqueued would leave the others waiting in it for ever, so (1) everything that can
    // fail without RCCL is checked on all ranks first, (2) a rank whose enqueue fails aborts every communicator of the
    // node (ncclCommAbort releases the peers' pending operations); the node cannot exchange films after that.
    for (int r = 0; r < n; ++r) {
        drmlt_ctx *c = node->subs[r];
        if (!c->comm || !c->comm->comm) return node->fail(DRMLT_E_STATE, "the node's communicators are gone (an earlier exchange failed)");
        if (hipSetDevice(c->device) != hipSuccess) return node->fail(DRMLT_E_DEVICE, "device " + std::to_string(c->device) + ": hipSetDevice failed");
        if (hipStreamQuery(c->stream) == hipErrorInvalidResourceHandle) return node->fail(DRMLT_E_DEVICE, "device " + std::to_string(c->device) + ": stream is gone");
    }
    //  abort_all runs on the failing rank's thread while its peers may be inside finish_tile: it only ABORTS the
    // communicators (ncclCommAbort is made to be called from another thread: it releases the peers' pending operations) and
    // raises the flag; the handles themselves are written (nulled) after every thread has joined, never under a reader.
    std::mutex abort_mutex;
    std::atomic<bool> aborted{false};
    auto abort_all = [&]() {
        std::lock_guard<std::mutex> g(abort_mutex);
        if (aborted.load(std::memory_order_relaxed)) return;
        aborted.store(true, std::memory_order_release);
        for (drmlt_ctx *c : node->subs)
            if (c->comm && c->comm->comm) (void) R->CommAbort(c->comm->comm);
    };
    const int rc_all = for_each_rank(node, [&](int r) {
        drmlt_ctx *c = node->subs[r];
        int rc = DRMLT_OK;
        if (hipSetDevice(c->device) != hipSuccess) rc = c->fail(DRMLT_E_DEVICE, "hipSetDevice failed");
        const size_t count = (size_t) c->comm->tile_rows * W * 3;
        if (rc == DRMLT_OK && aborted.load(std::memory_order_acquire)) rc = c->fail(DRMLT_E_DEVICE, "the film exchange was aborted by another rank");
        if (rc == DRMLT_OK) {
            const ncclResult_t nr = R->ReduceScatter(c->d_film.p, c->comm->tile.p, count, ncclFloat, ncclSum, c->comm->comm, c->stream);
            if (nr != ncclSuccess) rc = c->fail(DRMLT_E_DEVICE, "ncclReduceScatter: %s", R->GetErrorString(nr));
        }
        if (rc == DRMLT_OK) {
            int lo, hi;
            tile_range(c, r, n, lo, hi);
            const size_t off = (size_t) lo * W * 3;
            double b = node->b;
            rc = finish_tile(c, R, &b, direct_rgb_or_null ? direct_rgb_or_null + off : nullptr, out_rgb + off, true, &aborted);
        }
        if (rc != DRMLT_OK) abort_all();
        return rc;
    });
    if (aborted.load()) // every thread has joined: the aborted communicators are gone, forget their handles
        for (drmlt_ctx *c : node->subs) if (c->comm) c->comm->comm = nullptr;
    return rc_all;
}

int drmlt_node_stats_get(drmlt_node *node, drmlt_stats *out) {
    if (!node || !out) return DRMLT_E_INVALID;
    memset(out, 0, sizeof *out);
    for (size_t r = 0; r < node->subs.size(); ++r) {
        drmlt_stats s;
        int rc = drmlt_stats_get(node->subs[r], &s);
        if (rc != DRMLT_OK) return node->fail(rc, drmlt_last_error(node->subs[r]));
        uint64_t *o = &out->first_acc;
        const uint64_t *i = &s.first_acc;
        for (int k = 0; k < 18; ++k) o[k] += i[k]; // the 7 ratio pairs + mutations, path_evals, rays, accepted
        out->kernel_ms = std::max(out->kernel_ms, s.kernel_ms); // devices run concurrently
        out->seed_ms = std::max(out->seed_ms, s.seed_ms);
        out->n_chains += s.n_chains;
        out->max_dim = s.max_dim;
        out->launches += s.launches;
        out->bvh_node_visits += s.bvh_node_visits; out->bvh_prim_tests += s.bvh_prim_tests;
        out->bvh_node_iterations += s.bvh_node_iterations; out->bvh_leaf_iterations += s.bvh_leaf_iterations;
    }
    return DRMLT_OK;
}

int drmlt_node_set_importance_map(drmlt_node *node, const float *lum_map_or_null) {
    if (!node) return DRMLT_E_INVALID;
    for (drmlt_ctx *c : node->subs) {
        int rc = drmlt_set_importance_map(c, lum_map_or_null);
        if (rc != DRMLT_OK) return node->fail(rc, drmlt_last_error(c));
    }
    return DRMLT_OK;
}

} // extern "C"
