// Row partition of the film over the ranks of a node -- the arithmetic of the tiled exchange, host only, no HIP types:
// drmlt_node.cpp (reduce-scatter counts, develop ranges), drmlt_capi.cpp (film allocation) and the C-ABI's
// drmlt_film_tile() all use THIS copy, and tests/test_film_tiles.py checks it for every world size the node supports.
//
// ncclReduceScatter wants the same element count from every rank, so the film is treated as world * rows_per_rank rows:
// rows_per_rank = ceil(H / world); the rows beyond H are the zero rows the film allocation carries behind the image
// (FILM_PAD_ROWS of them: enough for any world size up to FILM_PAD_ROWS, since ceil(H / N) * N <= H + N - 1).
// Rank r then owns rows [r * rows_per_rank, min((r + 1) * rows_per_rank, H)) -- possibly none for the last ranks of a very
// low film. Reference merge point: DRMLTProcess::processResult (src/integrators/drmlt/drmlt_proc.cpp:856-867) sums whole
// frames; tiling happens only here, at the reduction.
#pragma once
#include <cstddef>

#define FILM_PAD_ROWS 16 // zero rows behind the image = the largest world size of one exchange

struct FilmTile {
    int rows_per_rank; // ceil(H / world): the reduce-scatter's count is rows_per_rank * W * 3 floats per rank
    int lo, hi;        // rows [lo, hi) of the image this rank owns (hi clipped to H; lo == hi: nothing)
};

// false: `world` is not a valid partition of this film (world < 1, rank out of range, or more padding needed than allocated)
inline bool film_tile(int height, int rank, int world, FilmTile &t) {
    if (height < 1 || world < 1 || rank < 0 || rank >= world || world > FILM_PAD_ROWS) return false;
    t.rows_per_rank = (height + world - 1) / world;
    if ((long long) t.rows_per_rank * world > (long long) height + FILM_PAD_ROWS) return false;
    const long long lo = (long long) rank * t.rows_per_rank;
    t.lo = lo < height ? (int) lo : height;
    t.hi = t.lo + t.rows_per_rank < height ? t.lo + t.rows_per_rank : height;
    return true;
}

// floats of the film allocation: the image + the zero rows the reduce-scatter may read
inline size_t film_alloc_floats(int width, int height) { return (size_t) (height + FILM_PAD_ROWS) * (size_t) width * 3u; }

// Chains of a job split over `world` participants that share one seed pool (drmlt_seed_pool): rank r runs chains
// [r * per_rank, (r + 1) * per_rank) of the pool's world * per_rank seeds (drmlt_proc.cpp:869-883: one seed per work unit).
struct ChainRange { unsigned first, count, pool; };
inline ChainRange chain_range(unsigned per_rank, int rank, int world) {
    return ChainRange{(unsigned) rank * per_rank, per_rank, per_rank * (unsigned) world};
}
