"""Film exchange between the GPUs of one node (one process per GPU, torch.distributed over RCCL).

The reference merges every work unit's ImageBlock into one accumulation buffer under a mutex
(DRMLTProcess::processResult, drmlt_proc.cpp:856-867) and normalises once at the end with the
bootstrap estimate b (develop, :813-854; per-thread estimates are averaged, drmlt.cpp:531-546).
Chains are independent, so ranks run disjoint chain-id ranges with no data-path collective; the only
exchange is this one: sum of the W*H*3 fp32 films + mean of the per-rank b estimates.
"""


def exchange_film(film, b, dist, out=None, b_out=None):
    """All-reduce `film` (flat fp32 tensor) and `b` (1-element fp64 tensor) across the default group.

    Returns (summed film, mean b). `out` / `b_out` receive the result so the local accumulation
    buffers stay untouched (the local film keeps accumulating across renders)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if out is None:
        out = film.clone()
    else:
        out.copy_(film)
    if b_out is None:
        b_out = b.clone()
    else:
        b_out.copy_(b)
    if world > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        dist.all_reduce(b_out, op=dist.ReduceOp.SUM)
        b_out /= world
    return out, b_out


def chain_range(rank, chains_per_rank):
    """Chain ids of a rank: disjoint ranges => disjoint Philox streams (RNG addressing)."""
    return rank * chains_per_rank, (rank + 1) * chains_per_rank
