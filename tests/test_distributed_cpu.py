"""N > 1 path on CPU: two gloo ranks re-enact what libdrmlt_amd.so does across GPUs (csrc/drmlt_node.cpp) --

  * ONE seed pool, rank r takes chains [r n, (r + 1) n) of it (drmlt_seed_pool; the oracle's seed_pool mirrors it),
  * every rank accumulates a full-frame film, no data-path collective,
  * film exchange = reduce-scatter(sum) over rows padded to world * rows_per_rank + a two-element all-reduce
    (total luminance, b) + develop of the rank's own tile,

with the row partition taken from the PRODUCT's own arithmetic (drmlt_film_tile -> csrc/film_tiles.h, callable without
a GPU) on a film whose height is NOT divisible by the world size. The oracle stands in for the device kernels; the
collectives are gloo's. What is asserted: the stitched tiles equal the image ONE context with 2 n chains develops --
i.e. the partition, the padding and the develop factor are right. The RCCL calls themselves only run on a GPU box
(tests/test_gpu_node.py: world size 1 and the loopback transport; N > 1 is the driver's scaling run)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LUM = (0.212671, 0.715160, 0.072169)
W, H, CHAINS, SPP = 12, 13, 96, 24     # 13 rows over 2 ranks: tiles of 7 and 6 rows, one padding row


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(chains):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    pkg, ob = entry.load_package(), entry.load_oracle()
    sd = pkg.scenes.cornell_c1(W)
    sd.camera.height = H
    cfg = pkg.abi.make_config(type="orbital", max_depth=6, direct_samples=-1, work_units=chains, luminance_samples=4000,
                              sample_count=SPP)
    return pkg, ob, sd, cfg


def exchange_tiled_gloo(pkg, film, b, rank, world):
    """drmlt_exchange_tiled's protocol (drmlt_node.cpp) with gloo collectives; returns (tile, (lo, hi))."""
    lo, hi, rows = pkg.binding.film_tile(H, rank, world)
    padded = torch.zeros(rows * world, W, 3, dtype=torch.float32)      # the film allocation's zero rows (FILM_PAD_ROWS)
    padded[:H] = torch.from_numpy(film)
    total = padded.clone()                                             # gloo has no reduce_scatter: all-reduce + slice
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    tile = total[rank * rows:(rank + 1) * rows][:hi - lo]
    lum = (tile.double() @ torch.tensor(LUM, dtype=torch.float64)).sum()
    scal = torch.stack([lum, torch.tensor(b, dtype=torch.float64)])
    dist.all_reduce(scal, op=dist.ReduceOp.SUM)                        # {sum of tile luminances, sum of the ranks' b}
    factor = (scal[1] / world) / (scal[0] / (W * H))                   # drmlt_proc.cpp:824-839
    return (tile.double() * factor).numpy(), (lo, hi)


def _worker(rank, world, port, out_dir):
    pkg, ob, sd, cfg = _setup(CHAINS)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = ob.Oracle(pkg.abi, cfg, sd, 64)
    b = orc.seed_pool(0x5EED, rank * CHAINS, world * CHAINS)
    orc.run(W * H * SPP // world, 1)                                   # the job's budget, split evenly (drmlt_node_run)
    tile, (lo, hi) = exchange_tiled_gloo(pkg, orc.film(), b, rank, world)
    np.save(os.path.join(out_dir, "tile%d.npy" % rank), tile)
    np.save(os.path.join(out_dir, "meta%d.npy" % rank), np.array([lo, hi, b]))
    dist.destroy_process_group()


def test_two_ranks_tile_a_film_whose_height_they_do_not_divide(tmp_path, pkg, ob):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    metas = [np.load(tmp_path / ("meta%d.npy" % r)) for r in range(world)]
    assert [(int(m[0]), int(m[1])) for m in metas] == [(0, 7), (7, 13)]
    assert metas[0][2] == metas[1][2]                                  # one pool: the same b on every rank
    stitched = np.concatenate([np.load(tmp_path / ("tile%d.npy" % r)) for r in range(world)], axis=0)
    assert stitched.shape == (H, W, 3)
    # the single-participant job: one context, all 2 n chains of the same pool
    _, _, sd, cfg = _setup(world * CHAINS)
    one = ob.Oracle(pkg.abi, cfg, sd, 64)
    b1 = one.seed_pool(0x5EED, 0, world * CHAINS)
    one.run(W * H * SPP, 1)
    want = one.develop().astype(np.float64)
    assert b1 == metas[0][2]
    assert np.abs(stitched - want).max() <= 2e-6 * want.max()          # fp32 films summed in a different order
    assert (stitched @ np.array(LUM)).mean() == pytest.approx(b1, rel=1e-6)   # develop: mean luminance of the image = b


def test_pool_slices_are_the_big_contexts_chains(pkg, ob):
    """Rank r's chains are chains [r n, (r + 1) n) of the pool: same seeds, same chain ids, same states after a run."""
    _, _, sd, cfg = _setup(CHAINS)
    _, _, _, cfg2 = _setup(2 * CHAINS)
    big = ob.Oracle(pkg.abi, cfg2, sd, 64)
    big.seed_pool(7, 0, 2 * CHAINS)
    big.run(2 * CHAINS * 8, 1)
    cur_big, u_big = big.chain_state(8)
    for r in range(2):
        part = ob.Oracle(pkg.abi, cfg, sd, 64)
        part.seed_pool(7, r * CHAINS, 2 * CHAINS)
        part.run(CHAINS * 8, 1)
        cur, u = part.chain_state(8)
        assert np.array_equal(u, u_big[r * CHAINS:(r + 1) * CHAINS])
        assert np.array_equal(cur["luminance"], cur_big["luminance"][r * CHAINS:(r + 1) * CHAINS])
