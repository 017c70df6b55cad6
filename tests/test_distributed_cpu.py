"""N > 1 path on CPU: two gloo ranks run disjoint chain ranges of the same render (the oracle stands in for
the device kernels; the exchange code is the one bench.py uses with RCCL) and combine their films."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    pkg, ob = entry.load_package(), entry.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    abi = pkg.abi
    chains = 128
    sd = pkg.scenes.cornell_c1(16)
    cfg = abi.make_config(type="orbital", max_depth=6, direct_samples=-1, work_units=chains, luminance_samples=4000,
                          sample_count=32)
    orc = ob.Oracle(abi, cfg, sd, 64)
    lo, hi = pkg.exchange.chain_range(rank, chains)
    b = orc.seed(0x5EED, chain_offset=lo)
    orc.run(16 * 16 * 32, 1)
    film = torch.from_numpy(orc.film().reshape(-1).copy())
    b_t = torch.tensor([b], dtype=torch.float64)
    total, b_mean = pkg.exchange.exchange_film(film, b_t, dist)
    # every rank must end with the same combined film and b, and its local buffer untouched
    gathered = [torch.empty_like(film) for _ in range(world)]
    dist.all_gather(gathered, film)
    bs = [torch.empty_like(b_t) for _ in range(world)]
    dist.all_gather(bs, b_t)
    # tiled exchange: this rank's rows of the developed image must equal the same rows of the all-reduce result
    tile, (r_lo, r_hi), b_t2 = pkg.exchange.exchange_film_tiled(film, b_t, dist, 16, 16)
    full = total.reshape(16, 16, 3).double()
    lum_mean = (full @ torch.tensor(pkg.exchange.LUM, dtype=torch.float64)).mean()
    want = (full * (b_mean.double() / lum_mean))[r_lo:r_hi]
    tile_err = float((tile.double() - want).abs().max() / want.abs().max())
    np.save(os.path.join(out_dir, "t%d.npy" % rank), np.array([tile_err, r_lo, r_hi, b_t2 - float(b_mean)]))
    np.save(os.path.join(out_dir, "r%d.npy" % rank),
            np.array([float((total - sum(gathered)).abs().max()), float(b_mean - sum(bs) / world),
                      float((gathered[0] - gathered[1]).abs().sum()), float(total.sum()), float(sum(x.sum() for x in gathered)),
                      float(lo), float(hi)]))
    dist.destroy_process_group()


def test_two_rank_film_exchange(tmp_path, ob):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    for r in (r0, r1):
        assert r[0] < 1e-5            # all-reduced film == sum of the per-rank films
        assert abs(r[1]) < 1e-12      # b is the mean of the per-rank bootstrap estimates
        assert r[2] > 0               # ranks ran different chains (disjoint chain ids => different films)
        assert r[3] == pytest.approx(r[4], rel=1e-6)
    t0, t1 = np.load(tmp_path / "t0.npy"), np.load(tmp_path / "t1.npy")
    assert t0[0] < 1e-5 and t1[0] < 1e-5 and abs(t0[3]) < 1e-12
    assert (t0[1], t0[2], t1[1], t1[2]) == (0, 8, 8, 16)       # rank r owns rows [8 r, 8 r + 8) of the 16-row film
    assert (r0[5], r0[6], r1[5], r1[6]) == (0, 128, 128, 256)
    assert r0[3] == pytest.approx(r1[3], rel=1e-7)


def test_exchange_is_identity_without_process_group(pkg):
    film = torch.arange(12, dtype=torch.float32)
    b = torch.tensor([0.5], dtype=torch.float64)
    out, bm = pkg.exchange.exchange_film(film, b, dist)
    assert torch.equal(out, film) and out.data_ptr() != film.data_ptr() and float(bm) == 0.5


def test_tiled_exchange_single_process(pkg):
    film = torch.rand(8 * 4 * 3, dtype=torch.float32)
    b = torch.tensor([0.25], dtype=torch.float64)
    tile, rows, bm = pkg.exchange.exchange_film_tiled(film, b, dist, 8, 4)
    assert rows == (0, 8) and bm == 0.25
    lum = (tile.double() @ torch.tensor(pkg.exchange.LUM, dtype=torch.float64)).mean()
    assert float(lum) == pytest.approx(0.25, rel=1e-6)       # develop: mean luminance of the image = b
    with pytest.raises(ValueError):
        pkg.exchange.exchange_film_tiled(film, b, dist, 7, 4)
