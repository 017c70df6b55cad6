"""The film exchange on the device buffer over RCCL (world size 1 here: one GPU per box; the N > 1 arithmetic is
covered by the gloo test). Checks the zero-copy view of the film, both exchange flavours and the develop identity."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_exchange_on_device_film(pkg, native_lib):
    import torch
    import torch.distributed as dist
    sd = pkg.scenes.cornell_c2(64)
    cfg = pkg.abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=4096, sample_count=16,
                              luminance_samples=40960)
    ctx = pkg.Context(cfg, sd)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    b = ctx.seed(7)
    ctx.run(64 * 64 * 16)
    torch.cuda.synchronize()

    class Dev:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}

    film = torch.as_tensor(Dev(ctx.film_device_ptr(), 64 * 64 * 3), device="cuda")
    np.testing.assert_array_equal(film.cpu().numpy().reshape(64, 64, 3), ctx.film())   # zero-copy view of the film
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        b_t = torch.tensor([b], dtype=torch.float64, device="cuda")
        total, b_mean = pkg.exchange.exchange_film(film, b_t, dist)
        assert torch.equal(total, film) and float(b_mean) == b
        tile, rows, bm = pkg.exchange.exchange_film_tiled(film, b_t, dist, 64, 64)
        assert rows == (0, 64) and bm == b
        np.testing.assert_allclose(tile.cpu().numpy(), ctx.develop(), rtol=2e-5, atol=1e-7)  # same develop as the C-ABI
    finally:
        dist.destroy_process_group()
