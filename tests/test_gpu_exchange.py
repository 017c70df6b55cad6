"""The film exchange on the device buffer, issued from C++ inside libdrmlt_amd.so (world size 1 here: one GPU per box;
the N > 1 arithmetic is covered on CPU by tests/test_distributed_cpu.py and tests/test_film_tiles.py): the zero-copy
view of the film, what the communicator reports about itself, and the gating of the rank -> device test hook."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg(pkg, n):
    return pkg.abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=n, sample_count=16, luminance_samples=40960)


def test_film_view_and_communicator_report(pkg, native_lib):
    import torch
    sd = pkg.scenes.cornell_c2(64)
    ctx = pkg.Context(_cfg(pkg, 4096), sd)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    b = ctx.seed(7)
    ctx.run(64 * 64 * 16)
    torch.cuda.synchronize()

    class Dev:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}

    film = torch.as_tensor(Dev(ctx.film_device_ptr(), 64 * 64 * 3), device="cuda")
    np.testing.assert_array_equal(film.cpu().numpy().reshape(64, 64, 3), ctx.film())   # zero-copy view of the film
    with pytest.raises(pkg.DrmltError, match="needs a communicator"):
        ctx.comm_info()
    ctx.comm_init(pkg.comm_unique_id(), 0, 1)
    assert ctx.comm_info() == (1, 0)                                    # ncclCommCount / ncclCommUserRank
    tile, rows, bm = ctx.exchange_tiled(b)
    assert rows == pkg.binding.film_tile(64, 0, 1)[:2] == (0, 64) and bm == b
    np.testing.assert_allclose(tile, ctx.develop(), rtol=2e-5, atol=1e-7)
    ctx.close()


def test_rank_to_device_hook_is_gated_and_validated(pkg, native_lib, monkeypatch):
    sd = pkg.scenes.cornell_c2(16)
    monkeypatch.delenv("DRMLT_TEST_HOOKS", raising=False)
    monkeypatch.setenv("DRMLT_NODE_DEVICES", "0,0,0")                   # a stale variable without the switch: ignored
    node = pkg.Node(_cfg(pkg, 256), sd, device_mask=1)
    assert node.device_count == 1
    node.close()
    monkeypatch.setenv("DRMLT_TEST_HOOKS", "1")
    node = pkg.Node(_cfg(pkg, 256), sd, device_mask=1)
    assert node.device_count == 3
    assert [pkg.Context.comm_info_of(node, r) for r in range(3)] == [(3, 0), (3, 1), (3, 2)]   # loopback ranks
    node.close()
    for bad in ("0,x", "0,99", "-1", "0;1"):
        monkeypatch.setenv("DRMLT_NODE_DEVICES", bad)
        with pytest.raises(pkg.DrmltError, match="DRMLT_NODE_DEVICES"):
            pkg.Node(_cfg(pkg, 256), sd, device_mask=1)
