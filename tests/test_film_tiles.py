"""Row partition of the tiled film exchange (csrc/film_tiles.h through the C-ABI's drmlt_film_tile -- no GPU needed):
every world size a node supports, film heights that the world size does and does not divide, the padding bound of the
film allocation, and the refusals. This is the arithmetic drmlt_node.cpp's reduce-scatter counts, tile offsets and
develop ranges are made of (reference merge point: DRMLTProcess::processResult, drmlt_proc.cpp:856-867)."""
import pytest

FILM_PAD_ROWS = 16


@pytest.mark.parametrize("world", list(range(1, 9)) + [16])
def test_tiles_partition_the_film(pkg, native_lib, world):
    for height in (1, 2, 7, 13, 16, 63, 64, 65, 100, 511, 512, 513, 2047, 2048):
        tiles = [pkg.binding.film_tile(height, r, world) for r in range(world)]
        rows = -(-height // world)                                  # ceil(H / N), independently
        assert all(t[2] == rows for t in tiles)
        assert rows * world <= height + FILM_PAD_ROWS              # what the film allocation has to hold
        assert rows * world - height < world                       # never more than N - 1 rows of padding
        covered = []
        for r, (lo, hi, _) in enumerate(tiles):
            assert lo == min(r * rows, height) and hi == min(lo + rows, height) and 0 <= lo <= hi <= height
            covered.extend(range(lo, hi))
        assert covered == list(range(height))                      # disjoint, ordered, complete


def test_config4_tiles(pkg, native_lib):
    # BASELINE configs[3]: 2048 rows over 8 GPUs -> 256-row tiles, no padding
    assert [pkg.binding.film_tile(2048, r, 8)[:2] for r in range(8)] == [(256 * r, 256 * (r + 1)) for r in range(8)]


def test_refusals(pkg, native_lib):
    for height, rank, world in ((64, 0, 0), (64, 2, 2), (64, -1, 2), (64, 0, 17), (0, 0, 1)):
        with pytest.raises(pkg.DrmltError):
            pkg.binding.film_tile(height, rank, world)
