"""ImageBlock::put restated (imageblock.h:150-216, rfilter.cpp:37-55): an independent numpy re-derivation."""
import math

import numpy as np
import pytest


def numpy_film(w, h, gauss, param, xy, rgb):
    res = 31
    radius = 4 * param if gauss else param + 1e-5
    vals = np.zeros(res + 1)
    for i in range(res):
        x = radius * i / res
        if gauss:
            a = -1 / (2 * param * param)
            vals[i] = max(0.0, math.exp(a * x * x) - math.exp(a * radius * radius))
        else:
            vals[i] = 1.0 if abs(x) <= radius else 0.0
    vals[:res] /= vals[:res].sum() * 2 * radius / res
    scale = res / radius
    border = math.ceil(radius - 0.5)
    img = np.zeros((h + 2 * border, w + 2 * border, 3))
    for (px, py), v in zip(xy, rgb):
        if not np.all(np.isfinite(v)) or np.any(v < 0):
            continue
        posx, posy = px - 0.5 + border, py - 0.5 + border
        for y in range(max(math.ceil(posy - radius), 0), min(math.floor(posy + radius), h + 2 * border - 1) + 1):
            wy = vals[min(int(abs((y - posy) * scale)), res)]
            for x in range(max(math.ceil(posx - radius), 0), min(math.floor(posx + radius), w + 2 * border - 1) + 1):
                img[y, x] += vals[min(int(abs((x - posx) * scale)), res)] * wy * v
    return img[border:border + h, border:border + w]


@pytest.mark.parametrize("gauss,param", [(False, 0.5), (True, 0.5)])
def test_film_put_matches_numpy(ob, gauss, param):
    rng = np.random.default_rng(4)
    w, h, n = 12, 9, 400
    xy = np.stack([rng.uniform(-1, w + 1, n), rng.uniform(-1, h + 1, n)], 1).astype(np.float32)
    xy[:8] = [[0, 0], [w, h], [3.0, 4.0], [3.5, 4.5], [0.25, 8.75], [11.999, 0.001], [-0.4, 2], [5, h + 0.4]]
    rgb = rng.uniform(0, 2, (n, 3)).astype(np.float32)
    rgb[10] = [np.nan, 1, 1]
    rgb[11] = [1, -1, 1]
    rgb[12] = [np.inf, 0, 0]
    got = ob.film_put(w, h, 1 if gauss else 0, param, xy, rgb)
    want = numpy_film(w, h, gauss, param, xy.astype(np.float64), rgb.astype(np.float64))
    assert np.allclose(got, want, rtol=2e-4, atol=1e-5)


def test_box_splat_lands_in_one_pixel(ob):
    out = ob.film_put(4, 4, 0, 0.5, [[2.3, 1.7]], [[1.0, 2.0, 3.0]])
    assert np.count_nonzero(out[..., 0]) == 1 and out[1, 2, 0] == pytest.approx(0.99998 ** 2, rel=1e-4)
