"""Acceptance-map tooling (reference: tools/heatmap.py:14-24, README "Acceptance Map")."""
import subprocess
import sys
import zlib

import numpy as np


def test_stage_ratio_and_colour_ramp(pkg):
    hm = pkg.heatmap
    film = np.zeros((2, 3, 3), dtype=np.float32)
    film[0, 0] = (3, 1, 0); film[0, 1] = (0, 5, 0); film[1, 2] = (7, 0, 0)
    r = hm.stage_ratio(film, eps=1e-2)
    assert r[0, 0] == np.float64(1) / (4 + 1e-2) and r[0, 1] == 5 / 5.01 and r[1, 2] == 0 and r[1, 0] == 0
    rgb = hm.false_colour(np.linspace(0, 1, 256), clip=(0, 1))
    lum = rgb.astype(float) @ [0.2126, 0.7152, 0.0722]
    assert np.all(np.diff(lum) > -1.0) and lum[-1] > lum[0] + 100       # perceptually increasing ramp
    np.testing.assert_array_equal(hm.false_colour([-1.0, 0.2], clip=(0.2, 0.8)), hm.false_colour([0.2, 0.2], (0.2, 0.8)))
    np.testing.assert_array_equal(hm.false_colour([0.8, 9.0], clip=(0.2, 0.8))[0], hm.false_colour([1.0], (0, 1))[0])


def test_pfm_roundtrip_and_cli(pkg, tmp_path):
    hm = pkg.heatmap
    rng = np.random.default_rng(0)
    film = rng.integers(0, 20, size=(5, 7, 3)).astype(np.float32)
    film[..., 2] = 0
    p = tmp_path / "acc.pfm"
    hm.write_pfm(str(p), film)
    np.testing.assert_array_equal(hm.read_pfm(str(p)), film)
    out = tmp_path / "acceptance-map.png"
    subprocess.check_call([sys.executable, "tools/acceptance_heatmap.py", "-t", str(p), "-c", "0.2", "0.8", "-o", str(out)])
    data = out.read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    i = data.index(b"IDAT")
    n = int.from_bytes(data[i - 4:i], "big")
    raw = zlib.decompress(data[i + 4:i + 4 + n])
    img = np.frombuffer(raw, dtype=np.uint8).reshape(5, 1 + 7 * 3)[:, 1:].reshape(5, 7, 3)
    np.testing.assert_array_equal(img, hm.heatmap(film, (0.2, 0.8)))
