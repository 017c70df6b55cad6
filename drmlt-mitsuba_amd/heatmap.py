"""Acceptance-map tooling (README "Acceptance Map" of the reference; tools/heatmap.py:14-24).

An acceptance map is the film of a render with `acceptanceMap=true`: R counts accepted first-stage small steps per
pixel, G accepted second-stage steps (drmlt_proc.cpp:697-709). The heat value is G / (R + G + eps); it is shown in
false colour between two clip values. numpy only (the reference's script needs pyexr + matplotlib)."""
import struct

import numpy as np

# anchor colours of a perceptually ordered dark-blue -> magenta -> orange -> yellow ramp (plasma-like)
_RAMP = np.array([
    [0.050, 0.030, 0.528], [0.255, 0.014, 0.615], [0.418, 0.001, 0.658], [0.563, 0.052, 0.642],
    [0.693, 0.165, 0.565], [0.798, 0.280, 0.470], [0.882, 0.393, 0.383], [0.949, 0.518, 0.296],
    [0.988, 0.652, 0.212], [0.988, 0.810, 0.145], [0.940, 0.975, 0.131]])


def stage_ratio(film, eps=1e-2):
    """G / (R + G + eps) per pixel."""
    film = np.asarray(film, dtype=np.float64)
    return film[..., 1] / (film[..., 0] + film[..., 1] + eps)


def false_colour(values, clip=(0.0, 1.0)):
    """Map values to RGB uint8 through the ramp, linearly between clip[0] and clip[1]."""
    lo, hi = float(clip[0]), float(clip[1])
    t = np.clip((np.asarray(values, dtype=np.float64) - lo) / max(hi - lo, 1e-12), 0.0, 1.0) * (len(_RAMP) - 1)
    i = np.minimum(t.astype(int), len(_RAMP) - 2)
    f = (t - i)[..., None]
    rgb = _RAMP[i] * (1.0 - f) + _RAMP[i + 1] * f
    return (rgb * 255.0 + 0.5).astype(np.uint8)


def read_pfm(path):
    """Colour PFM as written by host/drmlt_render (bottom-to-top scanlines) -> H x W x 3 float32, top-to-bottom."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"PF":
            raise ValueError("not a colour PFM file")
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(w * h * 12), dtype="<f4" if scale < 0 else ">f4")
    return data.reshape(h, w, 3)[::-1].astype(np.float32)


def write_pfm(path, img):
    img = np.asarray(img, dtype="<f4")
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (img.shape[1], img.shape[0]))
        f.write(img[::-1].tobytes())


def write_png(path, rgb8):
    """Minimal PNG writer (zlib + crc32 from the standard library)."""
    import zlib
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def heatmap(film, clip=(0.0, 1.0), eps=1e-2):
    return false_colour(stage_ratio(film, eps), clip)
