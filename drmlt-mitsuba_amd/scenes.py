"""Synthetic scenes (the reference ships none: README.md:155-160 links external downloads).

Every builder returns a SceneData whose .struct() is the flat drmlt_scene the C-ABI takes
(include/drmlt_abi.h). Geometry conventions are Mitsuba's: a `rectangle` is the local square
[-1,1]^2 in the z=0 plane with normal +z under a toWorld transform
(reference src/shapes/rectangle.cpp:80-112); cameras use Transform::lookAt and a horizontal fov.
All constants are deterministic.
"""
import ctypes as C
import math

import numpy as np

from . import abi


# ---- small transform helpers (4x4, column-vector convention) --------------------------------
def translate(x, y, z):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    return m


def scale(x, y=None, z=None):
    y = x if y is None else y
    z = x if z is None else z
    return np.diag([x, y, z, 1.0])


def rotate(axis, deg):
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    m = np.eye(4)
    if axis == "x":
        m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    elif axis == "y":
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    else:
        m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def lookat(origin, target, up):
    """Mitsuba's Transform::lookAt: columns (left, newUp, dir, origin)."""
    o, t, u = (np.asarray(v, dtype=np.float64) for v in (origin, target, up))
    d = t - o
    d /= np.linalg.norm(d)
    left = np.cross(u, d)
    left /= np.linalg.norm(left)
    new_up = np.cross(d, left)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, o
    return m


class SceneData:
    def __init__(self, name):
        self.name = name
        self.shapes, self.bsdfs, self.emitters = [], [], []
        self.camera = abi.Camera()
        self._keep = None

    # -- materials
    def diffuse(self, r, g=None, b=None):
        g = r if g is None else g
        b = r if b is None else b
        m = abi.Bsdf()
        m.type = abi.BSDF_DIFFUSE
        m.rgb[:] = (r, g, b)
        self.bsdfs.append(m)
        return len(self.bsdfs) - 1

    def dielectric(self, int_ior=1.5046, ext_ior=1.000277):
        m = abi.Bsdf()
        m.type = abi.BSDF_DIELECTRIC
        m.rgb[:] = (1, 1, 1)
        m.p[0], m.p[1] = int_ior, ext_ior
        self.bsdfs.append(m)
        return len(self.bsdfs) - 1

    def roughconductor(self, alpha=0.1, eta=(0.2004, 0.9240, 1.1022), k=(3.9129, 2.4528, 2.1421), ggx=False,
                       reflectance=(1, 1, 1)):
        m = abi.Bsdf()
        m.type = abi.BSDF_ROUGHCONDUCTOR
        m.rgb[:] = reflectance
        m.p[0] = alpha
        m.p[1], m.p[2], m.p[3] = eta
        m.p[4], m.p[5], m.p[6] = k
        m.p[7] = 1.0 if ggx else 0.0
        self.bsdfs.append(m)
        return len(self.bsdfs) - 1

    # -- shapes
    def _emit(self, radiance):
        e = abi.Emitter()
        e.type = abi.EMITTER_AREA
        e.shape = len(self.shapes)
        e.radiance[:] = radiance if hasattr(radiance, "__len__") else (radiance,) * 3
        e.sampling_weight = 1.0
        self.emitters.append(e)
        return len(self.emitters) - 1

    def rectangle(self, to_world, bsdf, radiance=None):
        s = abi.Shape()
        s.type = abi.SHAPE_RECTANGLE
        s.bsdf = bsdf
        s.emitter = self._emit(radiance) if radiance is not None else -1
        s.data[:] = np.asarray(to_world, dtype=np.float64)[:3, :].reshape(-1)
        self.shapes.append(s)
        return len(self.shapes) - 1

    def triangle(self, p0, p1, p2, bsdf, radiance=None):
        s = abi.Shape()
        s.type = abi.SHAPE_TRIANGLE
        s.bsdf = bsdf
        s.emitter = self._emit(radiance) if radiance is not None else -1
        s.data[:9] = list(p0) + list(p1) + list(p2)
        self.shapes.append(s)
        return len(self.shapes) - 1

    def sphere(self, center, radius, bsdf, radiance=None):
        s = abi.Shape()
        s.type = abi.SHAPE_SPHERE
        s.bsdf = bsdf
        s.emitter = self._emit(radiance) if radiance is not None else -1
        s.data[:4] = list(center) + [radius]
        self.shapes.append(s)
        return len(self.shapes) - 1

    def box(self, to_world, bsdf):
        """Mitsuba `cube`: [-1,1]^3 under to_world, 12 outward-facing triangles."""
        m = np.asarray(to_world, dtype=np.float64)
        v = np.array([[x, y, z, 1.0] for z in (-1, 1) for y in (-1, 1) for x in (-1, 1)]) @ m.T
        v = v[:, :3]
        quads = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]
        for a, b, c, d in quads:
            self.triangle(v[a], v[b], v[c], bsdf)
            self.triangle(v[a], v[c], v[d], bsdf)

    def set_camera(self, to_world, fov_x_deg, width, height, filt=abi.FILTER_BOX, filter_param=0.5,
                   near=1e-2, far=1e4):
        c = self.camera
        c.to_world[:] = np.asarray(to_world, dtype=np.float64).reshape(-1)
        c.fov_x_deg, c.near_clip, c.far_clip = fov_x_deg, near, far
        c.width, c.height, c.filter, c.filter_param = width, height, filt, filter_param

    def save(self, path):
        """Flat binary scene file read by the C++ host (host/drmlt_integrator.hpp: SceneFile::load)."""
        import struct as _st
        with open(path, "wb") as f:
            f.write(_st.pack("<8I", 0x4C4D5244, abi.ABI_VERSION, len(self.shapes), len(self.bsdfs), len(self.emitters),
                             C.sizeof(abi.Shape), C.sizeof(abi.Bsdf), C.sizeof(abi.Emitter)))
            for group in (self.shapes, self.bsdfs, self.emitters):
                for item in group:
                    f.write(bytes(item))
            f.write(bytes(self.camera))

    def struct(self):
        sh = (abi.Shape * len(self.shapes))(*self.shapes)
        bs = (abi.Bsdf * len(self.bsdfs))(*self.bsdfs)
        em = (abi.Emitter * max(1, len(self.emitters)))(*self.emitters)
        s = abi.Scene()
        s.struct_size = C.sizeof(abi.Scene)
        s.n_shapes, s.n_bsdfs, s.n_emitters = len(self.shapes), len(self.bsdfs), len(self.emitters)
        s.shapes = C.cast(sh, C.POINTER(abi.Shape))
        s.bsdfs = C.cast(bs, C.POINTER(abi.Bsdf))
        s.emitters = C.cast(em, C.POINTER(abi.Emitter))
        s.camera = self.camera
        self._keep = (sh, bs, em)
        return s


def _room(sd, white, red, green, walls=("floor", "ceiling", "back", "left", "right")):
    t = {
        "floor": (translate(0, -1, 0) @ rotate("x", -90), white),
        "ceiling": (translate(0, 1, 0) @ rotate("x", 90), white),
        "back": (translate(0, 0, -1), white),
        "left": (translate(-1, 0, 0) @ rotate("y", 90), red),
        "right": (translate(1, 0, 0) @ rotate("y", -90), green),
    }
    for w in walls:
        sd.rectangle(t[w][0], t[w][1])


def cornell_c1(res=256, filt=abi.FILTER_BOX):
    """floor + back wall diffuse quads + one area-light quad, pinhole fov 39 deg."""
    sd = SceneData("cornell_c1")
    grey = sd.diffuse(0.5)
    red = sd.diffuse(0.63, 0.065, 0.05)
    black = sd.diffuse(0.0)
    sd.rectangle(translate(0, -1, 0) @ rotate("x", -90), grey)
    sd.rectangle(translate(0, 0, -1), red)
    sd.rectangle(translate(0, 0.98, 0) @ rotate("x", 90) @ scale(0.25), black, radiance=15.0)
    sd.set_camera(lookat((0, 0, 3.9), (0, 0, 0), (0, 1, 0)), 39.0, res, res, filt,
                  0.5 if filt == abi.FILTER_BOX else 0.5)
    return sd


def cornell_c2(res=512, filt=abi.FILTER_BOX):
    """5 wall quads + tall and short box (24 triangles) + light quad."""
    sd = SceneData("cornell_c2")
    white = sd.diffuse(0.725, 0.71, 0.68)
    red = sd.diffuse(0.63, 0.065, 0.05)
    green = sd.diffuse(0.14, 0.45, 0.091)
    black = sd.diffuse(0.0)
    _room(sd, white, red, green)
    sd.box(translate(-0.33, -0.4, -0.3) @ rotate("y", 17) @ scale(0.3, 0.6, 0.3), white)
    sd.box(translate(0.33, -0.7, 0.3) @ rotate("y", -17) @ scale(0.3, 0.3, 0.3), white)
    sd.rectangle(translate(0, 0.995, 0) @ rotate("x", 90) @ scale(0.25), black, radiance=(17.0, 12.0, 4.0))
    sd.set_camera(lookat((0, 0, 3.9), (0, 0, 0), (0, 1, 0)), 39.3077, res, res, filt, 0.5)
    return sd


def glass_sphere(res=128):
    """Dielectric sphere above a diffuse floor inside the box, quad light (technique=path variant of C5)."""
    sd = SceneData("glass_sphere")
    white = sd.diffuse(0.725, 0.71, 0.68)
    red = sd.diffuse(0.63, 0.065, 0.05)
    green = sd.diffuse(0.14, 0.45, 0.091)
    black = sd.diffuse(0.0)
    glass = sd.dielectric(1.5, 1.0)
    _room(sd, white, red, green)
    sd.sphere((0.0, -0.55, 0.1), 0.3, glass)
    sd.rectangle(translate(0, 0.995, 0) @ rotate("x", 90) @ scale(0.15), black, radiance=40.0)
    sd.set_camera(lookat((0, 0, 3.9), (0, 0, 0), (0, 1, 0)), 39.3077, res, res, abi.FILTER_BOX, 0.5)
    return sd


def door_c3(res=256, ggx=False):
    """closed room, partition wall with a 5 % gap hiding a quad light, rough-conductor floor."""
    sd = SceneData("door_c3")
    white = sd.diffuse(0.725, 0.71, 0.68)
    red = sd.diffuse(0.63, 0.065, 0.05)
    green = sd.diffuse(0.14, 0.45, 0.091)
    black = sd.diffuse(0.0)
    copper = sd.roughconductor(alpha=0.1, ggx=ggx)
    sd.rectangle(translate(0, -1, 0) @ rotate("x", -90), copper)                      # floor: glossy metal
    _room(sd, white, red, green, walls=("ceiling", "back", "left", "right"))
    sd.rectangle(translate(0, 0, 1) @ rotate("y", 180), white)                        # front wall (room is closed)
    # partition at z = -0.4 leaving a 0.1-wide gap at the right wall (5 % of the room width); two one-sided faces
    sd.rectangle(translate(-0.05, 0, -0.4) @ scale(0.95, 1, 1), white)                # faces the camera room
    sd.rectangle(translate(-0.05, 0, -0.42) @ rotate("y", 180) @ scale(0.95, 1, 1), white)  # faces the light room
    sd.rectangle(translate(-0.3, 0.2, -0.99) @ scale(0.3), black, radiance=40.0)      # light on the back wall
    sd.set_camera(lookat((0, -0.2, 0.95), (0, -0.3, -0.4), (0, 1, 0)), 70.0, res, res, abi.FILTER_BOX, 0.5)
    return sd


def triangle_soup(n_tris=2000, res=128, seed=7):
    """Closed room filled with small random diffuse triangles: a scene large enough that the BVH matters."""
    rng = np.random.default_rng(seed)
    sd = SceneData("soup_%d" % n_tris)
    white = sd.diffuse(0.7)
    red = sd.diffuse(0.63, 0.065, 0.05)
    green = sd.diffuse(0.14, 0.45, 0.091)
    black = sd.diffuse(0.0)
    _room(sd, white, red, green)
    mats = [white, red, green]
    for i in range(n_tris):
        c = rng.uniform(-0.85, 0.85, 3)
        c[1] = rng.uniform(-0.95, 0.5)
        e1 = rng.normal(size=3) * 0.06
        e2 = rng.normal(size=3) * 0.06
        sd.triangle(c, c + e1, c + e2, mats[i % 3])
    sd.rectangle(translate(0, 0.995, 0) @ rotate("x", 90) @ scale(0.3), black, radiance=20.0)
    sd.set_camera(lookat((0, 0, 3.9), (0, 0, 0), (0, 1, 0)), 39.3077, res, res, abi.FILTER_BOX, 0.5)
    return sd


def caustic_c5(res=128):
    """SURVEY 8(d) C5 with the small SPHERE area light (the reference rejects point lights under mmlt, A11):
    dielectric sphere (eta 1.5, r 0.3) above the diffuse floor, emissive sphere r 0.06 under the ceiling, and a
    second, dimmer quad light so that emitter selection is exercised."""
    sd = SceneData("caustic_c5")
    white = sd.diffuse(0.725, 0.71, 0.68)
    red = sd.diffuse(0.63, 0.065, 0.05)
    green = sd.diffuse(0.14, 0.45, 0.091)
    black = sd.diffuse(0.0)
    glass = sd.dielectric(1.5, 1.0)
    _room(sd, white, red, green)
    sd.sphere((0.0, -0.55, 0.1), 0.3, glass)
    sd.sphere((0.25, 0.7, 0.0), 0.06, black, radiance=(300.0, 260.0, 200.0))
    sd.rectangle(translate(-0.6, 0.995, -0.4) @ rotate("x", 90) @ scale(0.1), black, radiance=12.0)
    sd.set_camera(lookat((0, 0, 3.9), (0, 0, 0), (0, 1, 0)), 39.3077, res, res, abi.FILTER_BOX, 0.5)
    return sd


SCENES = {"cornell_c1": cornell_c1, "cornell_c2": cornell_c2, "glass_sphere": glass_sphere, "door_c3": door_c3,
          "triangle_soup": triangle_soup, "caustic_c5": caustic_c5}
