"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes loader for oracle/liboracle.so (the CPU restatement of the reference's DRMLT path).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _native_name():
    """-march=native code must run on the CPU it was built for: tag the file with this host's instruction-set flags."""
    import hashlib
    flags = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = " ".join(sorted(line.split(":", 1)[1].split()))
                break
    except OSError:
        pass
    return "liboracle_native_%s.so" % hashlib.md5(flags.encode()).hexdigest()[:8]


def build(native=False):
    target = _native_name() if native else "liboracle.so"
    subprocess.check_call(["make", "-C", _HERE, target], stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, target)


_libs = {}


def lib(native=False):
    if native not in _libs:
        path = os.path.join(_HERE, _native_name() if native else "liboracle.so")
        if not os.path.exists(path):
            build(native)
        L = C.CDLL(path)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_last_error.argtypes = [C.c_void_p]
        L.oracle_eval_paths.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_seed.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
        L.oracle_seed_pool.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
        L.oracle_seed_indices.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_double)]
        L.oracle_run.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        L.oracle_film_read.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_develop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_stats_get.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_chain_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.oracle_render_pt.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p]
        L.oracle_bootstrap_lum.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_mmlt_render.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p]
        L.oracle_mmlt_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                       C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_philox.argtypes = [C.c_uint32] * 6 + [C.c_void_p]
        L.oracle_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_void_p]
        L.oracle_kernel_sample.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_uint64, C.c_uint32,
                                           C.c_void_p]
        L.oracle_kernel_pdf.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
        L.oracle_sampler_trace.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_uint64, C.c_uint32,
                                           C.c_uint32, C.c_int, C.c_uint32, C.c_uint32] + [C.c_void_p] * 7
        L.oracle_toy_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint64,
                                     C.c_uint32, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_toy_amap.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int,
                                      C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_toy_target.restype = C.c_double
        L.oracle_toy_target.argtypes = [C.c_double, C.c_double]
        L.oracle_film_put.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
        L.oracle_find_max_dim.argtypes = [C.c_int, C.c_int]
        L.oracle_set_importance_map.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_bdpt_render.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        L.oracle_bdpt_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.oracle_luminance_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_find_max_dim_mmlt.argtypes = [C.c_int]
        L.oracle_roughconductor.argtypes = [C.c_int, C.c_double] + [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 7
        _libs[native] = L
    return _libs[native]


class OracleError(RuntimeError):
    pass


class Oracle:
    """Mirrors drmlt_amd.Context so parity tests read the same on both sides."""

    def __init__(self, abi, cfg, scene_data, precision=64, native=False):
        self.abi = abi
        self.L = lib(native)
        self.cfg = cfg
        self.scene_data = scene_data
        self._scene = scene_data.struct()
        err = C.create_string_buffer(512)
        self.h = self.L.oracle_create(C.byref(cfg), C.byref(self._scene), precision, err, 512)
        if not self.h:
            raise OracleError(err.value.decode())
        self.width, self.height = scene_data.camera.width, scene_data.camera.height

    def close(self):
        if self.h:
            self.L.oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise OracleError("%d: %s" % (rc, self.L.oracle_last_error(self.h).decode()))

    def eval_paths(self, u):
        u = np.ascontiguousarray(u, dtype=np.float32)
        n, dim = u.shape
        out = (self.abi.Splat * n)()
        self._chk(self.L.oracle_eval_paths(self.h, u.ctypes.data, n, dim, out))
        return np.frombuffer(out, dtype=SPLAT_DTYPE).copy()

    def seed(self, seed, chain_offset=0):
        b = C.c_double()
        self._chk(self.L.oracle_seed(self.h, seed, chain_offset, C.byref(b)))
        return b.value

    def seed_with_indices(self, seed, indices, chain_offset=0, pool_chains=0):
        """Start the chains from explicit bootstrap sample indices (e.g. Context.seed_indices() of a device context)."""
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        assert idx.size == self.cfg.work_units
        b = C.c_double()
        self._chk(self.L.oracle_seed_indices(self.h, seed, chain_offset, pool_chains, idx.ctypes.data, C.byref(b)))
        return b.value

    def seed_indices(self):
        """Bootstrap sample indices of this context's chains after the last seed call (mirrors Context.seed_indices)."""
        out = np.empty(self.cfg.work_units, dtype=np.uint32)
        self.L.oracle_picked_seeds.argtypes = [C.c_void_p, C.c_void_p]
        self._chk(self.L.oracle_picked_seeds(self.h, out.ctypes.data))
        return out

    def seed_pool(self, seed, first_chain, pool_chains):
        """Seeds [first_chain, first_chain + work_units) of ONE pool drawn for `pool_chains` chains (mirrors Context.seed_pool)."""
        b = C.c_double()
        self._chk(self.L.oracle_seed_pool(self.h, seed, first_chain, pool_chains, C.byref(b)))
        return b.value

    def run(self, total_mutations, nthreads=1):
        self._chk(self.L.oracle_run(self.h, total_mutations, nthreads))

    def film(self):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._chk(self.L.oracle_film_read(self.h, out.ctypes.data))
        return out

    def develop(self, direct=None):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        d = None if direct is None else np.ascontiguousarray(direct, dtype=np.float32).ctypes.data
        self._chk(self.L.oracle_develop(self.h, d, out.ctypes.data))
        return out

    def stats(self):
        s = self.abi.Stats()
        self._chk(self.L.oracle_stats_get(self.h, C.byref(s)))
        return s

    def chain_state(self, dim):
        n = self.cfg.work_units
        cur = (self.abi.Splat * n)()
        u = np.empty((n, dim), dtype=np.float32)
        self._chk(self.L.oracle_chain_state(self.h, cur, u.ctypes.data, dim))
        return np.frombuffer(cur, dtype=SPLAT_DTYPE).copy(), u

    def render_pt(self, spp, seed=1, nthreads=1):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._chk(self.L.oracle_render_pt(self.h, spp, seed, nthreads, out.ctypes.data))
        return out

    def mmlt_render(self, depth, n, seed=1, light_image=True, nthreads=1):
        """Independent-sample image of the depth-`depth` paths with the multiplexed estimator (radiance units)."""
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        strat = np.zeros(depth + 2)
        self._chk(self.L.oracle_mmlt_render(self.h, depth, n, seed, int(light_image), nthreads, out.ctypes.data,
                                            strat.ctypes.data))
        return out, strat

    def mmlt_eval(self, depth, u_sensor, u_emitter, u_direct, light_image=True):
        us = np.ascontiguousarray(u_sensor, dtype=np.float32)
        ue = np.ascontiguousarray(u_emitter, dtype=np.float32)
        ud = np.ascontiguousarray(u_direct, dtype=np.float32)
        n, dim = us.shape
        assert ue.shape == us.shape and ud.shape == (n,)
        out = (self.abi.Splat * n)()
        st = np.zeros((n, 2), dtype=np.int32)
        self._chk(self.L.oracle_mmlt_eval(self.h, depth, int(light_image), us.ctypes.data, ue.ctypes.data,
                                          ud.ctypes.data, n, dim, out, st.ctypes.data))
        return np.frombuffer(out, dtype=SPLAT_DTYPE).copy(), st

    def bdpt_render(self, n, seed=1, nthreads=1):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._chk(self.L.oracle_bdpt_render(self.h, n, seed, nthreads, out.ctypes.data))
        return out

    def bdpt_eval(self, u_sensor, u_emitter, u_direct=None):
        """Rows: [lum, hasMain, px, py, r, g, b, nMore, nDims, nRays, nMore x (px, py, r, g, b)].
        u_direct: the direct sampler's components (directSampling = true), same shape."""
        us = np.ascontiguousarray(u_sensor, dtype=np.float32)
        ue = np.ascontiguousarray(u_emitter, dtype=np.float32)
        assert us.shape == ue.shape
        ud = np.ascontiguousarray(u_direct, dtype=np.float32) if u_direct is not None else None
        assert ud is None or ud.shape == us.shape
        n, dim = us.shape
        stride = 10 + 5 * (self.cfg.max_depth + 1)
        out = np.zeros((n, stride), dtype=np.float32)
        self._chk(self.L.oracle_bdpt_eval(self.h, us.ctypes.data, ue.ctypes.data, ud.ctypes.data if ud is not None else None, n, dim,
                                          out.ctypes.data, stride))
        return out

    def set_importance_map(self, lum_map):
        if lum_map is None:
            self._chk(self.L.oracle_set_importance_map(self.h, None))
            return
        self._imp = np.ascontiguousarray(lum_map, dtype=np.float32)
        assert self._imp.shape == (self.height, self.width)
        self._chk(self.L.oracle_set_importance_map(self.h, self._imp.ctypes.data))

    def bootstrap_lum(self, seed, stream, n):
        out = np.empty(n, dtype=np.float32)
        self._chk(self.L.oracle_bootstrap_lum(self.h, seed, stream, n, out.ctypes.data))
        return out


SPLAT_DTYPE = np.dtype([("luminance", "<f4"), ("x", "<f4"), ("y", "<f4"), ("rgb", "<f4", (3,)),
                        ("n_dims", "<i4"), ("n_rays", "<i4")])


def philox(k0, k1, c0, c1, c2, c3):
    out = np.zeros(4, dtype=np.uint32)
    lib().oracle_philox(k0, k1, c0, c1, c2, c3, out.ctypes.data)
    return out


def uniforms(seed, chain, tag, major, idx0, n):
    out = np.zeros(n, dtype=np.float32)
    lib().oracle_uniforms(seed, chain, tag, major, idx0, n, out.ctypes.data)
    return out


def kernel_sample(kind, p0, p1, precision, seed, n):
    out = np.zeros(n, dtype=np.float64)
    lib().oracle_kernel_sample(kind, p0, p1, precision, seed, n, out.ctypes.data)
    return out


def kernel_pdf(kind, p0, p1, precision, du):
    du = np.ascontiguousarray(du, dtype=np.float64)
    pdf = np.zeros_like(du)
    logpdf = np.zeros_like(du)
    lib().oracle_kernel_pdf(kind, p0, p1, precision, du.size, du.ctypes.data, pdf.ctypes.data, logpdf.ctypes.data)
    return pdf, logpdf


def sampler_trace(type_, sigma, scale_second, precision, seed, chain, mutation, large, x, used):
    x = np.ascontiguousarray(x, dtype=np.float64)
    dim = x.size
    y, z, ystar = (np.zeros(used) for _ in range(3))
    acc1, acc2 = np.zeros(dim), np.zeros(dim)
    ratio = C.c_double(1.0)
    rc = lib().oracle_sampler_trace(type_, sigma, scale_second, precision, seed, chain, mutation, int(large), dim,
                                    used, x.ctypes.data, y.ctypes.data, z.ctypes.data, ystar.ctypes.data,
                                    C.addressof(ratio), acc1.ctypes.data, acc2.ctypes.data)
    if rc != 0:
        raise OracleError("sampler_trace failed")
    return dict(y=y, z=z, ystar=ystar, ratio=ratio.value, acc1=acc1, acc2=acc2)


def toy_run(abi, type_, mixture, timid, p_large, sigma, scale_second, seed, n_chains, n_mut, w, h):
    hist = np.zeros((h, w), dtype=np.float64)
    st = abi.Stats()
    rc = lib().oracle_toy_run(type_, int(mixture), int(timid), p_large, sigma, scale_second, seed, n_chains, n_mut,
                              w, h, hist.ctypes.data, C.addressof(st))
    if rc != 0:
        raise OracleError("toy_run failed: %d" % rc)
    return hist, st


def toy_amap(type_, p_large, sigma, scale_second, seed, n_chains, n_mut, w, h):
    """Acceptance map (H x W x 3) of toy-target chains + their initial states (n_chains x 2)."""
    rgb = np.zeros((h, w, 3), dtype=np.float64)
    x0 = np.zeros((n_chains, 2), dtype=np.float64)
    rc = lib().oracle_toy_amap(type_, p_large, sigma, scale_second, seed, n_chains, n_mut, w, h, rgb.ctypes.data, x0.ctypes.data)
    if rc != 0:
        raise OracleError("toy_amap failed: %d" % rc)
    return rgb, x0


def toy_target(x, y):
    return lib().oracle_toy_target(x, y)


def select_seeds(lum, seed, stream, n_seeds):
    """generateSeeds' luminance-proportional picks (pathsampler.cpp:936-957) on the given luminance samples; sorted indices."""
    lum = np.ascontiguousarray(lum, dtype=np.float32)
    out = np.empty(n_seeds, dtype=np.uint32)
    L = lib()
    L.oracle_select_seeds.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    L.oracle_select_seeds.restype = None
    L.oracle_select_seeds(lum.ctypes.data, lum.size, seed, stream, n_seeds, out.ctypes.data)
    return out


def film_put(w, h, filt, param, xy, rgb):
    xy = np.ascontiguousarray(xy, dtype=np.float32)
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.zeros((h, w, 3), dtype=np.float32)
    lib().oracle_film_put(w, h, filt, param, xy.shape[0], xy.ctypes.data, rgb.ctypes.data, out.ctypes.data)
    return out


def roughconductor(ggx, alpha, eta, k, wi, wo=None, sxy=None):
    """eval/pdf for explicit directions and/or samples for explicit (sx, sy); local frame, double precision."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    eta, k, wi = f(eta), f(k), f(wi)
    out = {}
    n = len(wo) if wo is not None else len(sxy)
    ev, pdf = np.zeros((n, 3)), np.zeros(n)
    swo, sw, spdf = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n)
    wo_ = f(wo) if wo is not None else None
    sxy_ = f(sxy) if sxy is not None else None
    lib().oracle_roughconductor(int(ggx), alpha, eta.ctypes.data, k.ctypes.data, wi.ctypes.data, n,
                                wo_.ctypes.data if wo_ is not None else None, sxy_.ctypes.data if sxy_ is not None else None,
                                ev.ctypes.data, pdf.ctypes.data, swo.ctypes.data, sw.ctypes.data, spdf.ctypes.data)
    return dict(eval=ev, pdf=pdf, wo=swo, weight=sw, spdf=spdf)


def luminance_map(rgb_small, width, height):
    src = np.ascontiguousarray(rgb_small, dtype=np.float32)
    out = np.empty((height, width), dtype=np.float32)
    lib().oracle_luminance_map(src.ctypes.data, src.shape[1], src.shape[0], width, height, out.ctypes.data)
    return out
