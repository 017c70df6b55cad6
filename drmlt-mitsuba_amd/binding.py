"""ctypes host binding of libdrmlt_amd.so (the C-ABI of include/drmlt_abi.h).

`Context` mirrors the call sequence of the reference's DRMLT::render
(src/integrators/drmlt/drmlt.cpp:393-611): create -> seed -> run -> develop. There is no
CPU fallback: if the HIP library is missing or no GPU is visible, construction raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("DRMLT_LIBRARY") or os.path.join(_HERE, "libdrmlt_amd.so")  # DRMLT_LIBRARY: another build of the same ABI (codegen experiments)
_lib = None

SPLAT_DTYPE = np.dtype([("luminance", "<f4"), ("x", "<f4"), ("y", "<f4"), ("rgb", "<f4", (3,)),
                        ("n_dims", "<i4"), ("n_rays", "<i4")])

# every symbol include/drmlt_abi.h declares
ABI_SYMBOLS = (
    "drmlt_create", "drmlt_seed", "drmlt_run", "drmlt_develop", "drmlt_stats_get", "drmlt_eval_paths",
    "drmlt_film_read", "drmlt_film_clear", "drmlt_film_device_ptr", "drmlt_set_luminance", "drmlt_set_stream",
    "drmlt_kernel_time", "drmlt_render_pt", "drmlt_chain_state", "drmlt_last_error", "drmlt_abi_version",
    "drmlt_destroy", "drmlt_set_importance_map", "drmlt_luminance_map", "drmlt_eval_lists",
    "drmlt_seed_pool", "drmlt_comm_unique_id", "drmlt_comm_init", "drmlt_exchange_tiled",
    "drmlt_node_create", "drmlt_node_seed", "drmlt_node_run", "drmlt_node_develop", "drmlt_node_stats_get",
    "drmlt_node_set_importance_map", "drmlt_node_device_count", "drmlt_node_context", "drmlt_node_last_error",
    "drmlt_node_destroy", "drmlt_bootstrap_luminances", "drmlt_seed_indices", "drmlt_comm_info", "drmlt_film_tile",
)


class DrmltError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("drmlt error %d: %s" % (code, msg))
        self.code = code


def library_path():
    return _LIB_PATH


def build_native(force=False):
    """Compile the HIP kernels + C-ABI for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    args = ["make", "-C", src, "-j%d" % min(4, os.cpu_count() or 1)]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise DrmltError(abi.E_DEVICE, "libdrmlt_amd.so is not built (run __graft_entry__.build()); "
                                       "the DRMLT path has no CPU fallback")
    L = C.CDLL(_LIB_PATH)
    L.drmlt_create.restype = C.c_void_p
    L.drmlt_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    L.drmlt_seed.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
    L.drmlt_run.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.drmlt_develop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.drmlt_stats_get.argtypes = [C.c_void_p, C.c_void_p]
    L.drmlt_eval_paths.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.drmlt_eval_lists.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
    L.drmlt_film_read.argtypes = [C.c_void_p, C.c_void_p]
    L.drmlt_film_clear.argtypes = [C.c_void_p]
    L.drmlt_film_device_ptr.restype = C.c_void_p
    L.drmlt_film_device_ptr.argtypes = [C.c_void_p]
    L.drmlt_set_luminance.argtypes = [C.c_void_p, C.c_double]
    L.drmlt_set_importance_map.argtypes = [C.c_void_p, C.c_void_p]
    L.drmlt_luminance_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.drmlt_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.drmlt_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
    L.drmlt_render_pt.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
    L.drmlt_chain_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.drmlt_last_error.restype = C.c_char_p
    L.drmlt_last_error.argtypes = [C.c_void_p]
    L.drmlt_abi_version.restype = C.c_uint32
    L.drmlt_destroy.argtypes = [C.c_void_p]
    L.drmlt_seed_pool.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    L.drmlt_comm_unique_id.argtypes = [C.c_char_p]
    L.drmlt_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    L.drmlt_exchange_tiled.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.drmlt_node_create.restype = C.c_void_p
    L.drmlt_node_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]
    L.drmlt_node_seed.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
    L.drmlt_node_run.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.drmlt_node_develop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.drmlt_node_stats_get.argtypes = [C.c_void_p, C.c_void_p]
    L.drmlt_node_set_importance_map.argtypes = [C.c_void_p, C.c_void_p]
    L.drmlt_node_device_count.argtypes = [C.c_void_p]
    L.drmlt_node_context.restype = C.c_void_p
    L.drmlt_node_context.argtypes = [C.c_void_p, C.c_int]
    L.drmlt_node_last_error.restype = C.c_char_p
    L.drmlt_node_last_error.argtypes = [C.c_void_p]
    L.drmlt_node_destroy.argtypes = [C.c_void_p]
    L.drmlt_bootstrap_luminances.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    L.drmlt_seed_indices.argtypes = [C.c_void_p, C.c_void_p]
    L.drmlt_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.drmlt_film_tile.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    _lib = L
    return L


class Context:
    def __init__(self, cfg, scene_data, device=0):
        self.L = load_library()
        self.cfg = cfg
        self.scene_data = scene_data
        self._scene = scene_data.struct()
        err = C.create_string_buffer(512)
        self.h = self.L.drmlt_create(C.byref(cfg), C.byref(self._scene), device, err, 512)
        if not self.h:
            raise DrmltError(abi.E_INVALID, err.value.decode())
        self.width, self.height = scene_data.camera.width, scene_data.camera.height
        self._cb = None

    def close(self):
        if getattr(self, "h", None):
            self.L.drmlt_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise DrmltError(rc, self.L.drmlt_last_error(self.h).decode())

    # -- DRMLT::render sequence
    def seed(self, seed, chain_offset=0):
        b = C.c_double()
        self._chk(self.L.drmlt_seed(self.h, seed, chain_offset, C.byref(b)))
        return b.value

    def bootstrap_luminances(self, seed, stream, n):
        out = np.empty(n, dtype=np.float32)
        self._chk(self.L.drmlt_bootstrap_luminances(self.h, seed, stream, n, out.ctypes.data))
        return out

    def seed_indices(self):
        out = np.empty(self.cfg.work_units if self.cfg.work_units > 0 else self.stats().n_chains, dtype=np.uint32)
        self._chk(self.L.drmlt_seed_indices(self.h, out.ctypes.data))
        return out

    def seed_pool(self, seed, first_chain, pool_chains):
        """Seeds [first_chain, first_chain + work_units) of ONE pool drawn for `pool_chains` chains."""
        b = C.c_double()
        self._chk(self.L.drmlt_seed_pool(self.h, seed, first_chain, pool_chains, C.byref(b)))
        return b.value

    # -- film exchange over RCCL, called from C++ inside the library (one process per GPU)
    def comm_init(self, unique_id, rank, world):
        self._chk(self.L.drmlt_comm_init(self.h, unique_id, rank, world))

    def exchange_tiled(self, b, want_tile=True, wait=True):
        """reduce-scatter(sum) of the film + scalar all-reduce + develop of this rank's tile.
        Returns (tile [rows, W, 3] or None, (row_lo, row_hi), mean b). wait=False (with want_tile=False): everything is only
        ENQUEUED on the context's stream -- a render step's exchange; returns (None, None, b) without touching the host."""
        bb, lo, hi = C.c_double(b), C.c_int(), C.c_int()
        if not want_tile and not wait:
            self._chk(self.L.drmlt_exchange_tiled(self.h, C.byref(bb), None, None, None))
            return None, None, b
        buf = np.zeros((self.height, self.width, 3), dtype=np.float32) if want_tile else None
        self._chk(self.L.drmlt_exchange_tiled(self.h, C.byref(bb), buf.ctypes.data if want_tile else None,
                                              C.byref(lo), C.byref(hi)))
        tile = buf[:hi.value - lo.value] if want_tile else None
        return tile, (lo.value, hi.value), bb.value

    def comm_info(self):
        """(ranks, rank) as the communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        n, r = C.c_int(), C.c_int()
        self._chk(self.L.drmlt_comm_info(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    @staticmethod
    def comm_info_of(node, rank):
        """comm_info of rank `rank` of a Node (borrowed context)."""
        n, r = C.c_int(), C.c_int()
        h = node.L.drmlt_node_context(node.h, rank)
        rc = node.L.drmlt_comm_info(h, C.byref(n), C.byref(r))
        if rc != 0:
            raise DrmltError(rc, node.L.drmlt_last_error(h).decode())
        return n.value, r.value

    def run(self, total_mutations, stop=None, progress=None):
        cb = abi.PROGRESS_CB(lambda d, t, u: progress(d, t)) if progress else None
        self._cb = cb
        stop_p = C.cast(C.pointer(stop), C.c_void_p) if stop is not None else None
        self._chk(self.L.drmlt_run(self.h, total_mutations, stop_p, C.cast(cb, C.c_void_p) if cb else None, None))

    def develop(self, direct=None):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        d = None if direct is None else np.ascontiguousarray(direct, dtype=np.float32).ctypes.data
        self._chk(self.L.drmlt_develop(self.h, d, out.ctypes.data))
        return out

    def stats(self):
        s = abi.Stats()
        self._chk(self.L.drmlt_stats_get(self.h, C.byref(s)))
        return s

    # -- PathSampler::sampleSplats on explicit PSS points
    def eval_paths(self, u):
        u = np.ascontiguousarray(u, dtype=np.float32)
        n, dim = u.shape
        out = (abi.Splat * n)()
        self._chk(self.L.drmlt_eval_paths(self.h, u.ctypes.data, n, dim, out))
        return np.frombuffer(out, dtype=SPLAT_DTYPE).copy()

    def eval_lists_bdpt(self, u_sensor, u_emitter, u_direct=None):
        """technique=bdpt: rows [lum, hasMain, px, py, r, g, b, nMore, nDims, nRays, nMore x (px, py, r, g, b)].
        Points are [sensor S | emitter E | direct Dd]; u_direct is needed with directSampling=true (the default)."""
        rr = self.cfg.max_depth + 1 - max(self.cfg.rr_depth, 0)
        S = 2 * (self.cfg.max_depth + 1) + max(rr, 0); S += S & 1
        E = 2 * self.cfg.max_depth + max(rr - 1, 0); E += E & 1
        Dd = 0 if self.cfg.no_direct_sampling else 2 * (2 * self.cfg.max_depth - 1)
        us, ue = np.asarray(u_sensor, dtype=np.float32), np.asarray(u_emitter, dtype=np.float32)
        n = us.shape[0]
        u = np.zeros((n, S + E + Dd), dtype=np.float32)
        u[:, :min(S, us.shape[1])] = us[:, :S]
        u[:, S:S + min(E, ue.shape[1])] = ue[:, :E]
        if Dd:
            if u_direct is None:
                raise ValueError("directSampling=true: the direct sampler's components are part of the point")
            ud = np.asarray(u_direct, dtype=np.float32)
            u[:, S + E:S + E + min(Dd, ud.shape[1])] = ud[:, :Dd]
        stride = 10 + 5 * (self.cfg.max_depth + 1)
        out = np.zeros((n, stride), dtype=np.float32)
        self._chk(self.L.drmlt_eval_lists(self.h, u.ctypes.data, n, S + E + Dd, out.ctypes.data, stride))
        return out

    def eval_paths_mmlt(self, depth, u_sensor, u_emitter, u_direct):
        """technique=mmlt: points are [sensor S | emitter E | direct | depth]; returns (splats, (s, t))."""
        S, E = 2 * (self.cfg.max_depth + 1), 2 * self.cfg.max_depth
        us, ue = np.asarray(u_sensor, dtype=np.float32), np.asarray(u_emitter, dtype=np.float32)
        n = us.shape[0]
        u = np.zeros((n, S + E + 2), dtype=np.float32)
        u[:, :min(S, us.shape[1])] = us[:, :S]
        u[:, S:S + min(E, ue.shape[1])] = ue[:, :E]
        u[:, S + E] = np.asarray(u_direct, dtype=np.float32)
        u[:, S + E + 1] = depth
        sp = self.eval_paths(u)
        st = np.stack([(sp["n_dims"] >> 8) & 0xff, (sp["n_dims"] >> 16) & 0xff], axis=1)
        sp["n_dims"] &= 0xff
        return sp, st

    def film(self):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._chk(self.L.drmlt_film_read(self.h, out.ctypes.data))
        return out

    def film_clear(self):
        self._chk(self.L.drmlt_film_clear(self.h))

    def film_device_ptr(self):
        return self.L.drmlt_film_device_ptr(self.h)

    def set_luminance(self, b):
        self._chk(self.L.drmlt_set_luminance(self.h, b))

    def set_importance_map(self, lum_map):
        """Two-stage MLT: H x W luminance image of the first stage (None clears it). Call before seed()."""
        if lum_map is None:
            self._chk(self.L.drmlt_set_importance_map(self.h, None))
            return
        m = np.ascontiguousarray(lum_map, dtype=np.float32)
        assert m.shape == (self.height, self.width)
        self._chk(self.L.drmlt_set_importance_map(self.h, m.ctypes.data))

    def set_stream(self, stream_handle):
        self._chk(self.L.drmlt_set_stream(self.h, stream_handle))

    def kernel_time(self, reset=False):
        ms, n = C.c_double(), C.c_uint64()
        self._chk(self.L.drmlt_kernel_time(self.h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def render_pt(self, spp, seed=1):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._chk(self.L.drmlt_render_pt(self.h, spp, seed, out.ctypes.data))
        return out

    def chain_state(self, dim):
        n = self.stats().n_chains
        cur = (abi.Splat * n)()
        u = np.empty((n, dim), dtype=np.float32)
        self._chk(self.L.drmlt_chain_state(self.h, cur, u.ctypes.data, dim))
        return np.frombuffer(cur, dtype=SPLAT_DTYPE).copy(), u


def luminance_map(rgb_small, width, height):
    """BidirectionalUtils::mltLuminancePass tail: first-stage image -> full-size luminance image."""
    L = load_library()
    src = np.ascontiguousarray(rgb_small, dtype=np.float32)
    out = np.empty((height, width), dtype=np.float32)
    rc = L.drmlt_luminance_map(src.ctypes.data, src.shape[1], src.shape[0], width, height, out.ctypes.data)
    if rc != 0:
        raise DrmltError(rc, "drmlt_luminance_map failed")
    return out


def render_two_stage(cfg, scene_data, seed, size_reduction=16, device=0):
    """DRMLT::render with twoStage=true (drmlt.cpp:406-418): nested first stage on a film reduced by
    `size_reduction` with sample_count * size_reduction mutations per pixel and the film's default (gaussian)
    filter, then the full render weighted by its luminance image. Returns (image, importance_map, b)."""
    import copy
    cam = scene_data.camera
    small = copy.deepcopy(scene_data)
    w, h = max(1, cam.width // size_reduction), max(1, cam.height // size_reduction)
    small.camera.width, small.camera.height = w, h
    small.camera.filter, small.camera.filter_param = abi.FILTER_GAUSSIAN, 0.5
    cfg1 = abi.Config.from_buffer_copy(cfg)
    cfg1.sample_count = cfg.sample_count * size_reduction
    cfg1.direct_samples = -1 if cfg.direct_samples < 0 else cfg.direct_samples   # nested: no direct image is rendered
    cfg1.acceptance_map = 0
    with Context(cfg1, small, device) as first:
        first.seed(seed)
        first.run(w * h * cfg1.sample_count)
        lum = luminance_map(first.develop(), cam.width, cam.height)
    ctx = Context(cfg, scene_data, device)
    ctx.set_importance_map(lum)
    b = ctx.seed(seed)
    ctx.run(cam.width * cam.height * cfg.sample_count)
    return ctx.develop(), lum, b


def film_tile(height, rank, world):
    """(row_lo, row_hi, rows_per_rank) of the tiled exchange: the library's own arithmetic (csrc/film_tiles.h), no device."""
    L = load_library()
    lo, hi, rows = C.c_int(), C.c_int(), C.c_int()
    rc = L.drmlt_film_tile(height, rank, world, C.byref(lo), C.byref(hi), C.byref(rows))
    if rc != 0:
        raise DrmltError(rc, "no valid partition of %d rows over %d ranks (rank %d)" % (height, world, rank))
    return lo.value, hi.value, rows.value


def comm_unique_id():
    """ncclUniqueId of a new communicator (rank 0 creates it and hands it to the other ranks)."""
    L = load_library()
    buf = C.create_string_buffer(128)
    rc = L.drmlt_comm_unique_id(buf)
    if rc != 0:
        raise DrmltError(rc, "RCCL is not available")
    return buf.raw


class Node:
    """One process driving several GPUs (drmlt_node_*): what the Mitsuba plugin uses. `work_units` of the config is
    per device; chains, seeds and the film exchange (RCCL, in C++) are handled inside the library."""

    def __init__(self, cfg, scene_data, device_mask=1):
        self.L = load_library()
        self.cfg, self.scene_data = cfg, scene_data
        self._scene = scene_data.struct()
        err = C.create_string_buffer(512)
        self.h = self.L.drmlt_node_create(C.byref(cfg), C.byref(self._scene), device_mask, err, 512)
        if not self.h:
            raise DrmltError(abi.E_INVALID, err.value.decode())
        self.width, self.height = scene_data.camera.width, scene_data.camera.height
        self._cb = None

    def close(self):
        if getattr(self, "h", None):
            self.L.drmlt_node_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise DrmltError(rc, self.L.drmlt_node_last_error(self.h).decode())

    @property
    def device_count(self):
        return self.L.drmlt_node_device_count(self.h)

    def seed(self, seed):
        b = C.c_double()
        self._chk(self.L.drmlt_node_seed(self.h, seed, C.byref(b)))
        return b.value

    def run(self, total_mutations, stop=None, progress=None):
        cb = abi.PROGRESS_CB(lambda d, t, u: progress(d, t)) if progress else None
        self._cb = cb
        stop_p = C.cast(C.pointer(stop), C.c_void_p) if stop is not None else None
        self._chk(self.L.drmlt_node_run(self.h, total_mutations, stop_p, C.cast(cb, C.c_void_p) if cb else None, None))

    def develop(self, direct=None):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        d = np.ascontiguousarray(direct, dtype=np.float32) if direct is not None else None
        self._chk(self.L.drmlt_node_develop(self.h, d.ctypes.data if d is not None else None, out.ctypes.data))
        return out

    def set_importance_map(self, lum_map):
        m = np.ascontiguousarray(lum_map, dtype=np.float32) if lum_map is not None else None
        self._chk(self.L.drmlt_node_set_importance_map(self.h, m.ctypes.data if m is not None else None))

    def stats(self):
        st = abi.Stats()
        self._chk(self.L.drmlt_node_stats_get(self.h, C.byref(st)))
        return st

    def film(self, rank):
        """Raw film of one rank (test inspection)."""
        return self.context(rank).film()

    def context(self, rank):
        """Rank `rank`'s context, BORROWED (drmlt_node_context): film / chain inspection, kernel timing, render_pt. The node
        owns it; closing the returned object only forgets the handle."""
        return BorrowedContext(self, rank)


class BorrowedContext(Context):
    """A drmlt_ctx owned by a Node. Every inspection call of Context works on it; seed / run / develop belong to the node."""

    def __init__(self, node, rank):
        h = node.L.drmlt_node_context(node.h, rank)
        if not h:
            raise DrmltError(abi.E_INVALID, "node has no rank %d" % rank)
        self.L, self.h, self.cfg, self.scene_data = node.L, h, node.cfg, node.scene_data
        self.width, self.height = node.width, node.height
        self._node, self._cb = node, None   # keeps the owner alive

    def close(self):
        self.h = None
