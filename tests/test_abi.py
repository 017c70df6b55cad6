"""The drop-in boundary: libdrmlt_amd.so loads without a GPU, exports every symbol include/drmlt_abi.h
declares, the ctypes mirror matches the C struct layout, and the product path refuses to run without its
HIP device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "drmlt_abi.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(drmlt_[a-z_]+)\s*\(", src)
    return sorted(set(n for n in names if n not in ("drmlt_progress_cb",)))


def test_library_exports_every_declared_symbol(pkg, native_lib):
    names = declared_functions()
    assert len(names) >= 15
    assert set(names) == set(pkg.binding.ABI_SYMBOLS)
    for n in names:
        assert hasattr(native_lib, n), n
    assert native_lib.drmlt_abi_version() == 4   # 2: + pool seeding, RCCL exchange, drmlt_node_*; 3: + drmlt_comm_info, drmlt_film_tile; 4: + seed_rule, work_units_rule (no getenv in result-affecting decisions)


def test_struct_layouts_match_the_header(abi):
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "drmlt_abi.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(drmlt_config), sizeof(drmlt_shape), sizeof(drmlt_bsdf),
         sizeof(drmlt_emitter), sizeof(drmlt_camera), sizeof(drmlt_scene), sizeof(drmlt_stats), sizeof(drmlt_splat));
  printf("%zu %zu %zu %zu\n", offsetof(drmlt_config, p_large), offsetof(drmlt_scene, camera),
         offsetof(drmlt_stats, kernel_ms), offsetof(drmlt_camera, width));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = [int(v) for v in out[:8]]
    mirror = [C.sizeof(t) for t in (abi.Config, abi.Shape, abi.Bsdf, abi.Emitter, abi.Camera, abi.Scene, abi.Stats,
                                    abi.Splat)]
    assert sizes == mirror
    offs = [int(v) for v in out[8:]]
    assert offs == [abi.Config.p_large.offset, abi.Scene.camera.offset, abi.Stats.kernel_ms.offset,
                    abi.Camera.width.offset]


def test_defaults_follow_the_reference_ctor(abi):
    c = abi.make_config()
    assert (c.max_depth, c.rr_depth, c.direct_samples, c.luminance_samples, c.work_units) == (-1, 5, 16, 100000, -1)
    assert c.p_large == pytest.approx(0.3) and c.sigma == pytest.approx(1 / 64) and c.scale_second == pytest.approx(0.1)
    assert c.average_luminance == -1.0 and not c.acceptance_map and not c.use_mixture and not c.timid_after_large
    assert abi.TYPE_NAMES["mirasym"] == abi.TYPE_ORBITAL  # drmlt.cpp:318


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_create_validates_like_the_reference_ctor(pkg, abi, native_lib):
    sd = pkg.scenes.cornell_c1(8)
    bad = [
        (dict(type="orbital", max_depth=8, scale_second=1.5), "scaleSecond"),
        (dict(type="orbital", max_depth=8, fix_emitter_path=1), "fixEmitterPath"),
        (dict(type="orbital", technique="mmlt", max_depth=-1), "MMLT"),
        (dict(type="orbital", max_depth=-1), "maxDepth"),
        (dict(type=7, max_depth=8), "implementation type"),
        (dict(type="orbital", technique="mmlt", max_depth=8, algo=1), "pssmlt"),
        (dict(type="orbital", technique="mmlt", max_depth=8, timid_after_large=1), "timidAfterLarge"),
        # a wave's sampler and density rows pass the 64 KB of a workgroup at maxDepth 26 (round 4: the flag words hold 2 * 24 + 1 slots)
        (dict(type="orbital", technique="bdpt", max_depth=25), "maxDepth above 24"),
        (dict(type="orbital", technique="mmlt", max_depth=25), "maxDepth above 24"),
    ]
    for kw, needle in bad:
        with pytest.raises(pkg.DrmltError) as e:
            pkg.Context(abi.make_config(**kw), sd)
        assert needle in str(e.value), str(e.value)
    gauss = pkg.scenes.cornell_c1(8, filt=abi.FILTER_GAUSSIAN)
    with pytest.raises(pkg.DrmltError) as e:
        pkg.Context(abi.make_config(type="orbital", max_depth=8, acceptance_map=1), gauss)
    assert "Box filter required" in str(e.value)


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a machine without a GPU")
def test_product_path_fails_loudly_without_a_gpu(pkg, abi, native_lib):
    sd = pkg.scenes.cornell_c1(8)
    with pytest.raises(pkg.DrmltError) as e:
        pkg.Context(abi.make_config(type="orbital", max_depth=8), sd)
    assert "no HIP device" in str(e.value) and "no CPU fallback" in str(e.value)


def test_missing_rccl_is_an_error_return_not_a_crash(pkg, abi):
    """the load-failure path called dlerror() twice (the second call returns NULL -> strlen(NULL)). A host
    without librccl must get DRMLT_E_DEVICE from every entry point that needs a communicator. Own process: the library
    caches the outcome of its one dlopen attempt."""
    code = r"""
import ctypes as C, sys
L = C.CDLL(sys.argv[1])
buf = C.create_string_buffer(128)
print(L.drmlt_comm_unique_id(buf), L.drmlt_comm_unique_id(buf))
"""
    env = dict(os.environ, DRMLT_RCCL_LIB="/nonexistent/librccl.so.1")
    out = subprocess.run([sys.executable, "-c", code, pkg.library_path()], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == [str(abi.E_DEVICE)] * 2
