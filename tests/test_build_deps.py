"""Build hygiene: every header of csrc/ is a dependency of the objects that include it.

`csrc/Makefile` generates its dependencies (-MMD -MP). The test touches each header in turn (time stamps only -- the
content is not changed and the stamp is put back) and asks `make -n` whether a rebuild would happen."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "drmlt-mitsuba_amd", "csrc")
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "drmlt_abi.h")]


def _would_rebuild():
    out = subprocess.run(["make", "-C", CSRC, "-n"], capture_output=True, text=True, check=True).stdout
    return [l for l in out.splitlines() if "hipcc" in l and " -c " in l]


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)
    assert _would_rebuild() == [], "a fresh build must be up to date"


@pytest.mark.parametrize("header", HEADERS, ids=[os.path.basename(h) for h in HEADERS])
def test_touching_a_header_rebuilds_its_users(built, header):
    st = os.stat(header)
    try:
        os.utime(header, None)  # now: newer than every object
        units = _would_rebuild()
        assert units, "%s is included by no object's dependency file: an edit would rebuild nothing" % os.path.basename(header)
        if os.path.basename(header) in ("device_mh.h", "device_path.h", "device_math.h"):
            # the ONE copy of the decision logic / path machine: all three kernel units depend on it
            for u in ("kernels.hip", "kernels_mmlt.hip", "kernels_bdpt.hip"):
                assert any(u in l for l in units), (u, units)
    finally:
        os.utime(header, ns=(st.st_atime_ns, st.st_mtime_ns))
    assert _would_rebuild() == []
