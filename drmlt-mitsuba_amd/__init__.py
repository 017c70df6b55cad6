"""drmlt-mitsuba_amd: MI355X-native DRMLT hot path (HIP kernels behind a C-ABI).

The directory name is not a valid Python identifier; load it with
`__graft_entry__.load_package()` (registers it as module `drmlt_mitsuba_amd`).
"""
from . import abi, heatmap, scenes  # noqa: F401
from .binding import Context, DrmltError, Node, build_native, comm_unique_id, library_path  # noqa: F401
