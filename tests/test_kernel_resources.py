"""Occupancy of the chain kernels, read from the compiler's resource remarks of the build (libdrmlt_amd.so.resources).
Two waves per SIMD are worth almost 2x on this hardware (a single wave64 issues a vector instruction every 4 cycles at
best, the SIMD one every 2): a kernel that silently grows past 256 registers loses half its throughput."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(path):
    out, name = {}, None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            out[name][m.group(1).split(" ")[0]] = int(m.group(2))
    return out


def test_chain_kernels_keep_their_occupancy(native_lib):
    res = os.path.join(ROOT, "drmlt-mitsuba_amd", "libdrmlt_amd.so.resources")
    assert os.path.exists(res), "the Makefile writes it next to the library"
    k = kernels(res)
    v4 = {n: r for n, r in k.items() if n.startswith("_Z11k_mutate_v4")}
    assert len(v4) >= 8
    for n, r in v4.items():
        assert r["Occupancy"] >= 2 and r["VGPRs"] + r["AGPRs"] <= 256, (n, r)
    headline = k["_Z11k_mutate_v4ILi0ELb1ELb0ELb0ELb0EEv7DParamsjj"]      # config 2's build
    assert headline["VGPRs"] <= 168 and headline["ScratchSize"] <= 16, headline
    bidir = {n: r for n, r in k.items() if n.startswith("_Z13k_mutate_mmlt") or n.startswith("_Z13k_mutate_bdptILi7ELi2E") or n.startswith("_Z13k_mutate_bdptILi15ELi2E")}
    assert len(bidir) >= 5, sorted(bidir)
    for n, r in bidir.items():
        assert r["Occupancy"] >= 2 and r["VGPRs"] + r["AGPRs"] <= 256, (n, r)
    # static LDS: only the builds that traverse a BVH carry a stack
    assert k["_Z13k_mutate_mmltILi7ELb1EEv7DParamsjj"]["LDS"] == 0 and k["_Z13k_mutate_bdptILi7ELi2ELb1EEv7DParamsjj"]["LDS"] == 0
    # k_mutate_v5 with its proposal rows in device memory (traversed scenes, >= 163 840 chains): registers for three waves per SIMD,
    # nothing spilled inside the traversal loop (what is spilled belongs to the bookkeeping branch: tools/isa_scratch.py)
    rows_mem = {n: r for n, r in k.items() if n.startswith("_Z11k_mutate_v5") and n.endswith("ELb0ELb0ELb1EEv7DParamsjj")}
    assert len(rows_mem) == 6, sorted(rows_mem)
    for n, r in rows_mem.items():
        assert r["Occupancy"] >= 3 and r["VGPRs"] + r["AGPRs"] <= 168, (n, r)
        # LDS beside the stack column: 4.5 KB (pool, pending-ray queue, chain list) + 2.9 KB (splat queue, coin rows) in the builds with
        # 16-bit stacks: twelve waves per CU
        stack16 = re.match(r"_Z11k_mutate_v5ILi\d+ELb1E", n) is not None
        assert r["LDS"] + 4608 + (2944 if stack16 else 0) <= 160 * 1024 // 12, (n, r)
    flat3 = {n: r for n, r in k.items() if n.startswith("_Z11k_mutate_v5") and n.endswith("ELb0ELb1ELb1EEv7DParamsjj")}   # flat scenes, tables in LDS, ROWS_MEM
    assert len(flat3) == 4, sorted(flat3)
    for n, r in flat3.items():
        assert r["Occupancy"] >= 3 and r["VGPRs"] + r["AGPRs"] <= 168 and r["ScratchSize"] <= 16, (n, r)
