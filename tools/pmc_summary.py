"""Turn the rocprofv3 --pmc passes of tools/pmc_run.sh into the summary bench.py reads (profiles/rNN_<config>_pmc.json).

  python tools/pmc_summary.py <tag> <mutations per launch> <out.json> ["command line that was profiled"]

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads (doubled here); float atomics: WRITE_SIZE reads the bytes
exactly (one dword per lane), and every memory-side atomic request (TCC_EA0_ATOMIC) is one 32-byte write.
"""
import collections
import csv
import glob
import json
import os
import sys

tag, muts, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
cmd = sys.argv[4] if len(sys.argv) > 4 else ""
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
acc, n, kernel = collections.defaultdict(float), collections.defaultdict(int), None
for f in glob.glob(os.path.join(root, "pmc_%s_*" % tag, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mutate" in r["Kernel_Name"]:
            kernel = r["Kernel_Name"].split("(")[0]
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
if not acc:
    raise SystemExit("no k_mutate rows under gpurun_out/pmc_%s_*" % tag)
# per CALL (the profiled command is one drmlt_run call of `muts` mutations): a call of the bidirectional kernels is cut into a short first
# launch and the rest (regrouping), so the launches are SUMMED, not averaged -- round 4; with one launch per call the two are the same
launches_per_call = max(n.values())
c = {k: acc[k] for k in acc}
rd = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0
wr_atomic = c.get("TCC_EA0_ATOMIC_sum", 0.0) * 32.0
wr = max(c.get("WRITE_SIZE", 0.0) * 1024.0, wr_atomic)
s = {
    "command": cmd or "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py ... (tools/pmc_run.sh %s, one pass per counter group; FETCH_SIZE and WRITE_SIZE in separate passes)" % tag,
    "kernel": kernel, "mutations_per_launch": muts, "launches_per_call": launches_per_call, "counters_per_launch": c,
    "corrections": "FETCH_SIZE/WRITE_SIZE in KB; FETCH_SIZE doubled (gfx950 reports half of coalesced reads); write side = max(WRITE_SIZE x 1024, TCC_EA0_ATOMIC_sum x 32 B)",
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
    "hbm_bytes_per_mutation": (rd + wr) / muts,
    "atomic_requests_per_mutation": c.get("TCC_EA0_ATOMIC_sum", 0.0) / muts,
}
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    s["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
s["instructions_per_mutation"] = {k: c.get("SQ_INSTS_" + k.upper(), 0.0) / muts for k in ("valu", "salu", "smem", "lds")}
if "SQ_WAVE_CYCLES" in c:
    w = c["SQ_WAVE_CYCLES"]
    s["wave_time_split"] = {"issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / w, "parked_on_waitcnt": c.get("SQ_WAIT_ANY", 0.0) / w,
                            "issue_stall": c.get("SQ_WAIT_INST_ANY", 0.0) / w}
if "TCC_HIT_sum" in c:
    s["l2_hit_rate_excluding_atomics"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0) - c.get("TCC_EA0_ATOMIC_sum", 0.0), 1.0)
json.dump(s, open(out, "w"), indent=1)
print(json.dumps({k: s[k] for k in ("kernel", "hbm_bytes_per_mutation", "valu_lane_utilisation", "instructions_per_mutation") if k in s}))
