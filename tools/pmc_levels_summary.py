"""Condense the rocprofv3 passes of tools/pmc_levels.sh into profiles/<tag>_summary.json (per level: kernel duration from
the trace, every counter as mean per launch, derived figures).

    python tools/pmc_levels_summary.py <tag>

Corrections as MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE, WRITE_SIZE in KB; FETCH_SIZE x2 on gfx950 for wide
streaming reads; WRITE_SIZE exact.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1]
root = os.path.join("gpurun_out", tag)
levels = {}
for line in open(os.path.join(root, "trace.log")):
    m = re.match(r"LEVEL (\d+) rows (\d+) nnz (\d+) tiles (\d+) bytes (\d+) : ([\d.]+) ms", line)
    if m:
        lv, rows, nnz, tiles, by, ms = (int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)),
                                        float(m.group(6)))
        levels[lv] = {"rows": rows, "nnz": nnz, "tiles": tiles, "algorithmic_bytes_per_launch": by, "hip_event_ms_per_launch": ms}


def level_of(grid_size, wg=256):
    """launch -> level: the tile-per-workgroup kernel has one 256-thread workgroup per tile (grid padded to 64)"""
    wgs = grid_size // wg
    for lv, d in levels.items():
        if d["tiles"] <= wgs < d["tiles"] + 64:
            return lv
    return None


kernel_rows = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmv_tiled_kernel" in r["Kernel_Name"] or "spmv_xs_kernel" in r["Kernel_Name"] or "spmv_sl_kernel" in r["Kernel_Name"] or "spmv_rs_kernel" in r["Kernel_Name"]:
            lv = level_of(int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))
            if lv is not None:
                kernel_rows[lv].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for lv, v in kernel_rows.items():
    v = v[3:] if len(v) > 3 else v                # drop the warm-up launches
    levels[lv]["trace_ms_per_launch"] = sum(v) / len(v)
    levels[lv]["trace_launches"] = len(v)
    levels[lv]["achieved_GBps"] = levels[lv]["algorithmic_bytes_per_launch"] / (sum(v) / len(v)) / 1e6
    levels[lv]["frac_of_8TBps"] = levels[lv]["achieved_GBps"] / 8000.0

for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "spmv_tiled_kernel" in r["Kernel_Name"] or "spmv_xs_kernel" in r["Kernel_Name"] or "spmv_sl_kernel" in r["Kernel_Name"] or "spmv_rs_kernel" in r["Kernel_Name"]:
            lv = level_of(int(r["Grid_Size"]))
            if lv is not None:
                agg[lv][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for lv, cs in agg.items():
        levels[lv].setdefault("counters", {})
        for c, v in cs.items():
            levels[lv]["counters"][c] = sum(v) / len(v)

for lv, d in levels.items():
    c = d.get("counters", {})
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["hbm_traffic_bytes_per_launch"] = 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
        d["traffic_over_algorithmic"] = d["hbm_traffic_bytes_per_launch"] / d["algorithmic_bytes_per_launch"]
    if "TCC_HIT_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "SQ_WAVE_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        d["mean_waves_per_busy_cycle"] = c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
        d["wave_time_waiting_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        d["wave_time_issue_stalled_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        d["wave_time_issuing_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
import hashlib
with open(os.path.join("hypre_amd", "csrc", "spmv_kernels.hip"), "rb") as fh:
    kernel_sha = hashlib.sha256(fh.read()).hexdigest()[:16]
out = {"tag": tag, "kernel_source_sha16": kernel_sha, "value_codes": os.environ.get("HYPRE_AMD_SPMV_VALUE_CODES", "1") != "0",
       "slice_form": os.environ.get("HYPRE_AMD_SPMV_VALUE_CODES", "1") != "0" and os.environ.get("HYPRE_AMD_SPMV_SLICE_FORM", "1") != "0",
        "command": "bash tools/pmc_levels.sh %s  (rocprofv3 --kernel-trace --stats, then one --pmc pass per counter set, "
                              "each over python3 tools/bench_levels_spmv.py 256 3 10)" % tag,
       "matrix": "levels 0-2 of the 256^3 7-pt hierarchy (PMIS, ext+i(4))",
       "note": "SQ_* cycle counters are in units of 4 cycles and sampled; FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE doubled "
               "(gfx950 reports half of wide streaming reads, MI355X_MICROARCH.md); the x-staged kernel reads 8 + 2 bytes "
               "per entry instead of CSR's 12, so its traffic may be below the algorithmic (CSR) byte count",
       "levels": {str(k): levels[k] for k in sorted(levels)}}
os.makedirs("profiles", exist_ok=True)
json.dump(out, open(os.path.join("profiles", "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1))
