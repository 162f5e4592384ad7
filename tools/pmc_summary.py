"""Condense rocprofv3 --pmc counter CSVs of tools/bench_spmv.py into profiles/<tag>_spmv_pmc_summary.json.

    python tools/pmc_summary.py <tag> <dir-with-pmc-subdirs-prefix>     e.g.  r01b gpurun_out/pmc2_

Corrections follow MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KB;
on gfx950 FETCH_SIZE reports half of the bytes of wide streaming reads (x2), WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, prefix = sys.argv[1], sys.argv[2]
out = {}
kernel = None
for d in sorted(glob.glob(prefix + "*/**/*_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "spmv_tiled_kernel<0, false, false" in r["Kernel_Name"]:
            kernel = r["Kernel_Name"]
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        out[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    name = [c for c in d.split(os.sep) if c.startswith(os.path.basename(prefix))][0]
    shutil.copy(d, os.path.join("profiles", "%s_%s.csv" % (tag, name)))
fetch = out["FETCH_SIZE"]["mean_per_launch"] * 1024
write = out["WRITE_SIZE"]["mean_per_launch"] * 1024
summary = {
    "tag": tag,
    "command": "rocprofv3 --pmc <counter set> --output-format csv -- python3 tools/bench_spmv.py 256 10 "
               "(one pass per counter set, no tracing options)",
    "kernel": kernel,
    "matrix": "256^3 7-pt Laplacian (16,777,216 rows, 117,047,296 nnz)",
    "algorithmic_bytes_per_launch": 117047296 * 12 + (16777216 + 1) * 4 + 16777216 * 16,
    "counters": out,
    "FETCH_SIZE_bytes_raw": fetch,
    "FETCH_SIZE_bytes_corrected_x2": 2 * fetch,
    "WRITE_SIZE_bytes": write,
    "hbm_traffic_bytes_per_launch": 2 * fetch + write,
    "l2_hit_rate": out["TCC_HIT_sum"]["mean_per_launch"] /
                   (out["TCC_HIT_sum"]["mean_per_launch"] + out["TCC_MISS_sum"]["mean_per_launch"]),
    "cross_check_TCC_MISS_x_128B": out["TCC_MISS_sum"]["mean_per_launch"] * 128,
    "note": "FETCH_SIZE counts L2 misses sent to the fabric; misses served by the 256 MB Infinity Cache are "
            "included, so this is an upper bound of the HBM read traffic.",
}
json.dump(summary, open(os.path.join("profiles", "%s_spmv_pmc_summary.json" % tag), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "counters"}, indent=1))
for k, v in out.items():
    print(k, v)
