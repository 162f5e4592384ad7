"""Per-launch duration of the roofline kernel out of a rocprofv3 --kernel-trace of bench.py.

    python tools/trace_summary.py <kernel_trace.csv> <out.json>

`--stats` averages one kernel symbol over every level of the hierarchy (the tiled SpMV serves
residuals, prolongations and restrictions of all levels).  bench.py's roofline leg is the only place
where the same kernel is dispatched many times back to back on the fine-level matrix, so the longest
run of consecutive dispatches of one kernel with one grid size is that loop; its mean duration is the
number to set against `roofline.ms_per_launch` of the JSON line.
"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
best = (0, 0, 0)
i = 0
while i < len(rows):
    j = i
    key = (rows[i]["Kernel_Name"], rows[i].get("Grid_Size", rows[i].get("Grid_Size_X")))
    while j + 1 < len(rows) and (rows[j + 1]["Kernel_Name"], rows[j + 1].get("Grid_Size", rows[j + 1].get("Grid_Size_X"))) == key:
        j += 1
    if j - i + 1 > best[0]:
        best = (j - i + 1, i, j)
    i = j + 1
n, a, b = best
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows[a:b + 1]]
timed = dur[5:] if len(dur) > 10 else dur           # bench.py discards 5 warm-up launches
out = {"kernel": rows[a]["Kernel_Name"], "grid_size": rows[a].get("Grid_Size", rows[a].get("Grid_Size_X")),
       "consecutive_launches": n, "mean_ms_all": sum(dur) / len(dur), "mean_ms_after_5_warmup": sum(timed) / len(timed),
       "min_ms": min(dur), "max_ms": max(dur)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
