"""Per-launch durations of bench.py's kernel figures out of a rocprofv3 --kernel-trace of bench.py.

    python tools/trace_summary.py <kernel_trace.csv> <out.json>

`--stats` averages one kernel symbol over every level of the hierarchy (the SpMV kernels serve residuals, prolongations
and restrictions of all levels).  bench.py's kernel figures are the only places where the same kernel is dispatched many
times back to back on one matrix (5 warm-up launches, one for the byte counters, 50 timed): every run of 40 or more
consecutive dispatches of one kernel with one grid size is one of those loops — the fine-level operator as launched, the
same with the value codes off, the level-1 operator.  The mean of a run's last 50 launches is the number to set against
`ms_per_launch` of the JSON line; the run of the kernel `roofline.kernel` names is the roofline's.
"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
runs = []
i = 0
while i < len(rows):
    j = i
    key = (rows[i]["Kernel_Name"], rows[i].get("Grid_Size", rows[i].get("Grid_Size_X")))
    while j + 1 < len(rows) and (rows[j + 1]["Kernel_Name"], rows[j + 1].get("Grid_Size", rows[j + 1].get("Grid_Size_X"))) == key:
        j += 1
    if j - i + 1 >= 40:
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows[i:j + 1]]
        timed = dur[-50:]
        runs.append({"kernel": rows[i]["Kernel_Name"], "grid_size": key[1], "consecutive_launches": j - i + 1,
                     "mean_ms_last_50": sum(timed) / len(timed), "min_ms": min(dur), "max_ms": max(dur)})
    i = j + 1
out = {"runs_of_40_or_more_consecutive_launches": runs}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
