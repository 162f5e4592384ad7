"""Condense the rocprofv3 passes of tools/pmc_multivector.sh into profiles/<tag>_summary.json: per kernel of the
single-vector and multivector products on the 256^3 7-point operator its mean duration (kernel trace) and every counter
as mean per launch, with the derived traffic figures.

    python tools/pmc_multivector_summary.py <tag>

Corrections as MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE, WRITE_SIZE in KB; FETCH_SIZE x2 on gfx950 for wide
streaming reads; WRITE_SIZE exact.  TCP_TCC_READ_REQ counts 64-byte (sometimes 128-byte) requests from the CUs' vector
caches to the L2: multiplied by 64 it is a LOWER bound of the bytes that crossed from the L2 to the CUs.
"""
import collections, csv, glob, json, os, re, sys

tag = sys.argv[1]
root = os.path.join("gpurun_out", tag)
ROWS = 256 ** 3


def short(name):
    m = re.search(r"(spmv_\w+_kernel<[^>]*>)", name)
    return m.group(1) if m else None


def wanted(name):
    s = short(name)
    return s if s and (("mv_kernel" in s) or s.startswith("spmv_sl_kernel<0") or s.startswith("spmv_xs_kernel<0, 0")) else None


kern = collections.defaultdict(dict)
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        s = wanted(r["Kernel_Name"])
        if s and int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) >= ROWS // 2:
            dur[s].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for s, v in dur.items():
    v = v[3:] if len(v) > 6 else v
    kern[s]["trace_ms_per_launch"] = round(sum(v) / len(v), 4)
    kern[s]["trace_launches"] = len(v)
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        s = wanted(r["Kernel_Name"])
        if s and int(r["Grid_Size"]) >= ROWS // 2:
            agg[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for s, cs in agg.items():
        kern[s].setdefault("counters", {})
        for c, v in cs.items():
            kern[s]["counters"][c] = sum(v) / len(v)
for s, d in kern.items():
    c = d.get("counters", {})
    ms = d.get("trace_ms_per_launch")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c and ms:
        hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        d["hbm_traffic_GB"] = round(hbm / 1e9, 3)
        d["hbm_TBps"] = round(hbm / ms / 1e9, 2)
    if "TCP_TCC_READ_REQ_sum" in c and ms:
        d["l2_to_cu_GB_at_64B_per_request"] = round(c["TCP_TCC_READ_REQ_sum"] * 64 / 1e9, 3)
        d["l2_to_cu_TBps_at_64B_per_request"] = round(c["TCP_TCC_READ_REQ_sum"] * 64 / ms / 1e9, 2)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0), 3)
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
        d["lds_bank_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), 3)
    if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
        d["waves_waiting_share"] = round(c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0), 3)
out = {"workload": "256^3 7-point operator, Y = A X (tools/bench_multivector.py): single-vector kernels and the fused multivector kernels",
       "kernels": {k: kern[k] for k in sorted(kern)}}
os.makedirs("profiles", exist_ok=True)
with open(os.path.join("profiles", tag + "_summary.json"), "w") as f:
    json.dump(out, f, indent=1)
for k in sorted(kern):
    d = kern[k]
    print(k, {x: d[x] for x in d if x != "counters"})
