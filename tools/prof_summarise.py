#!/usr/bin/env python3
"""Summaries of tools/prof_workload.sh: kernel_stats (short names), per-kernel PMC medians, HBM bytes / clock / matrix-pipe share."""
import collections, csv, glob, json, statistics as st, sys

out, tag = sys.argv[1], sys.argv[2]


def short(name: str) -> str:
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    for cut in ("(float", "(double", "(unsigned", "(HIP_vector", "(long", "(int", "(Roi", "(char", "(void"):
        name = name.split(cut)[0]
    return name.strip()[:120]


# kernel stats
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.reader(open(f[0])))
    keep = [rows[0]]
    for r in rows[1:]:
        if "at::native" in r[0] or "rocclr" in r[0] or "rocprim" in r[0]:
            continue
        r[0] = short(r[0])
        keep.append(r)
    csv.writer(open(f"gpurun_out/{tag}_kernel_stats.csv", "w")).writerows(keep)
    for r in keep[:14]:
        print(",".join(r[:5]))
# durations per kernel (ns) from the trace, for the clock estimate
dur = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
# ... and from the pass that counted GRBM_GUI_ACTIVE (a counted launch runs longer than an untraced one)
dur_pmc = collections.defaultdict(list)
for f in glob.glob(out + "/mfma/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur_pmc[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "at::native" in k or "rocclr" in k or "rocprim" in k:
            continue
        acc[short(k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = [("kernel", "counter", "median_per_launch", "launches")]
res = {}
for k, d in sorted(acc.items()):
    med = {c: st.median(v) for c, v in d.items()}
    for c, v in sorted(med.items()):
        rows.append((k, c, v, len(d[c])))
    e = {"launches_profiled": len(next(iter(d.values())))}
    if k in dur:
        e["avg_us_in_trace"] = round(st.mean(dur[k]) / 1e3, 2)
    if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
        e["hbm_bytes_per_launch"] = int(2 * med["FETCH_SIZE"] * 1024 + med["WRITE_SIZE"] * 1024)
        e["FETCH_SIZE_KiB"], e["WRITE_SIZE_KiB"] = med["FETCH_SIZE"], med["WRITE_SIZE"]
    if "SQ_INSTS_VALU" in med:
        e["valu_insts_per_launch"] = int(med["SQ_INSTS_VALU"])
        if med.get("SQ_WAVES"):
            e["valu_insts_per_wave"] = round(med["SQ_INSTS_VALU"] / med["SQ_WAVES"], 1)
    if "SQ_INSTS_MFMA" in med:
        e["mfma_insts_per_launch"] = int(med["SQ_INSTS_MFMA"])
    if "GRBM_GUI_ACTIVE" in med and med["GRBM_GUI_ACTIVE"] > 0:
        cyc = med["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        e["gpu_cycles_per_launch"] = int(cyc)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in med:
            e["matrix_pipe_busy_share"] = round(med["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
        if k in dur_pmc:
            e["clock_ghz_held"] = round(cyc / st.median(dur_pmc[k]), 3)  # GRBM_GUI_ACTIVE / 8 over the same launches' durations
    res[k] = e
csv.writer(open(f"gpurun_out/{tag}_pmc_raw.csv", "w")).writerows(rows)
json.dump(res, open(f"gpurun_out/{tag}_pmc_summary.json", "w"), indent=1)
big = sorted(res.items(), key=lambda kv: -kv[1].get("gpu_cycles_per_launch", 0))[:8]
print(json.dumps(dict(big), indent=1)[:3000])
