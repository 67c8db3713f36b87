#!/bin/bash
# PMC passes of the bench command (through gpurun, from the repo root): counters in their own runs, --kernel-trace only.
#   tools/pmc_bench.sh <tag>   ->  gpurun_out/<tag>_pmc_raw.csv (per-kernel medians) + gpurun_out/<tag>_pmc_traffic.json
set -e
tag=${1:-pmc}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 4 --clock-warmup 0.2 --no-cpu-baseline --no-api-level --no-stages > $out/$name.json 2> $out/$name.err || echo "pass $name failed"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
cd $GRAFT_REPO_ROOT
python3 - "$out" "$tag" <<'PY'
import csv, glob, sys, json, collections, statistics as st
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "mc_entropy_kernel" if "mc_entropy_kernel" in k else "mc_mask_bits_kernel" if "mc_mask_bits" in k else "proj_sq_kernel" if "proj_sq_kernel" in k else None
        if name: acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = [("kernel", "counter", "median_per_launch", "launches")]
res = {}
for k, d in acc.items():
    med = {c: st.median(v) for c, v in d.items()}
    for c, v in sorted(med.items()): rows.append((k, c, v, len(d[c])))
    e = {"workload": "bench.py N=10000 images/launch", "images_per_launch": 10000, "profile": f"profiles/{tag}_pmc_raw.csv",
         "correction": "hbm = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE reports 1/2 of wide streaming reads; separate --pmc passes)"}
    if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
        e.update(FETCH_SIZE_KB_per_launch=med["FETCH_SIZE"], WRITE_SIZE_KB_per_launch=med["WRITE_SIZE"],
                 hbm_bytes_per_launch=int(2 * med["FETCH_SIZE"] * 1024 + med["WRITE_SIZE"] * 1024))
    if "SQ_INSTS_VALU" in med:
        e.update(valu_insts_per_launch=int(med["SQ_INSTS_VALU"]), waves_per_launch=int(med.get("SQ_WAVES", 0)),
                 valu_insts_per_wave=round(med["SQ_INSTS_VALU"] / max(1.0, med.get("SQ_WAVES", 1.0)), 1))
    res[k] = e
csv.writer(open(f"gpurun_out/{tag}_pmc_raw.csv", "w")).writerows(rows)
json.dump(res, open(f"gpurun_out/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])
PY
