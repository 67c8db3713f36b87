#!/bin/bash
# PMC passes of K1 on the GPU box (run through gpurun from the repo root).  Counters in their own runs, --kernel-trace only.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_k1}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 $GRAFT_REPO_ROOT/tools/k1_probe.py --iters 40 --warm 0.3 > $out/$name.log 2>&1 || echo "pass $name failed"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES
run b SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES
run c FETCH_SIZE GRBM_GUI_ACTIVE
run d WRITE_SIZE GRBM_COUNT
python3 - <<PY
import csv, glob, collections, statistics as st
for p in sorted(glob.glob("$out/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(p.split("/")[-3] if "/" in p else p, k, {c: round(st.median(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
