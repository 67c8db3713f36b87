#!/bin/bash
# PMC passes of K2' alone (through gpurun, from the repo root): tools/pmc_k2.sh <tag>
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_k2}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 $GRAFT_REPO_ROOT/tools/ablate/run_proj_acc.py 10000 > $out/$name.log 2>&1 || echo "pass $name failed"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES
run b SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT
run c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
run d TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run e GRBM_GUI_ACTIVE GRBM_COUNT
cd $GRAFT_REPO_ROOT
python3 - "$out" <<'PY'
import csv, glob, sys, collections, statistics as st
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "proj_sq" in r["Kernel_Name"] and "combine" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print(f"{c:36s} median {st.median(v):16.1f}  launches {len(v)}")
PY
