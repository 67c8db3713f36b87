#!/bin/bash
# Kernel trace + PMC passes (own runs, --kernel-trace only) of any python command of this repo, through gpurun from the repo root:
#   tools/prof_workload.sh <tag> <script.py> [args...]
# -> gpurun_out/<tag>_kernel_stats.csv  (rocprofv3 --kernel-trace --stats, short kernel names)
#    gpurun_out/<tag>_pmc_raw.csv       (per kernel and counter: median per launch, launches)
#    gpurun_out/<tag>_pmc_summary.json  (per kernel: HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE in KiB (gfx950 correction,
#                                        MI355X_MICROARCH.md), VALU / MFMA instructions, matrix-pipe busy share, clock held)
set -e
tag=$1; shift
script=$GRAFT_REPO_ROOT/$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $script "$@" > $out/trace.out 2> $out/trace.err || echo "trace run failed"
echo "trace done"
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 $script $ARGS > $out/$name.out 2> $out/$name.err || echo "pass $name failed"; echo "pass $name done"; }
ARGS="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
cd $GRAFT_REPO_ROOT
python3 tools/prof_summarise.py "$out" "$tag"
