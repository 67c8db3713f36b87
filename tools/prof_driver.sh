#!/bin/bash
# rocprofv3 kernel trace (+ optionally PMC passes) of the DRIVER's bench command, through gpurun from the repo root:
#   tools/prof_driver.sh <tag> trace          -> gpurun_out/<tag>_kernel_stats.csv + <tag>_bench.json (the line printed under the tracer)
#   tools/prof_driver.sh <tag> pmc            -> gpurun_out/<tag>_pmc_raw.csv + <tag>_pmc_summary.json (no CPU legs: --no-cpu-baseline)
set -e
tag=$1; mode=${2:-trace}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
if [ "$mode" = trace ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $out/trace.out 2> $out/trace.err || echo "trace run failed"
  cp $out/trace.out $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json
  cd $GRAFT_REPO_ROOT && python3 tools/prof_summarise.py "$out" "$tag" | head -40
else
  run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-api-level > $out/$name.out 2> $out/$name.err || echo "pass $name failed"; echo "pass $name done"; }
  run fetch FETCH_SIZE
  run write WRITE_SIZE
  run sq SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
  run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
  cd $GRAFT_REPO_ROOT && python3 tools/prof_summarise.py "$out" "$tag" | tail -5
  rm -rf $out/fetch $out/write $out/sq $out/mfma   # (the per-dispatch counter files of four passes exceed what gpurun copies back)
fi
