#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (run through gpurun from the repo root): tools/prof_bench.sh <tag>
set -e
tag=${1:-prof}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2000 --warmup 10 --no-cpu-baseline --no-api-level --no-stages > $out/bench.json 2> $out/bench.err || echo "rocprof run failed"
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
head -8 "$f"
cat $out/bench.json | head -c 600
