#!/bin/bash
# one PMC pass of any python script: tools/pmc_cmd.sh <tag> "<counters>" <script.py> [args...]   (through gpurun)
set -e
tag=$1; counters=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
script=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $out -- python3 $script "$@" > $out/stdout.txt 2> $out/stderr.txt || echo "rocprof run failed"
cd $GRAFT_REPO_ROOT
python3 - "$out" <<'PY'
import csv, glob, sys, collections, statistics as st
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "at::native" in k or "rocclr" in k: continue
    k = k.split("(float")[0].split("((anon")[0].replace("void (anonymous namespace)::", "")[:60]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(st.median(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
