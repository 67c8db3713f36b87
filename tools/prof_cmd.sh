#!/bin/bash
# rocprofv3 kernel-trace summary of any python script (run through gpurun from the repo root):
#   tools/prof_cmd.sh <tag> <script.py> [args...]     -> gpurun_out/<tag>_kernel_stats.csv (short kernel names)
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
script=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $script "$@" > $out/stdout.txt 2> $out/stderr.txt || echo "rocprof run failed"
cd $GRAFT_REPO_ROOT
python3 - "$out" "$tag" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(f)))
keep = [rows[0]]
for r in rows[1:]:
    name = r[0]
    if "at::native" in name or "rocclr" in name:
        continue
    r[0] = name.split("(float")[0].split("((anon")[0].split("(double")[0].split("(unsigned")[0].strip()[:110]
    keep.append(r)
csv.writer(open(f"gpurun_out/{tag}_kernel_stats.csv", "w")).writerows(keep)
for r in keep[:12]:
    print(",".join(r[:5]))
PY
tail -3 $out/stdout.txt
