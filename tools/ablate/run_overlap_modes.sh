#!/bin/bash
mkdir -p gpurun_out
for f in "" "--overlap" "--chunks 2" "--chunks 4"; do
  echo "flags: $f" >> gpurun_out/overlap_now.txt
  timeout -k 10 200 python bench.py --steps 600 --no-stages --no-cpu-baseline --no-api-level $f > gpurun_out/ov_tmp.json 2> gpurun_out/ov_tmp.err || { tail -3 gpurun_out/ov_tmp.err >> gpurun_out/overlap_now.txt; }
  python - >> gpurun_out/overlap_now.txt <<'P'
import json
try:
    d=json.loads([l for l in open('gpurun_out/ov_tmp.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])
except Exception as e: print('no line', e)
P
done
