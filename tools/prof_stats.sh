#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_stats.sh <tag> [bench args]
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/bench.py --no-cpu-baseline --steps 120 --warmup 120 "$@" > $root/gpurun_out/prof_$tag.log 2>&1
grep -o '"value": [0-9.]*' $root/gpurun_out/prof_$tag.log
python3 - <<PY
import csv,glob
f=glob.glob('$root/gpurun_out/prof_$tag/**/*kernel_stats.csv',recursive=True)[0]
for row in list(csv.DictReader(open(f)))[:10]:
    print(row['Name'][:50].ljust(50), row['Calls'], round(float(row['AverageNs'])/1e3,1), round(float(row['MaxNs'])/1e3,1), row['Percentage'])
PY
