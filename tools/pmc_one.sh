#!/bin/bash
# usage (GPU box, repo root): tools/pmc_one.sh "<counters>" <tag>  -- one short --pmc pass, hard timeout
root=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $root/gpurun_out/pmc_$2 -o p -- python3 $root/bench.py --no-cpu-baseline --steps 10 --warmup 100 > $root/gpurun_out/pmc_$2.log 2>&1
echo "$2 rc=$?"
