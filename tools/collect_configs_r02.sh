#!/bin/bash
# On the GPU box, from the repo root: the other BASELINE configurations at the current build.
out=gpurun_out/r02_configs.txt
: > $out
for R in 1024 8192; do
  timeout 300 python3 bench.py --no-cpu-baseline --replicas $R --steps 40 --warmup 5 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('4x4 @ $R (%s): %.2f M agent-env-steps/s, %.3f ms per env step, V %.0f' % (d['config']['kernel_family'], d['value'] / 1e6, d['ms_per_step'], d['config']['mean_live_vehicles_per_replica']))
" >> $out
done
timeout 300 python3 bench.py --no-cpu-baseline --family list --steps 40 --warmup 5 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('4x4 @ 4096 list family: %.2f M agent-env-steps/s, %.3f ms per env step' % (d['value'] / 1e6, d['ms_per_step']))
" >> $out
timeout 300 python3 tools/jinan_scale.py 8192 auto 2>/dev/null | grep jinan >> $out
timeout 300 python3 tools/jinan_scale.py 8192 list 2>/dev/null | grep jinan >> $out
timeout 600 python3 bench.py --no-cpu-baseline --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('16x16 @ 4096 (%s): %.2f M agent-env-steps/s, %.3f ms per env step' % (d['config']['kernel_family'], d['value'] / 1e6, d['ms_per_step']))
" >> $out
cat $out
