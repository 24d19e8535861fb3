#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 600 python3 -m pytest tests/test_hip_paths.py tests/test_hip_parity.py -m gpu -x -q > gpurun_out/r4_call3_tests.txt 2>&1; echo "tests rc=$?" )
tail -5 gpurun_out/r4_call3_tests.txt
( time timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4_call3_bench.txt 2> gpurun_out/r4_call3_bench.err ) 2>&1 | tail -4
tail -3 gpurun_out/r4_call3_bench.err
python3 - <<'PY'
import json
for l in open('gpurun_out/r4_call3_bench.txt'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value']/1e6, d['ms_per_step'], d['config']['V_a'], [ (s.get('note','')[:30], round(s['value']/1e6,2)) for s in d.get('states',[])])
PY
