#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 600 python3 -m pytest tests/test_hip_spill.py tests/test_hip_paths.py -m gpu -x -q > gpurun_out/r4_call15_tests.txt 2>&1; echo "tests rc=$?" )
tail -4 gpurun_out/r4_call15_tests.txt
for i in 1 2; do
PYTSC_AMD_RATE_MIX="700:0.97,1500:0.03" python3 bench.py --hour > gpurun_out/r04_hour_mix.json 2> gpurun_out/r04_hour_mix.err; echo "hour mix rc=$?"
PYTSC_AMD_POOL_SERIAL=1 PYTSC_AMD_RATE_MIX="700:0.97,1500:0.03" python3 bench.py --hour > gpurun_out/r04_hour_mix_serial.json 2> gpurun_out/r04_hour_mix_serial.err; echo "hour mix serial rc=$?"
python3 - <<'PY'
import json
for f in ('r04_hour_mix','r04_hour_mix_serial'):
    d=json.loads([l for l in open(f'gpurun_out/{f}.json') if l.startswith('{')][-1])
    print(f, round(d['value']/1e6,2), d['spill_pool']['live'] if d['spill_pool'] else None, [round(w['value']/1e6,1) for w in d['windows']], d['device_errors'])
PY
done
