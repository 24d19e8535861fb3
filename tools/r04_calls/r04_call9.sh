#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 1100 python3 -m pytest tests/test_vec_obs_spaces.py tests/test_hip_multirank.py tests/test_hip_scale.py tests/test_hip_spill.py -m gpu -x -q > gpurun_out/r4_call9_tests.txt 2>&1; echo "tests rc=$?" )
tail -30 gpurun_out/r4_call9_tests.txt
B="python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5"
for i in 1 2 3; do
  for lim in 0 -1; do
    echo "== spill limit $lim"; PYTSC_AMD_SPILL_LIMIT=$lim $B 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1e6,2), d['ms_per_step'])"
  done
done > gpurun_out/r4_ab6.txt 2>&1
cat gpurun_out/r4_ab6.txt
echo "== 16x16 list (k_agent row stores)"; python3 bench.py --no-cpu-baseline --no-extras --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 2 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1e6,2), d['ms_per_step'])"
