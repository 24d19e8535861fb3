#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 900 python3 -m pytest tests/test_hip_parity.py tests/test_hip_spill.py tests/test_hip_scale.py tests/test_hip_fullsize_oracle.py -m gpu -x -q > gpurun_out/r4_call14_tests.txt 2>&1; echo "tests rc=$?" )
tail -6 gpurun_out/r4_call14_tests.txt
timeout -k 10 500 python3 tools/ab_bench.py --steps 20 --warmup 5 --rounds 3 ab/c2.so ab/c3.so ab/first.so > gpurun_out/r4_ab9.txt 2>&1
cat gpurun_out/r4_ab9.txt
python3 bench.py --hour > gpurun_out/r04_hour.json 2> gpurun_out/r04_hour.err; echo "hour rc=$?"
PYTSC_AMD_RATE_MIX="700:0.97,1500:0.03" python3 bench.py --hour > gpurun_out/r04_hour_mix.json 2> gpurun_out/r04_hour_mix.err; echo "hour mix rc=$?"
PYTSC_AMD_SPILL_LIMIT=-1 PYTSC_AMD_RATE_MIX="700:0.97,1500:0.03" python3 bench.py --hour > gpurun_out/r04_hour_mix_nopool.json 2> gpurun_out/r04_hour_mix_nopool.err; echo "hour mix nopool rc=$?"
