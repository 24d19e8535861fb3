#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 600 python3 -m pytest tests/test_vec_domain_random.py tests/test_vec_obs_spaces.py tests/test_hip_parity.py -m gpu -x -q > gpurun_out/r4_call6_tests.txt 2>&1; echo "tests rc=$?" )
tail -25 gpurun_out/r4_call6_tests.txt
timeout -k 10 500 python3 tools/ab_bench.py --steps 20 --warmup 5 --rounds 3 ab/prev.so ab/noise.so > gpurun_out/r4_ab3.txt 2>&1
cat gpurun_out/r4_ab3.txt
timeout -k 10 300 python3 tools/ab_bench.py --steps 20 --warmup 5 --rounds 2 --flow-sets=8 ab/prev.so ab/noise.so > gpurun_out/r4_ab3_sets8.txt 2>&1
cat gpurun_out/r4_ab3_sets8.txt
