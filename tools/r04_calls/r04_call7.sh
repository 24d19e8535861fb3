#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 900 python3 -m pytest tests/test_vec_domain_random.py tests/test_vec_obs_spaces.py tests/test_hip_multirank.py -m gpu -x -q > gpurun_out/r4_call7_tests.txt 2>&1; echo "tests rc=$?" )
tail -25 gpurun_out/r4_call7_tests.txt
timeout -k 10 500 python3 tools/ab_bench.py --steps 20 --warmup 5 --rounds 3 ab/prev.so ab/cur.so > gpurun_out/r4_ab4.txt 2>&1
cat gpurun_out/r4_ab4.txt
