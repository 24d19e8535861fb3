#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 500 python3 -m pytest tests/test_vec_domain_random.py tests/test_vec_obs_spaces.py -m gpu -x -q > gpurun_out/r4_call5_tests.txt 2>&1; echo "tests rc=$?" )
tail -25 gpurun_out/r4_call5_tests.txt
timeout -k 10 400 python3 tools/ab_bench.py --steps 20 --warmup 5 --rounds 2 --flow-sets=8 ab/head0.so ab/ret.so ab/retcdiv.so > gpurun_out/r4_ab2_sets8.txt 2>&1
cat gpurun_out/r4_ab2_sets8.txt
