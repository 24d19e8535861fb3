#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 1100 python3 -m pytest tests/test_hip_scale.py tests/test_hip_spill.py tests/test_hip_paths.py tests/test_hip_cross_kat.py tests/test_vec_obs_spaces.py tests/test_hip_multirank.py -m gpu -x -q > gpurun_out/r4_call10_tests.txt 2>&1; echo "tests rc=$?" )
tail -8 gpurun_out/r4_call10_tests.txt
timeout -k 10 400 python3 tools/ab_bench.py --steps 20 --warmup 5 --rounds 3 ab/c2.so ab/cur3.so > gpurun_out/r4_ab7.txt 2>&1
cat gpurun_out/r4_ab7.txt
