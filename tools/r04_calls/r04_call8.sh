#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 1000 python3 -m pytest tests/test_hip_spill.py tests/test_vec_obs_spaces.py tests/test_hip_multirank.py tests/test_hip_paths.py -m gpu -x -q > gpurun_out/r4_call8_tests.txt 2>&1; echo "tests rc=$?" )
tail -30 gpurun_out/r4_call8_tests.txt
timeout -k 10 400 python3 tools/ab_bench.py --steps 20 --warmup 5 --rounds 3 ab/prev.so ab/cur2.so > gpurun_out/r4_ab5.txt 2>&1
cat gpurun_out/r4_ab5.txt
