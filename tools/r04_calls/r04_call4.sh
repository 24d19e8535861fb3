#!/bin/bash
mkdir -p gpurun_out
( PYTSC_AMD_LIB=$PWD/ab/retcdiv.so timeout -k 10 500 python3 -m pytest tests/test_hip_parity.py tests/test_hip_scale.py -m gpu -x -q > gpurun_out/r4_call4_tests.txt 2>&1; echo "tests rc=$?" )
tail -3 gpurun_out/r4_call4_tests.txt
timeout -k 10 600 python3 tools/ab_bench.py --steps 20 --warmup 5 --rounds 3 ab/head0.so ab/ret.so ab/retcdiv.so > gpurun_out/r4_ab2_sets4096.txt 2>&1
cat gpurun_out/r4_ab2_sets4096.txt
timeout -k 10 400 python3 tools/ab_bench.py --steps 20 --warmup 5 --rounds 2 ab/head0.so ab/ret.so ab/retcdiv.so --flow-sets 8 > gpurun_out/r4_ab2_sets8.txt 2>&1
cat gpurun_out/r4_ab2_sets8.txt
