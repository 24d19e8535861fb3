#!/bin/bash
mkdir -p gpurun_out
( PYTSC_AMD_LIB=$PWD/ab/vpf.so timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_hip_paths.py -m gpu -x -q > gpurun_out/r4_call16_tests.txt 2>&1; echo "tests rc=$?" )
tail -4 gpurun_out/r4_call16_tests.txt
timeout -k 10 500 python3 tools/ab_bench.py --steps 20 --warmup 5 --rounds 3 ab/first.so ab/vpf.so > gpurun_out/r4_ab10.txt 2>&1
cat gpurun_out/r4_ab10.txt
timeout -k 10 300 python3 tools/ab_bench.py --steps 20 --warmup 5 --rounds 2 --flow-sets=8 ab/first.so ab/vpf.so > gpurun_out/r4_ab10_sets8.txt 2>&1
cat gpurun_out/r4_ab10_sets8.txt
