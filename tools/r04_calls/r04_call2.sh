#!/bin/bash
# round 4, perf batch 1: parity of the two source variants, then a same-box A/B of seven builds
mkdir -p gpurun_out
( PYTSC_AMD_LIB=$PWD/ab/both.so timeout -k 10 400 python3 -m pytest tests/test_hip_parity.py tests/test_hip_cross_kat.py -m gpu -x -q > gpurun_out/r4_call2_parity.txt 2>&1; echo "parity rc=$?" ) 
tail -3 gpurun_out/r4_call2_parity.txt
timeout -k 10 700 python3 tools/ab_bench.py --steps 20 --warmup 5 --rounds 3 ab/base.so ab/xlds.so ab/cdiv.so ab/both.so ab/ifcvt.so ab/wprio.so ab/nopost.so > gpurun_out/r4_ab1.txt 2>&1
cat gpurun_out/r4_ab1.txt
