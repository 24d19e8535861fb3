#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 300 python3 -m pytest tests/test_hip_cross_kat.py -m gpu -q -k priority > gpurun_out/r4_call11_tests.txt 2>&1; echo "tests rc=$?" )
tail -12 gpurun_out/r4_call11_tests.txt
timeout -k 10 600 python3 tools/ab_bench.py --steps 20 --warmup 5 --rounds 3 ab/c2.so ab/cur3.so ab/noprio.so ab/nowant.so ab/noboth.so > gpurun_out/r4_ab8.txt 2>&1
cat gpurun_out/r4_ab8.txt
