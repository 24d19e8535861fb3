#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 1150 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_call12_tests.txt 2>&1; echo "tests rc=$?" )
tail -8 gpurun_out/r4_call12_tests.txt
