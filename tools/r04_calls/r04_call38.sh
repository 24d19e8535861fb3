cd /root/repo
export PYTSC_AMD_LIB=/root/repo/ab/notify2.so
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_hip_paths.py tests/test_hip_cross_kat.py tests/test_hip_spill.py -m gpu -x -q > gpurun_out/r4_call38_tests.txt 2>&1; tail -2 gpurun_out/r4_call38_tests.txt
unset PYTSC_AMD_LIB
timeout -k 10 400 python3 tools/ab_bench.py --steps 100 --warmup 30 --rounds 3 ab/meas.so ab/notify2.so 2>&1 | tail -3
