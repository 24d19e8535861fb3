cd /root/repo
export PYTSC_AMD_LIB=/root/repo/ab/flat.so
timeout -k 10 400 python3 -m pytest tests/test_hip_parity.py tests/test_vec_obs_spaces.py -m gpu -x -q > gpurun_out/r4_call64_tests.txt 2>&1; tail -1 gpurun_out/r4_call64_tests.txt
unset PYTSC_AMD_LIB
timeout -k 10 400 python3 tools/ab_bench.py --steps 100 --warmup 30 --rounds 3 ab/plain2.so ab/flat.so 2>&1 | tail -2
