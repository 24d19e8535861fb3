cd /root/repo
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_hip_scale.py tests/test_hip_facade.py tests/test_hip_cross_kat.py tests/test_vec_obs_spaces.py tests/test_hip_paths.py -m gpu -x -q > gpurun_out/r4_call29_tests.txt 2>&1; tail -2 gpurun_out/r4_call29_tests.txt
timeout -k 10 600 python3 tools/ab_bench.py --steps 10 --warmup 2 --rounds 2 --rows=16 --cols=16 --replicas=4096 --flow-sets=2 ab/lb8.so ab/obpack.so 2>&1 | tail -3
