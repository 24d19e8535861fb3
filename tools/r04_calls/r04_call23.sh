cd /root/repo
export PYTSC_AMD_LIB=/root/repo/ab/grab.so
timeout -k 10 600 python3 -m pytest tests/test_hip_paths.py tests/test_hip_parity.py tests/test_hip_spill.py tests/test_hip_scale.py -m gpu -x -q > gpurun_out/ab_grab_tests.txt 2>&1; tail -2 gpurun_out/ab_grab_tests.txt
export PYTSC_AMD_LIB=/root/repo/ab/freg1.so
timeout -k 10 400 python3 -m pytest tests/test_hip_paths.py tests/test_hip_parity.py -m gpu -x -q > gpurun_out/ab_freg1_tests.txt 2>&1; tail -2 gpurun_out/ab_freg1_tests.txt
unset PYTSC_AMD_LIB
timeout -k 10 400 python3 tools/ab_bench.py --steps 100 --warmup 30 --rounds 2 ab/base.so ab/loadall.so ab/grab.so ab/grab1.so ab/grab3.so 2>&1 | tail -5
