cd /root/repo
timeout -k 10 400 python3 tools/ab_bench.py --steps 100 --warmup 30 --rounds 2 ab/meas.so ab/nt640.so ab/nt896.so 2>&1 | tail -4
export PYTSC_AMD_LIB=/root/repo/ab/nt640.so
timeout -k 10 300 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k resident > gpurun_out/r4_call33_nt640.txt 2>&1; tail -1 gpurun_out/r4_call33_nt640.txt
