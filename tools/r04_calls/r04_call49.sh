cd /root/repo
root=$(pwd)
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_hip_scale.py tests/test_hip_facade.py tests/test_vec_obs_spaces.py -m gpu -x -q > gpurun_out/r4_call50_tests.txt 2>&1; tail -2 gpurun_out/r4_call47_tests.txt
timeout -k 10 300 python3 tools/fuzz_parity.py 24 123 > gpurun_out/r4_fuzz5.txt 2>&1; tail -1 gpurun_out/r4_fuzz2.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r04j_stats_c16 -o s -- python3 $root/bench.py --no-cpu-baseline --no-extras --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 2 > $root/gpurun_out/r04j_stats_c16.log 2>&1
echo "stats rc=$?"
grep -h "k_agent\|k_exec\|k_net_sums" $root/gpurun_out/r04j_stats_c16/s_kernel_stats.csv | cut -c1-120
grep "^{" $root/gpurun_out/r04j_stats_c16.log | cut -c1-260
