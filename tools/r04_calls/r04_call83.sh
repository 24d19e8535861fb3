cd /root/repo
root=$(pwd)
timeout -k 10 700 python3 -m pytest tests/test_hip_parity.py tests/test_hip_cross_kat.py tests/test_hip_scale.py tests/test_hip_facade.py tests/test_shipped_scenarios.py tests/test_hip_paths.py -m gpu -x -q > gpurun_out/r4_call83_tests.txt 2>&1; tail -3 gpurun_out/r4_call73_tests.txt
timeout -k 10 300 python3 tools/fuzz_parity.py 40 16180 > gpurun_out/r4_fuzz15.txt 2>&1; tail -1 gpurun_out/r4_fuzz9.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r04y_stats_c16 -o s -- python3 $root/bench.py --no-cpu-baseline --no-extras --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 2 > $root/gpurun_out/r04y_stats_c16.log 2>&1
echo "stats rc=$?"
grep "^{" $root/gpurun_out/r04y_stats_c16.log | cut -c85-135
