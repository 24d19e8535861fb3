cd /root/repo
root=$(pwd)
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_hip_scale.py tests/test_hip_facade.py tests/test_vec_obs_spaces.py tests/test_shipped_scenarios.py -m gpu -x -q > gpurun_out/r4_call61_tests.txt 2>&1; tail -2 gpurun_out/r4_call61_tests.txt
timeout -k 10 300 python3 tools/fuzz_parity.py 30 999 > gpurun_out/r4_fuzz8.txt 2>&1; tail -1 gpurun_out/r4_fuzz8.txt
cd /tmp && export TMPDIR=/tmp
for v in headb sym headb sym; do
  export PYTSC_AMD_LIB=$root/ab/$v.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r04o_${v}_$RANDOM -o s -- python3 $root/bench.py --no-cpu-baseline --no-extras --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 2 > $root/gpurun_out/r04o_$v.log 2>&1
done
