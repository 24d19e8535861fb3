cd /root/repo
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d $root/gpurun_out/r04f_hip_c16 -o s -- python3 $root/bench.py --no-cpu-baseline --no-extras --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 2 > $root/gpurun_out/r04f_hip_c16.log 2>&1
echo rc=$?
ls $root/gpurun_out/r04f_hip_c16
head -30 $root/gpurun_out/r04f_hip_c16/s_hip_api_stats.csv | cut -c1-150
