cd /root/repo
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in packed rc; do
  export PYTSC_AMD_LIB=$root/ab/$v.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r04l_$v -o s -- python3 $root/bench.py --no-cpu-baseline --no-extras --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 2 > $root/gpurun_out/r04l_$v.log 2>&1
  echo "$v: $(grep -h 'k_agent' $root/gpurun_out/r04l_$v/s_kernel_stats.csv | cut -d, -f4)  exec $(grep -h 'k_exec' $root/gpurun_out/r04l_$v/s_kernel_stats.csv | cut -d, -f4)"
done
