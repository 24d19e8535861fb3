cd /root/repo
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in nt plain nt plain; do
  export PYTSC_AMD_LIB=$root/ab/$v.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r04p_${v}_$RANDOM -o s -- python3 $root/bench.py --no-cpu-baseline --no-extras --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 2 > $root/gpurun_out/r04p_$v.log 2>&1
done
