#!/bin/bash
# On the GPU box, from the repo root: the round's rocprofv3 evidence into gpurun_out/r02_* (copied into profiles/ by
# tools/summarise_profiles_r02.py).  Counters in their own passes (kernel-trace only), as MI355X_MICROARCH.md prescribes.
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu-baseline"
for fam in resident list; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r02_stats_$fam -o s -- $B --family $fam > $root/gpurun_out/r02_stats_$fam.log 2>&1
  echo "stats $fam rc=$?"
done
pass() { # tag, counters...
  tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $root/gpurun_out/r02_pmc_$tag -o p -- $B --family resident --steps 20 --warmup 5 > $root/gpurun_out/r02_pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
# the same two counters in the state the default command line times (240 steps after 20 of warm-up)
pass240() {
  tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $root/gpurun_out/r02_pmc_$tag -o p -- $B --family resident > $root/gpurun_out/r02_pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
}
pass240 fetch240 FETCH_SIZE
pass240 write240 WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pass sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass tcc TCC_HIT_sum TCC_MISS_sum
cd $root
timeout 600 python3 bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench_line.err
echo "bench rc=$?"
tail -c 600 gpurun_out/r02_bench_line.json
