#!/bin/bash
# On the GPU box, from the repo root: round 4's rocprofv3 evidence into gpurun_out/r04_* (copied into profiles/ by
# tools/summarise_profiles_r04.py).  Counters in their own passes (kernel-trace only), as MI355X_MICROARCH.md prescribes;
# the program itself follows `--` (python3 bench.py ...), never a shell or env wrapper.
# usage: tools/collect_profiles_r04.sh [part ...]   parts: headline configs sweep (default: all three)
root=$(pwd)
parts=${@:-headline configs hours}
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu-baseline --no-extras"
stats() { # tag, bench args...
  tag=$1; shift
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r04_stats_$tag -o s -- $B "$@" > $root/gpurun_out/r04_stats_$tag.log 2>&1
  echo "stats $tag rc=$?"
}
pass() { # tag, "counters", bench args...
  tag=$1; ctr=$2; shift; shift
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $root/gpurun_out/r04_pmc_$tag -o p -- $B "$@" > $root/gpurun_out/r04_pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
}
for part in $parts; do
case $part in
headline)
  stats resident --family resident
  stats list --family list --steps 60 --warmup 5
  pass fetch FETCH_SIZE --family resident --steps 20 --warmup 5
  pass write WRITE_SIZE --family resident --steps 20 --warmup 5
  pass fetch240 FETCH_SIZE --family resident
  pass write240 WRITE_SIZE --family resident
  pass sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" --family resident --steps 20 --warmup 5
  pass sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" --family resident --steps 20 --warmup 5
  pass sq3 "SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" --family resident --steps 20 --warmup 5
  pass tcc "TCC_HIT_sum TCC_MISS_sum" --family resident --steps 20 --warmup 5
  ;;
configs)
  # BASELINE config 2 (16x16 @ 4096, list family) and config 4 (Jinan @ 8192) at their full sizes
  stats c16 --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 2
  pass c16_fetch FETCH_SIZE --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 2
  pass c16_write WRITE_SIZE --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 2
  stats jinan --scenario jinan --replicas 8192 --steps 40 --warmup 5
  pass jinan_fetch FETCH_SIZE --scenario jinan --replicas 8192 --steps 40 --warmup 5
  pass jinan_write WRITE_SIZE --scenario jinan --replicas 8192 --steps 40 --warmup 5
  ;;
hours)
  # sustained hours outside rocprof: the bench workload, and the same grid at 1500 veh/h per entry road -- replicas outgrow the
  # LDS-sized capacity and are promoted one by one to the spill pool
  python3 $root/bench.py --hour > $root/gpurun_out/r04_hour.json 2> $root/gpurun_out/r04_hour.err; echo "hour rc=$?"
  python3 $root/bench.py --hour --rate 1500 --flow-sets 256 > $root/gpurun_out/r04_hour_1500.json 2> $root/gpurun_out/r04_hour_1500.err; echo "hour1500 rc=$?"
  PYTSC_AMD_SPILL_LIMIT=-1 python3 $root/bench.py --hour --rate 1500 --flow-sets 256 > $root/gpurun_out/r04_hour_1500_nopool.json 2> $root/gpurun_out/r04_hour_1500_nopool.err; echo "hour1500 nopool rc=$?"
  ;;
sweep)
  # per-replica flow seeds ("seed = replica id for throughput runs"): what do the static per-set tables cost?
  for fs in 8 64 512 4096; do  # (4096 = one set per replica; at most 64 distinct demands are generated and tiled: bench.py)
    pass fs${fs}_fetch FETCH_SIZE --family resident --steps 20 --warmup 5 --flow-sets $fs
  done
  ;;
esac
done
cd $root
