#!/bin/bash
# On the GPU box, from the repo root: round 5's rocprofv3 evidence into gpurun_out/r05_* (copied into profiles/ by
# tools/summarise_profiles_r05.py).  Counters in their own passes (kernel-trace only), as MI355X_MICROARCH.md prescribes;
# the program itself follows `--` (python3 bench.py ...), never a shell or env wrapper.
# usage: tools/collect_profiles_r05.sh [part ...]   parts: headline c16 c16_fetch c16_write configs jinan hours curve (default: headline c16 configs hours)
root=$(pwd)
parts=${@:-headline c16 configs hours}
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu-baseline --no-extras"
stats() { # tag, bench args...
  tag=$1; shift
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r05_stats_$tag -o s -- $B "$@" > $root/gpurun_out/r05_stats_$tag.log 2>&1
  echo "stats $tag rc=$?"
}
pass() { # tag, "counters", bench args...
  tag=$1; ctr=$2; shift; shift
  timeout -k 10 ${PMC_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $root/gpurun_out/r05_pmc_$tag -o p -- $B "$@" > $root/gpurun_out/r05_pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
}
for part in $parts; do
case $part in
headline)
  stats resident --steps 20 --warmup 5
  pass fetch FETCH_SIZE --steps 20 --warmup 5
  pass write WRITE_SIZE --steps 20 --warmup 5
  pass sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" --steps 20 --warmup 5
  pass sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" --steps 20 --warmup 5
  pass tcc "TCC_HIT_sum TCC_MISS_sum" --steps 20 --warmup 5
  stats resident240
  ;;
c16)
  # BASELINE config 2 (16x16 @ 4096, list family) on DISTINCT demands (64 sets: every lane of a wavefront its own)
  stats c16 --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 64
  ;;
c16_fetch)  # (the 64 demands take 75 s to generate and a counter pass reads every launch out: one pass per GPU call)
  PMC_TIMEOUT=1000 pass c16_fetch FETCH_SIZE --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 64
  ;;
c16_write)
  PMC_TIMEOUT=900 pass c16_write WRITE_SIZE --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 64
  ;;
configs)
  # ... on 2 shared sets; config 4 (Jinan @ 8192)
  stats c16_sets2 --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 2
  stats jinan --scenario jinan --replicas 8192 --steps 40 --warmup 5
  pass jinan_fetch FETCH_SIZE --scenario jinan --replicas 8192 --steps 40 --warmup 5
  pass jinan_write WRITE_SIZE --scenario jinan --replicas 8192 --steps 40 --warmup 5
  ;;
jinan)
  stats jinan --scenario jinan --replicas 8192 --steps 40 --warmup 5
  pass jinan_fetch FETCH_SIZE --scenario jinan --replicas 8192 --steps 40 --warmup 5
  pass jinan_write WRITE_SIZE --scenario jinan --replicas 8192 --steps 40 --warmup 5
  ;;
hours)
  python3 $root/bench.py --hour > $root/gpurun_out/r05_hour.json 2> $root/gpurun_out/r05_hour.err; echo "hour rc=$?"
  ;;
curve)
  # the demand cliff: the sustained hour at rising demand, every replica at the same rate
  for rate in 700 900 1100 1300 1500; do
    python3 $root/bench.py --hour --rate $rate --flow-sets 256 > $root/gpurun_out/r05_hour_rate$rate.json 2> $root/gpurun_out/r05_hour_rate$rate.err; echo "hour $rate rc=$?"
  done
  ;;
esac
done
cd $root
