#!/bin/bash
# On the GPU box, from the repo root: the round's rocprofv3 evidence into gpurun_out/final_*.
# Counters in their own passes (kernel-trace only), as MI355X_MICROARCH.md prescribes.
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/final_stats -o s -- $B --steps 240 --warmup 120 > $root/gpurun_out/final_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/final_fetch -o p -- $B --steps 20 --warmup 150 > $root/gpurun_out/final_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/final_write -o p -- $B --steps 20 --warmup 150 > $root/gpurun_out/final_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $root/gpurun_out/final_sq -o p -- $B --steps 20 --warmup 150 > $root/gpurun_out/final_sq.log 2>&1
cd $root
timeout 400 python3 bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err
tail -1 gpurun_out/bench_r01.json | cut -c1-300
