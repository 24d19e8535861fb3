#!/bin/bash
# round 4, first GPU call: HEAD baselines + kr_step on the 4x4 at the 16x16's per-agent load
mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-extras"
{
echo "== driver command (HEAD)"; $B --steps 20 --warmup 5
for rate in 150 200 250 300; do echo "== 4x4 rate $rate"; $B --steps 40 --warmup 5 --rate $rate; done
echo "== 4x4 rate 700 list"; $B --steps 20 --warmup 5 --family list
echo "== 16x16 list"; $B --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 2
echo "== hour"; $B --hour
} > gpurun_out/r4_call1.txt 2>&1
echo done
