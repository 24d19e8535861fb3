cd /root/repo
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu-baseline --no-extras --rows 16 --cols 16 --replicas 4096 --steps 6 --warmup 2 --flow-sets 2"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $root/gpurun_out/r04e_pmc_c16_sq -o p -- $B > $root/gpurun_out/r04e_pmc_c16_sq.log 2>&1
echo "sq rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $root/gpurun_out/r04e_pmc_c16_tcc -o p -- $B > $root/gpurun_out/r04e_pmc_c16_tcc.log 2>&1
echo "tcc rc=$?"
ls $root/gpurun_out/r04e_pmc_c16_sq $root/gpurun_out/r04e_pmc_c16_tcc
