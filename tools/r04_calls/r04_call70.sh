cd /root/repo
for K in 0 128 256 512 0 256; do
  PYTSC_AMD_POOL_PREFILL=$K PYTSC_AMD_SPILL_LIMIT=1024 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 30 2>gpurun_out/r04r_prefill_$K.err | python3 -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('prefill $K', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'ms err', d['config'].get('device_errors'), 'stalled', d['config'].get('stalled_replicas'))"
done
