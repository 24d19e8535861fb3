cd /root/repo
for v in pool128 pool192 pool320; do
  export PYTSC_AMD_LIB=/root/repo/ab/$v.so
  timeout -k 10 300 python3 -m pytest tests/test_hip_spill.py -m gpu -x -q > gpurun_out/r4_call69_$v.txt 2>&1; echo "$v: $(tail -1 gpurun_out/r4_call69_$v.txt)"
done
for v in pool256 pool128 pool192 pool320 pool256 pool128; do
  export PYTSC_AMD_LIB=/root/repo/ab/$v.so
  PYTSC_AMD_RATE_MIX="700:0.97,1500:0.03" timeout -k 10 200 python3 bench.py --hour > gpurun_out/r04q_hour_mix_$v.json 2> gpurun_out/r04q_hour_mix_$v.err
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04q_hour_mix_$v.json') if l.startswith('{')][-1]); print('$v', round(d['value']/1e6,2), d.get('device_errors'), d.get('stalled_replicas'))"
done
