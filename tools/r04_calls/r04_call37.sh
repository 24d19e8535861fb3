cd /root/repo
bash tools/r04_final_lines.sh
python3 bench.py --hour > gpurun_out/r04_hour.json 2> gpurun_out/r04_hour.err; echo "hour rc=$?"
PYTSC_AMD_RATE_MIX=1 python3 bench.py --hour > gpurun_out/r04_hour_mix.json 2> gpurun_out/r04_hour_mix.err; echo "hour mix rc=$?"
python3 - <<'PY'
import json
for f in ('r04_hour','r04_hour_mix'):
    d=json.loads([l for l in open(f'gpurun_out/{f}.json') if l.startswith('{')][-1])
    print(f, round(d['value']/1e6,2), d.get('hour',{}).get('windows_M') or d.get('config',{}).get('windows'))
PY
