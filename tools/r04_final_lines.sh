#!/bin/bash
# the two bench lines quoted in DESIGN.md 5: the driver's command and the default command, as the driver runs them
mkdir -p gpurun_out
( time python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_line_driver_cmd.json 2> gpurun_out/r04_bench_line_driver_cmd.err ) 2>&1 | grep real
( time python3 bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err ) 2>&1 | grep real
python3 - <<'PY'
import json
for f in ('r04_bench_line_driver_cmd','r04_bench_line'):
    d=json.loads([l for l in open(f'gpurun_out/{f}.json') if l.startswith('{')][-1])
    print(f, round(d['value']/1e6,2), round(d['ms_per_step'],4), 'V_a', round(d['config']['V_a'],1), 'frac', round(d['roofline']['frac'],4), 'cpu', round(d['cpu_baseline']['value']/1e3,1), d['cpu_baseline']['cores'])
    for s in d.get('states',[]): print('   ', s.get('note','')[:60], round(s['value']/1e6,2), s.get('last_window_value'))
PY
