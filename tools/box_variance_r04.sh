#!/bin/bash
# On the GPU box: config 2 (16x16 @ 4096, list family) under rocprofv3 --stats with the box's clocks sampled beside it
# (the family's throughput follows the box -- quote k_agent / k_exec with the clocks next to them).
root=$(pwd)
tag=$(hostname | tr -c 'a-zA-Z0-9\n' '_')_$(date +%H%M%S)
out=$root/gpurun_out/r04_box_$tag
mkdir -p $out
rocm-smi --showproductname --showclocks --showperflevel > $out/idle.txt 2>&1
( while true; do rocm-smi --showclocks --csv 2>/dev/null | tail -n +2; sleep 0.5; done > $out/clocks.csv ) &
CL=$!
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $root/bench.py --no-cpu-baseline --no-extras --rows 16 --cols 16 --replicas 4096 --steps 10 --warmup 2 --flow-sets 2 > $out/bench.log 2>&1
rc=$?
kill $CL
cd $root
python3 - "$out" <<'PY'
import sys, csv, json, collections, os
out = sys.argv[1]
line = json.loads([l for l in open(out + "/bench.log") if l.startswith("{")][-1])
ks = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(out + "/stats/s_kernel_stats.csv"))}
sclk, mclk, fclk = [], [], []
mhz = lambda c: float(c.strip("() ")[:-3])
for row in csv.reader(open(out + "/clocks.csv")):
    if len(row) >= 6 and row[0].startswith("card"):
        try:
            f, m, s_ = mhz(row[1]), mhz(row[3]), mhz(row[5])
        except ValueError:
            continue
        if s_ > 500.0:  # (samples taken while kernels ran: the idle state reads 95-160 MHz)
            sclk.append(s_); mclk.append(m); fclk.append(f)
med = lambda v: sorted(v)[len(v) // 2] if v else None
ident = {}
for l in open(out + "/idle.txt"):
    for key in ("GUID", "Node ID", "Card Model"):
        if key + ":" in l: ident[key] = l.split(":")[-1].strip()
rec = {"box": os.path.basename(out), "value_M": round(line["value"] / 1e6, 2), "ms_per_step": round(line["ms_per_step"], 3),
       "k_exec_timed_us": round(line["roofline"]["avg_launch_us"], 1),
       "avg_us_whole_run": {k: round(v, 1) for k, v in ks.items() if k.startswith(("k_agent", "k_exec", "k_enum", "k_commit", "k_notify"))},
       "box_ident": ident, "clocks_under_load_MHz": {"samples": len(sclk), "sclk_median": med(sclk), "sclk_min": min(sclk) if sclk else None, "sclk_max": max(sclk) if sclk else None, "mclk_median": med(mclk), "fclk_median": med(fclk)}}
print(json.dumps(rec))
open(out + "/summary.json", "w").write(json.dumps(rec, indent=1))
PY
echo "box rc=$rc"
