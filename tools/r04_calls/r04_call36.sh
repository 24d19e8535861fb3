cd /root/repo
for R in 1024 2048 4096; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --rows 16 --cols 16 --replicas $R --steps 10 --warmup 2 --flow-sets 2 2>/dev/null | python3 -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('R=$R', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],3), 'ms')"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "[A-Z0-9_]*UTCL[A-Z0-9_]*\|[A-Z0-9_]*TLB[A-Z0-9_]*" | sort -u | head -40
