cd /root/repo
timeout -k 10 900 python3 tools/ab_bench.py --steps 10 --warmup 2 --rounds 2 --rows=16 --cols=16 --replicas=4096 --flow-sets=2 ab/dsplain.so ab/dsnt.so 2>&1 | tail -2
