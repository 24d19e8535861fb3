cd /root/repo
timeout -k 10 400 python3 tools/ab_bench.py --steps 100 --warmup 30 --rounds 3 ab/plain2.so ab/rplain.so 2>&1 | tail -2
