for c in C2 C1; do PTK_STATS_SPP=64 timeout -k 10 100 python tools/stats_probe.py $c 256 | cut -c1-200; done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('full', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'])"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --share-of 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share8', d['ms_per_step'], d['roofline']['kernel_ms'])"
