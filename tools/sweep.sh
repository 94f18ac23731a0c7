PTK_STATS_SPP=64 timeout -k 10 100 python tools/stats_probe.py C2 256 | cut -c1-100
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('full', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'])"
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --config C4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'])"
