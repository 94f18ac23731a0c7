for o in "overlap=1" "overlap=0"; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --share-of 8 --opts $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share8', '$o', d['ms_per_step'], d['roofline']['kernel_ms'])"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --share-of 8 --force-exchange --opts $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share8 exch', '$o', d['ms_per_step'], d['roofline']['kernel_ms'])"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --share-of 4 --opts $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share4', '$o', d['ms_per_step'], d['roofline']['kernel_ms'])"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --opts $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('full', '$o', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'])"
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --config C4 --opts $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4', '$o', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'])"
done
