for o in "generations=1" "generations=2" "generations=3" "generations=4"; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --share-of 8 --force-exchange --opts $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share8 exch', '$o', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
for o in "generations=1" "generations=3"; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --share-of 2 --force-exchange --opts $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share2 exch', '$o', d['ms_per_step'], d['roofline']['kernel_ms'])"
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --config C4 --share-of 8 --force-exchange --opts $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 share8 exch', '$o', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
