#!/bin/bash
# A/B through bench.py (consecutive batches overlapped, as the headline is measured): bash tools/ab_bench.sh C2 100 [C1 200 ...]
cd "$(dirname "$0")/.."
export PTK_DEV_TOOLS=1
while [ $# -ge 2 ]; do
  for arm in A B A B; do
    if [ $arm = B ]; then export PTK_LIB_PATH=$PWD/pbrpathtracer_amd/libptk_B.so; else unset PTK_LIB_PATH; fi
    echo "== $arm $1: $(timeout -k 10 300 python3 bench.py --config $1 --steps $2 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_isolated"])')"
  done
  shift 2
done
