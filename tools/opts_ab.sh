#!/bin/bash
# GPU box: A/B of ptk_set_option settings within ONE gpurun call, same library.
#   bash tools/opts_ab.sh "contract=0" "contract=1" "contract=2" -- C2 256 C4 64
cd "$(dirname "$0")/.."
arms=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do arms+=("$1"); shift; done
shift
while [ $# -ge 2 ]; do
  for rep in 1 2; do
    for a in "${arms[@]}"; do
      echo "== [$a] $1 $2: $(PTK_OPTS="$a" timeout -k 10 300 python3 tools/c5_probe.py $1 $2 2>&1 | grep -E "spp" | tail -1)"
    done
  done
  shift 2
done
