#!/bin/bash
# GPU box: A/B of library builds within ONE gpurun call: bash tools/lib_ab.sh "A B" C2 256 C4 64 ...   ("-" = libptk.so)
cd "$(dirname "$0")/.."
export PTK_DEV_TOOLS=1
arms=$1; shift
while [ $# -ge 2 ]; do
  for rep in 1 2; do
    for a in $arms; do
      if [ "$a" = "-" ]; then unset PTK_LIB_PATH; else export PTK_LIB_PATH=$PWD/pbrpathtracer_amd/libptk_$a.so; fi
      echo "== [$a] $1 $2: $(timeout -k 10 300 python3 tools/c5_probe.py $1 $2 2>&1 | grep -E "spp" | tail -1)"
    done
  done
  shift 2
done
