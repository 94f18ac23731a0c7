#!/bin/bash
# GPU box: like ab.sh run, three arms (A = libptk.so, B, C).  bash tools/abc.sh C4 64 C5 64 ...
cd "$(dirname "$0")/.."
export PTK_DEV_TOOLS=1
while [ $# -ge 2 ]; do
  for arm in A B C A B C; do
    if [ $arm = A ]; then unset PTK_LIB_PATH; else export PTK_LIB_PATH=$PWD/pbrpathtracer_amd/libptk_$arm.so; fi
    echo "== $arm $1 $2: $(timeout -k 10 200 python3 tools/c5_probe.py $1 $2 2>&1 | grep -E "spp" | tail -1)"
  done
  shift 2
done
