#!/bin/bash
# A/B of a compile-time kernel variant within ONE gpurun call (boxes differ by several per cent, so arms are only comparable
# on the same box).  Here (build container):  bash tools/ab.sh build -DPTK_X=0      -> pbrpathtracer_amd/libptk_B.so
# On the GPU box:                              bash tools/ab.sh run C4 64 [C5 64 ...]  (alternates A, B, A, B)
set -e
cd "$(dirname "$0")/.."
export PTK_DEV_TOOLS=1
if [ "$1" = build ]; then shift; make -C pbrpathtracer_amd/csrc -j8 OUT=../libptk_B.so BUILD=build_B EXTRA="$*" 2>&1 | grep -E "error|warning" || true; ls -la pbrpathtracer_amd/libptk_B.so; exit 0; fi
shift
while [ $# -ge 2 ]; do
  for arm in A B A B; do
    if [ $arm = B ]; then export PTK_LIB_PATH=$PWD/pbrpathtracer_amd/libptk_B.so; else unset PTK_LIB_PATH; fi
    echo "== $arm $1 $2: $(timeout -k 10 200 python3 tools/c5_probe.py $1 $2 2>&1 | grep -E "spp" | tail -1)"
  done
  shift 2
done
