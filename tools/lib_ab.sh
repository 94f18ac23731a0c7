#!/bin/bash
# GPU box: A/B of library builds within ONE gpurun call (boxes differ by several per cent, arms are interleaved and repeated):
#   bash tools/lib_ab.sh "A B" C2 256 C4 64 ...        ("-" = libptk.so; X = pbrpathtracer_amd/libptk_X.so)
# Build an arm with  make -C pbrpathtracer_amd/csrc OUT=../libptk_X.so BUILD=build_X EXTRA=-DPTK_SOMETHING=1
# PARITY=1 first runs the random-scene parity tests against the oracle with every arm (an arm that fails is dropped).
cd "$(dirname "$0")/.."
export PTK_DEV_TOOLS=1
arms=$1; shift
setlib() { if [ "$1" = "-" ]; then unset PTK_LIB_PATH; else export PTK_LIB_PATH=$PWD/pbrpathtracer_amd/libptk_$1.so; fi; }
if [ "${PARITY:-0}" = "1" ]; then
  good=""
  for a in $arms; do
    setlib $a
    if timeout -k 10 600 python3 -m pytest tests/test_gpu_random_scenes.py tests/test_gpu_parity.py -x -q -m gpu -k "${PARITY_K:-random_scene or grazing or render_matches or closest_hit}" > gpurun_out/parity_$a.log 2>&1; then
      echo "== parity [$a] OK: $(tail -1 gpurun_out/parity_$a.log)"; good="$good $a"
    else
      echo "== parity [$a] FAILED: $(tail -3 gpurun_out/parity_$a.log)"
    fi
  done
  arms=$good
fi
while [ $# -ge 2 ]; do
  for rep in 1 2; do
    for a in $arms; do
      setlib $a
      out=$(timeout -k 10 300 python3 tools/c5_probe.py $1 $2 2>&1)
      echo "== [$a] $1 $2: $(echo "$out" | grep -E "spp" | sort -t'>' -k2 -n | tail -1)"
      if [ $rep = 1 ]; then echo "   [$a] $(echo "$out" | grep -E "^\{" | tail -1)"; fi
    done
  done
  shift 2
done
