#!/bin/bash
# GPU box: the round's soak set against the oracle with the kernels as built (logs under gpurun_out/soak_<tag>_*.log, one summary
# line each on stdout).  bash tools/soak_round.sh <tag> [part ...]     parts: scenes deep nan bvh splits api exchange big full cull
cd "$(dirname "$0")/.."
TAG=${1:-r04}; shift
PARTS=${@:-scenes deep nan bvh splits api exchange cull}
run() { name=$1; shift; t0=$(date +%s); "$@" > gpurun_out/soak_${TAG}_$name.log 2>&1; rc=$?; echo "== $name rc=$rc $(( $(date +%s) - t0 )) s: $(tail -1 gpurun_out/soak_${TAG}_$name.log | cut -c1-200)"; }
mkdir -p gpurun_out
for p in $PARTS; do
  case $p in
    scenes)   run scenes   timeout -k 10 900 python3 tools/soak_random_scenes.py 7000 ${N_SCENES:-200} ;;
    deep)     SOAK_DEEP=1 run deep timeout -k 10 900 python3 tools/soak_random_scenes.py 8000 ${N_DEEP:-100} ;;
    nan)      SOAK_NAN=1 run nan timeout -k 10 900 python3 tools/soak_random_scenes.py 9000 ${N_NAN:-100} ;;
    bvh)      run bvh      timeout -k 10 900 python3 tools/soak_bvh.py 3000 ${N_BVH:-150} ;;
    splits)   run splits   timeout -k 10 900 python3 tools/soak_splits.py 4000 ${N_SPLITS:-120} ;;
    api)      run api      timeout -k 10 900 python3 tools/soak_api.py 700 ${N_API:-40} ;;
    exchange) run exchange timeout -k 10 600 python3 tools/soak_exchange.py ${N_EXCHANGE:-100} ;;
    cull)     run cull     timeout -k 10 900 python3 tools/soak_lens_cull.py 1000 ${N_CULL:-300} ;;
    big)      run big      timeout -k 10 900 python3 tools/big_scene_check.py 2828 1414 16 499 ;;
    full)     for c in "C3 32" "C4 16" "C5 8"; do set -- $c; run full_$1 timeout -k 10 900 python3 tools/fullframe_check.py $1 $2; done ;;
  esac
done
