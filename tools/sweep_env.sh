#!/bin/bash
# GPU box: sweep builder knobs (environment) of the BVH kernel on one config.  bash tools/sweep_env.sh C4 64 <outfile>
CFG=${1:-C4}; SPP=${2:-64}; OUT=${3:-gpurun_out/sweep_env.log}
: > $OUT
for tc in 0.25 0.5 1.0 2.0 4.0; do
  echo "== bvh_trav_cost=$tc" >> $OUT
  PTK_OPTS=bvh_trav_cost=$tc timeout -k 10 120 python3 tools/c5_probe.py $CFG $SPP 2>&1 | grep -E "spp|max nodes|bvh" | tail -3 >> $OUT
done
