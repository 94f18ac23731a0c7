#!/bin/bash
# GPU box: sweep scheduling options of the BVH kernel on one config.  bash tools/sweep_opts.sh C4 64 <outfile>
CFG=${1:-C4}; SPP=${2:-64}; OUT=${3:-gpurun_out/sweep.log}
: > $OUT
run() { echo "== $1" >> $OUT; PTK_OPTS="$1" timeout -k 10 120 python3 tools/c5_probe.py $CFG $SPP 2>&1 | grep -E "spp" | tail -1 >> $OUT; }
run ""
for t in 2 3 5 6 8; do run "tri_threshold=$t"; done
for s in 16 24 32 56 80; do run "shade_threshold=$s"; done
for g in 4 8 32 64; do run "gen_threshold=$g"; done
run ""
