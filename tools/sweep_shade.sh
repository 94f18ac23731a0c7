#!/bin/bash
# GPU box: shade_threshold sweep on several configs.  bash tools/sweep_shade.sh <outfile>
OUT=${1:-gpurun_out/sweep_shade.log}; : > $OUT
for s in 32 40 48 56 72 40; do
  for c in C3 C4 C5; do
    echo "shade_threshold=$s $(PTK_OPTS="shade_threshold=$s" timeout -k 10 120 python3 tools/c5_probe.py $c 64 2>&1 | grep -E "spp" | tail -1 | sed -e 's/spp 64: wall [0-9.]* ms trace [0-9.]* acc [0-9.]* ->//')" >> $OUT
  done
done
cat $OUT
