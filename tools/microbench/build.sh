#!/bin/bash
# Builds the two microbenchmarks for gfx950 with the render kernel's float flags.
cd "$(dirname "$0")"
for p in valu_calib gather_bench exact_math exec_half bvh_width; do
  /opt/rocm/bin/hipcc -O3 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -o $p $p.hip || exit 1
done
