#!/bin/bash
export PTK_DEV_TOOLS=1
# GPU box: L2 hit rate / bytes that left the L2 for the trace kernel of one config, library builds side by side.
#   bash tools/pmc_l2.sh <outdir-under-gpurun_out> C5 32 A -      ("-" = libptk.so)
OUT=$PWD/gpurun_out/$1; ROOT=$PWD; CFG=$2; SPP=$3; shift 3
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
a=0
for lib in "$@"; do
  a=$((a+1)); if [ "$lib" = "-" ]; then unset PTK_LIB_PATH; else export PTK_LIB_PATH=$ROOT/pbrpathtracer_amd/libptk_$lib.so; fi
  export PTK_PROBE_REPS=2
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/arm$a/p$i -- python3 $ROOT/tools/c5_probe.py $CFG $SPP > $OUT/arm${a}_p$i.log 2>&1
  done
  echo "lib=$lib" > $OUT/arm$a/opts.txt
done
cd $ROOT; python3 tools/pmc_quick_collect.py $OUT | tee $OUT/summary.txt
