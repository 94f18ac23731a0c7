#!/bin/bash
# GPU box: SQ / L2 counters of the trace kernel for one config under several ptk_set_option settings, side by side.
#   bash tools/pmc_quick.sh <outdir-under-gpurun_out> C4 64 "pool=0" "pool=256" ...
# Every rocprofv3 call profiles `python3 tools/c5_probe.py` directly; counters in --pmc passes of their own.
OUT=$PWD/gpurun_out/$1; ROOT=$PWD; CFG=$2; SPP=$3; shift 3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum")
a=0
for opts in "$@"; do
  a=$((a+1)); export PTK_OPTS="$opts"; export PTK_PROBE_REPS=2
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/arm$a/p$i -- python3 $ROOT/tools/c5_probe.py $CFG $SPP > $OUT/arm${a}_p$i.log 2>&1
  done
  echo "$opts" > $OUT/arm$a/opts.txt
done
cd $ROOT
python3 tools/pmc_quick_collect.py $OUT | tee $OUT/summary.txt
