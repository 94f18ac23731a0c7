#!/bin/bash
# GPU box: instruction-cache / scalar-cache / instruction-fetch counters of the trace kernel for one config.
#   bash tools/pmc_icache.sh <outdir-under-gpurun_out> C4 64
OUT=$PWD/gpurun_out/$1; ROOT=$PWD; CFG=$2; SPP=$3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PTK_PROBE_REPS=2
SETS=("SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/arm1/p$i -- python3 $ROOT/tools/c5_probe.py $CFG $SPP > $OUT/arm1_p$i.log 2>&1
done
echo "default" > $OUT/arm1/opts.txt
cd $ROOT
python3 tools/pmc_quick_collect.py $OUT | tee $OUT/summary.txt
