#!/bin/bash
# On the GPU box, from the repo root:  bash tools/microbench/run.sh <outdir-under-gpurun_out>
# wall-clock tables first, then the SQ counters of the same kernels at 4 waves per SIMD (separate --pmc passes).
OUT=$PWD/gpurun_out/${1:-micro}
ROOT=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 $ROOT/tools/microbench/valu_calib 1 2 4 8 > $OUT/valu_calibration.json 2> $OUT/valu_calibration.err || exit 1
timeout -k 10 300 $ROOT/tools/microbench/gather_bench > $OUT/gather_ceiling.json 2> $OUT/gather_ceiling.err || exit 1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/valu_pmc/p$i -- $ROOT/tools/microbench/valu_calib 4 > $OUT/valu_pmc_p$i.log 2>&1 || exit 1
done
ls $OUT > $OUT/done.txt
