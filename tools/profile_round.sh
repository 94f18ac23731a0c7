#!/bin/bash
# Round profile refresh, run on the GPU box from the repo root:  bash tools/profile_round.sh <outdir-under-gpurun_out>
# Every step writes a file under gpurun_out/ so the run never looks silent.
OUT=$PWD/gpurun_out/${1:-prof}
ROOT=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 10 --warmup 2 > $OUT/bench_C2.json 2> $OUT/bench_C2.err
python3 $ROOT/bench.py --config C1 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_C1.json 2> $OUT/bench_C1.err
python3 $ROOT/bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_C3.json 2> $OUT/bench_C3.err
python3 $ROOT/bench.py --config C4 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_C4.json 2> $OUT/bench_C4.err
python3 $ROOT/bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_C5.json 2> $OUT/bench_C5.err
# (--opts overlap=0: per-launch durations are only meaningful un-overlapped; with the default overlap each launch also
#  waits for wave slots during its predecessor's tail and shows ~4 % longer)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --opts overlap=0 > $OUT/ktrace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --opts overlap=0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --opts overlap=0 > $OUT/pmc_write.log 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  for cfg in C2 C4; do
    rocprofv3 --pmc $set --output-format csv -d $OUT/sq_$cfg/p$i -- python3 $ROOT/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --opts overlap=0 > $OUT/sq_${cfg}_p$i.log 2>&1
  done
done
cd $ROOT
ls $OUT > $OUT/done.txt
