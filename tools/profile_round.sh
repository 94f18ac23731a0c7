#!/bin/bash
# THE script that regenerates a round's profiles/<round>/ (VERDICT r03 item 7).  Two steps:
#   1. on the GPU box, from the repo root:   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r04 [what ...]'
#        what = bench ktrace traffic sq class mem   (default: bench ktrace traffic sq class); raw output -> gpurun_out/prof_r04/
#   2. here:                                  python tools/profile_collect.py gpurun_out/prof_r04 profiles/r04
#        summaries (bench lines, kernel-stats CSVs, traffic / SQ / class counters), each tied to the kernel sources by sha256
# then `python tools/design_table.py profiles/r04` prints DESIGN.md 7's tables (incl. registers / LDS / spills read from libptk.so).
# Every rocprofv3 call profiles `python3 bench.py` directly (no shell / env hop after `--`), counters in passes of their
# own (--pmc only, no trace flags), each pass into its own directory.
ROUND=${1:-r04}
OUT=$PWD/gpurun_out/prof_$ROUND
ROOT=$PWD
shift
WHAT=${@:-bench ktrace traffic sq class}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
B="--no-cpu-baseline --no-other-configs --no-parity --no-contracted --no-interactive --opts overlap=0"
steps_of() { case $1 in C5) echo "--steps 1 --warmup 0";; C3|C4) echo "--steps 2 --warmup 1";; *) echo "--steps 3 --warmup 1";; esac; }

if has bench; then
  python3 $ROOT/bench.py --steps 100 --warmup 5 > $OUT/bench_C2.json 2> $OUT/bench_C2.err
  python3 $ROOT/bench.py --config C1 --steps 200 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench_C1.json 2> $OUT/bench_C1.err
  python3 $ROOT/bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_C3.json 2> $OUT/bench_C3.err
  python3 $ROOT/bench.py --config C4 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_C4.json 2> $OUT/bench_C4.err
  python3 $ROOT/bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_C5.json 2> $OUT/bench_C5.err
fi
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.kernel_source_sha256())" > $OUT/kernel_source_sha256.txt
if has ktrace; then
  # the DEFAULT run (overlap = 1): consecutive launches on two streams, for profiles/<round>/overlap_trace_C2.json
  timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/ktrace_overlap_C2 -- python3 $ROOT/bench.py --config C2 --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-contracted --no-interactive > $OUT/ktrace_overlap_C2.log 2>&1
  for cfg in C2 C4; do
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace_$cfg -- python3 $ROOT/bench.py --config $cfg $(steps_of $cfg) $B > $OUT/ktrace_$cfg.log 2>&1
  done
fi
CFGS_TRAFFIC=${CFGS_TRAFFIC:-C2 C3 C4 C5}
if has traffic; then
  for cfg in $CFGS_TRAFFIC; do
    i=0
    for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
      i=$((i+1))
      timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/traffic_$cfg/p$i -- python3 $ROOT/bench.py --config $cfg $(steps_of $cfg) $B > $OUT/traffic_${cfg}_p$i.log 2>&1
    done
  done
fi
CFGS_SQ=${CFGS_SQ:-C2 C4 C5}
if has sq; then
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    for cfg in $CFGS_SQ; do
      timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/sq_$cfg/p$i -- python3 $ROOT/bench.py --config $cfg $(steps_of $cfg) $B > $OUT/sq_${cfg}_p$i.log 2>&1
    done
  done
fi
if has class; then
  i=0
  for set in "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
             "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" \
             "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH SQ_INSTS_VALU"; do
    i=$((i+1))
    for cfg in $CFGS_SQ; do
      timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/class_$cfg/p$i -- python3 $ROOT/bench.py --config $cfg $(steps_of $cfg) $B > $OUT/class_${cfg}_p$i.log 2>&1
    done
  done
fi
if has mem; then
  i=0
  # one or two counters per pass: a larger TA / TCP set is refused ("exceeds the capabilities of the hardware") and the
  # profiler then hangs instead of exiting, hence the timeouts
  for set in "TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum" \
             "TCP_PENDING_STALL_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    for cfg in ${CFGS_MEM:-C4 C5}; do
      timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/mem_$cfg/p$i -- python3 $ROOT/bench.py --config $cfg $(steps_of $cfg) $B > $OUT/mem_${cfg}_p$i.log 2>&1
    done
  done
fi
cd $ROOT
ls $OUT > $OUT/done.txt
