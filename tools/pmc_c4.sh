# PMC passes over tools/stats_probe.py for one config ($1, default C4); run from the repo root on the GPU box
CFG=${1:-C4}; SPP=${2:-64}
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_PERF_SEL_TOTAL_READ TCP_TCC_READ_REQ_LATENCY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $ROOT/gpurun_out/pmc_$CFG/p$i -- python3 $ROOT/tools/stats_probe.py $CFG $SPP > $ROOT/gpurun_out/pmc_$CFG/log$i.txt 2>&1 || echo "pass $i failed"
done
cd $ROOT && python3 tools/pmc_summary.py gpurun_out/pmc_$CFG/p*
