#!/bin/bash
# tools/collect_counters.sh PREFIX WORKLOAD -- the counter passes of one frame of WORKLOAD (c3 | c4 | c5share, tools/pmc_frame.py),
# one counter block per pass, each pass a process of its own under its own timeout; joined with && so that a pass that is killed
# ends the call.  Output: gpurun_out/PREFIX_<pass>/; tools/pmc_evidence.py gpurun_out/PREFIX out.json turns them into the tracked JSON.
set -e
P=$1
export PRT_PMC_WORKLOAD=${2:-c3}
export PRT_PMC_TIMEOUT=${3:-180}
cd $GRAFT_REPO_ROOT
bash tools/pmc_pass.sh ${P}_sq "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" &&
bash tools/pmc_pass.sh ${P}_sq2 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" &&
bash tools/pmc_pass.sh ${P}_tcp "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum" &&
bash tools/pmc_pass.sh ${P}_tcc "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum" &&
bash tools/pmc_pass.sh ${P}_fetch "FETCH_SIZE" &&
bash tools/pmc_pass.sh ${P}_write "WRITE_SIZE" &&
bash tools/pmc_pass.sh ${P}_ta "TA_BUSY_avr TA_TA_BUSY_sum" &&
bash tools/pmc_pass.sh ${P}_ta2 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" &&
bash tools/pmc_pass.sh ${P}_grbm "GRBM_COUNT GRBM_GUI_ACTIVE" &&
echo "counters collected: $P $PRT_PMC_WORKLOAD"
