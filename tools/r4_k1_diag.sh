#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# Round 4: what bounds K1 besides issue?  46 % of a wave's resident cycles are SQ_WAIT_ANY (r03_k1_counters.json) while the
# vector ALU is busy 40 % of the time and a sixth wave per SIMD buys 1 %: is it the vector-memory path (53 M scattered
# wave-instructions per launch, every lane in another 128-byte line of the [slot][lane] scratch)?  One counter group per pass.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r04_k1_diag.txt
: > $O
pass() { tag=$1; shift; timeout 300 bash tools/pmc_pass.sh $tag "$@" >> $O 2>&1 || echo "$tag: pass failed or timed out" >> $O; }
pass sq1 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
pass sq2 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_BRANCH SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
pass tcc2 TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum
cat $O
