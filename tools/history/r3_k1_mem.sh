#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/pmc_pass.sh ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum
bash tools/pmc_pass.sh tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr
bash tools/pmc_pass.sh tcc2 TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_WRITEBACK_sum TCC_TAG_STALL_sum
bash tools/pmc_pass.sh sq3 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
tail -3 gpurun_out/pmc_ta.log
