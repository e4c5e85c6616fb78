#!/bin/bash
tag=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmcb_${tag}_1 -- python3 tools/qb2.py > gpurun_out/pmcb_${tag}_1.log 2>&1
rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d gpurun_out/pmcb_${tag}_2 -- python3 tools/qb2.py > gpurun_out/pmcb_${tag}_2.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d gpurun_out/pmcb_${tag}_3 -- python3 tools/qb2.py > gpurun_out/pmcb_${tag}_3.log 2>&1
