#!/bin/bash
# usage: tools/pmc.sh <tag> [env...]; collects two PMC passes of tools/qb2.py
tag=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_1 -- python3 tools/qb2.py > gpurun_out/pmc_${tag}_1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_2 -- python3 tools/qb2.py > gpurun_out/pmc_${tag}_2.log 2>&1
