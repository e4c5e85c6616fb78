#!/bin/bash
tag=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 --kernel-trace --output-format csv -d gpurun_out/pmcc_${tag}_1 -- python3 tools/qb2.py > gpurun_out/pmcc_${tag}_1.log 2>&1
echo pass1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmcc_${tag}_2 -- python3 tools/qb2.py > gpurun_out/pmcc_${tag}_2.log 2>&1
echo pass2
