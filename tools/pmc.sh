#!/bin/bash
# usage: tools/pmc.sh <tag> [ENV=value ...]      (PMC_SCRIPT=tools/heavy_tail_time.py HT_BETA=2: another workload)
# SQ counters of one tools/stage_times.py run, one rocprofv3 pass per counter group
# (--pmc is never combined with the hip/hsa/sys traces).  Read with tools/pmc_show.py.
tag=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
pass() {
  n=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv \
    -d gpurun_out/pmc_${tag}_${n} -- python3 ${PMC_SCRIPT:-tools/stage_times.py} > gpurun_out/pmc_${tag}_${n}.log 2>&1
  echo "pass $n done"
}
pass 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS &&
pass 2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE &&
pass 3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
