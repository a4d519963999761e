#!/bin/bash
# SQ counter passes for the dominant kernel (separate --pmc runs, kernel-trace only).
# (A pass with TCP_* / TA_* counters aborted rocprofv3 on this pool and hung until the time limit: SQ only.)
# usage (on the GPU box): bash tools/pmc_sq.sh <outdir>
OUT=${1:-gpurun_out/pmc_sq}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $line --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_workload.py > $GRAFT_REPO_ROOT/$OUT/p$i.log 2>&1
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC
SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
LIST
cd $GRAFT_REPO_ROOT
python3 tools/pmc_sq_parse.py $OUT
