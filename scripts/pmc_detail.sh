#!/bin/bash
# per-dispatch SQ / memory counters of one bench configuration (run through gpurun) -> gpurun_out/pmcs/
#   bash scripts/pmc_detail.sh f32s
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcs
DT=${1:-f32s}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SMALL="--dtype $DT --no-cpu-baseline --no-profile --steps 1 --warmup 1 --frames-per-step 1"
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python $R/bench.py $SMALL > /dev/null 2> $OUT/p$i.err; echo "pmc $P rc=$?"
done
du -sh $OUT
