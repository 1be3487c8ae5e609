#!/bin/bash
# sq_rdb.sh SHAPE (e.g. 16,64,64): two rocprofv3 --pmc passes of 8 SQ counters each over the fused dense-block
# forward (tools/mb_rdb.py SHAPE), per-launch means of rdb_kernel into gpurun_out/sq_rdb_SHAPE.txt
R=$(pwd); S=$1; OUT=$R/gpurun_out/sq_rdb_$S.txt
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY"
P2="SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"
rm -rf $R/gpurun_out/sq1 $R/gpurun_out/sq2
timeout -s KILL 120 rocprofv3 --pmc $P1 -d $R/gpurun_out/sq1 --output-format csv -- python3 $R/tools/mb_rdb.py $S,6,2 > $R/gpurun_out/sq1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc $P2 -d $R/gpurun_out/sq2 --output-format csv -- python3 $R/tools/mb_rdb.py $S,6,2 > $R/gpurun_out/sq2.log 2>&1 || exit 1
cd $R
{ echo "rdb_kernel<false,true,16,8>, shape $S (B,H,W), per launch:"; python3 tools/pmc_mean.py gpurun_out/sq1 rdb_kernel; python3 tools/pmc_mean.py gpurun_out/sq2 rdb_kernel; } > $OUT
rm -rf gpurun_out/sq1 gpurun_out/sq2
cat $OUT
