#!/bin/bash
# PMC passes for one Winograd kernel: tools/pmc_wino.sh <l1..l4> <fwd|wgrad> [B] [mem]   ("mem": + the L1 / TA single-block passes)
export TMPDIR=/tmp
cd /root/repo
. tools/pmc_lib.sh
SHAPE=$1; OP=$2; BB=${3:-12}
OUT=/root/repo/gpurun_out/pmc_wino_${SHAPE}_${OP}
rm -rf $OUT; mkdir -p $OUT
P="python3 /root/repo/tools/wino_one.py $SHAPE $OP $BB"
pmc_pass $OUT sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA -- $P || exit 1
pmc_pass $OUT sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD -- $P || exit 1
pmc_pass $OUT sq3 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -- $P || exit 1
pmc_pass $OUT fetch FETCH_SIZE -- $P || exit 1
pmc_pass $OUT tcc TCC_HIT_sum TCC_MISS_sum -- $P || exit 1
if [ "$4" = mem ]; then          # one or two counters of one block per pass (the six-counter TA / TCP lists of round 2 aborted)
  pmc_pass $OUT tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum -- $P
  pmc_pass $OUT tcp2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -- $P
  pmc_pass $OUT tcp3 TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum -- $P
  pmc_pass $OUT td1 TD_TC_STALL_sum TD_TD_BUSY_sum -- $P
fi
python3 tools/pmc_conv_report.py $OUT wino_$( [ "$OP" = fwd ] && echo fwd || echo wgrad )
