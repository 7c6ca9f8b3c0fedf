#!/bin/bash
# PMC passes for one Winograd kernel: tools/pmc_wino.sh <l1..l4> <fwd|wgrad> [B]
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_wino_$1_$2
rm -rf $OUT; mkdir -p $OUT
cd /root/repo
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 /root/repo/tools/wino_one.py $SHAPE $OP $BB > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
SHAPE=$1; OP=$2; BB=${3:-12}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD
run sq3 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_conv_report.py $OUT wino_$( [ "$OP" = fwd ] && echo fwd || echo wgrad )
