#!/bin/bash
# PMC passes for the loss-chain kernels (separate passes as the MI355X guide prescribes; fast fail, see tools/pmc_lib.sh).
export TMPDIR=/tmp
cd /root/repo
. tools/pmc_lib.sh
OUT=/root/repo/gpurun_out/pmc_chain
rm -rf $OUT; mkdir -p $OUT
P="python3 /root/repo/tools/chain_bench.py 12 4"
pmc_pass $OUT sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM -- $P || exit 1
pmc_pass $OUT sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD -- $P || exit 1
if [ "$1" = mem ]; then
  pmc_pass $OUT fetch FETCH_SIZE -- $P || exit 1
  pmc_pass $OUT write WRITE_SIZE -- $P || exit 1
  pmc_pass $OUT tcc TCC_HIT_sum TCC_MISS_sum -- $P || exit 1
fi
python3 tools/pmc_summary.py $OUT chain_ > $OUT/summary.txt
grep -E "chain_(fwd|bwd)_kernel " $OUT/summary.txt
