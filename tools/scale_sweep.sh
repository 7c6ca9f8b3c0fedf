#!/bin/bash
# The four one-variable experiments of DESIGN.md section 8 for an 8-GPU node (this round's boxes have one GPU: NOT run here; the
# N > 1 numbers of this repo come from the driver's SCALE run).  Each line of gpurun_out/scale_sweep/<experiment>.jsonl is one
# bench.py JSON line; the baseline arm is bench.py's default at that N.
#   usage: tools/scale_sweep.sh [N ...]        (default: 2 4 8)
# One variable per arm against the default (direct RCCL communicator, 6 hardware queues, two weight-gradient side streams):
#   allreduce : DVS_ALLREDUCE=torch            torch.distributed's process group instead of include/dvslam_rccl.h
#   algo      : NCCL_ALGO=Ring | Tree          RCCL's algorithm choice for the 107 MB of gradient buckets (0.18 vs 1.23 ms)
#   queues    : GPU_MAX_HW_QUEUES=4            the all-reduce stream shares a hardware queue with a compute stream
#   wgstream  : DVS_WGRAD_STREAM=shared        three compute streams + RCCL = the default four queues
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/scale_sweep
mkdir -p $OUT
NS=${@:-2 4 8}
PORT=29511
run() {   # run <experiment> <label> <N> [VAR=value ...]
  local exp=$1 label=$2 n=$3; shift 3
  echo "== $exp / $label / N=$n" >&2
  env HSA_ENABLE_IPC_MODE_LEGACY=0 "$@" timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
      --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $n --steps 50 --warmup 10 --no-kernel-timing 2> $OUT/$exp.$label.$n.err \
    | grep '^{' | sed "s/^{/{\"arm\": \"$label\", /" >> $OUT/$exp.jsonl
  PORT=$((PORT + 1))
}
for n in $NS; do
  run baseline default $n
  run allreduce torch $n DVS_ALLREDUCE=torch
  run algo ring $n NCCL_ALGO=Ring
  run algo tree $n NCCL_ALGO=Tree
  run queues q4 $n GPU_MAX_HW_QUEUES=4
  run wgstream shared $n DVS_WGRAD_STREAM=shared
done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/scale_sweep/*.jsonl")):
    for l in open(f):
        d = json.loads(l)
        print("%-28s %-8s N=%d  %8.1f frames/s  %6.2f ms/step  (%s)" % (f.split("/")[-1], d["arm"], d["n_gpus"], d["value"], d["ms_per_step"],
                                                                       d["config"]["allreduce"]))
PY
