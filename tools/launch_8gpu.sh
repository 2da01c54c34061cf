#!/bin/bash
# One node, one process per GPU, RCCL over xGMI (backend "nccl"): the exact launch lines for N GPUs (default 8).
#   tools/launch_8gpu.sh bench  [N] [bench.py args...]      weak-scaling bench, prints ONE JSON line on rank 0
#   tools/launch_8gpu.sh train  [N] [train.py args...]      e.g. --config config/diffma_l2_synthetic.yaml --synthetic --autocast
#   tools/launch_8gpu.sh sample [N] [sample.py args...]     replicas: every rank samples its own shard, no collective
# HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver only supports dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise).
# DIFFMA_GRAD_COMPRESSION=bf16 (bench) / --grad-compression bf16 (train): opt-in 16-bit gradient all-reduce.
# Small per-GPU batches (the reference's brain.yaml: 1 sample per GPU): `bench ... --batch-per-gpu 1 --graph` / `train ... --graph-train`
# replay the step from two hipGraphs around one gradient all-reduce instead of the host-bound eager DDP step.
set -e
MODE=${1:-bench}; N=${2:-8}; shift 2 || true
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=${MASTER_PORT:-29531}
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT"
case $MODE in
  bench)  exec $RUN bench.py --gpus $N --steps ${STEPS:-20} --warmup ${WARMUP:-5} "$@" ;;
  train)  exec $RUN train.py "$@" ;;
  sample) exec $RUN sample.py "$@" ;;
  *) echo "usage: $0 bench|train|sample [N] [args...]"; exit 2 ;;
esac
