#!/bin/bash
# BASELINE configs[3] on the 8 GPUs of one node: one process per GPU, RCCL over xGMI, the gradient exchange recorded into the replayed
# backward pass.  The JSON line carries clips/s (whole job), exchange_ms (blocking form), exchange_overlapped_fraction, ranks_seen.
#   bash tools/run_ddp8.sh [N=8] [steps=20] [warmup=8]
cd "$(dirname "$0")/.."
N=${1:-8}; K=${2:-20}; W=${3:-8}
export HSA_ENABLE_IPC_MODE_LEGACY=0
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29517}" \
  bench.py --mode train --gpus "$N" --steps "$K" --warmup "$W"
