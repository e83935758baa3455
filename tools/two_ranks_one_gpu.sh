#!/bin/bash
# functional check of bench.py's N > 1 path on a ONE-GPU box: two ranks (gloo rendezvous) share GPU 0
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 LOCAL_RANK=0 RABE_DIST_BACKEND=gloo GPU_MAX_HW_QUEUES=8
RANK=1 timeout 600 python bench.py --gpus 2 --steps 64 --warmup 1 --inflight 8 --g-window 24 --no-cpu-baseline --no-host-io-leg > /dev/null 2>&1 &
P=$!
RANK=0 timeout 600 python bench.py --gpus 2 --steps 64 --warmup 1 --inflight 8 --g-window 24 --no-cpu-baseline --no-host-io-leg 2>&1 | tail -1 | cut -c1-400
wait $P; echo "rank 1 exit $?"
