#!/bin/bash
# functional check of bench.py's N > 1 path on a ONE-GPU box: `--gpus 2` spawns two ranks that share GPU 0 (gloo rendezvous)
timeout 900 python bench.py --gpus 2 --steps 8 --warmup 1 --g-window 22 --min-time 0 --no-cpu-baseline --no-object-api 2>&1 | tail -1 | cut -c1-1200
