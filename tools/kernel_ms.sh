#!/bin/bash
# per-kernel ms of a full config-2 group (HIP events, bench.py's roofline leg)
cd "$(dirname "$0")/.."
Q="--no-single-batch --no-configs-leg --no-host-io-leg --no-cpu-baseline --no-object-api --wide-window 0"
RABE_BENCH_FULL_LINE=1 python bench.py --steps 16 --warmup 16 $Q 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['roofline']['kernels_ms'])"
