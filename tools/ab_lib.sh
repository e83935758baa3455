#!/bin/bash
# A/B of two builds of the engine on ONE box: per-kernel ms of a full config-2 group and the config-3 launch set, alternating libraries
# usage: tools/ab_lib.sh build/variants/libbase.so [rounds=2]
cd "$(dirname "$0")/.."
Q="--no-single-batch --no-configs-leg --no-host-io-leg --no-cpu-baseline --no-object-api --wide-window 0"
for r in $(seq 1 ${2:-2}); do
  for lib in "$1" ""; do
    RABE_BENCH_FULL_LINE=1 RABE_HIP_LIB=$lib python bench.py --steps 16 --warmup 16 $Q 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('cfg2 lib=${lib:-current}', d['value'], d['roofline']['kernels_ms'])"
    RABE_BENCH_FULL_LINE=1 RABE_HIP_LIB=$lib python bench.py --config 3 --steps 8 --warmup 8 $Q 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('cfg3 lib=${lib:-current}', d['value'], {k:v for k,v in list(d['roofline']['kernels_ms'].items())[:3]})"
  done
done
