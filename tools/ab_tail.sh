#!/bin/bash
# A/B of the remainder group's schedule in the driver's form (--steps 20 --warmup 5): tail-mode pairing / final-exp / no overlap
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
Q="--no-single-batch --no-configs-leg --no-host-io-leg --no-cpu-baseline --no-object-api --wide-window 0"
for m in pairing final-exp; do
  RABE_BENCH_FULL_LINE=1 python bench.py --steps 20 --warmup 5 --tail-mode $m $Q 2>gpurun_out/ab_tail_$m.err | tail -1 > gpurun_out/ab_tail_$m.json
done

RABE_BENCH_FULL_LINE=1 python bench.py --steps 20 --warmup 5 --no-tail-overlap $Q 2>gpurun_out/ab_tail_none.err | tail -1 > gpurun_out/ab_tail_none.json
RABE_BENCH_FULL_LINE=1 python bench.py --steps 16 --warmup 16 $Q 2>gpurun_out/ab_tail_16.err | tail -1 > gpurun_out/ab_tail_16.json
for f in pairing final-exp none 16; do python - "$f" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/ab_tail_%s.json" % f).read())
    print(f, d["value"], d["ms_per_step"], d["roundtrip_bit_exact"], d["timed_regions"]["ms_mean"])
except Exception as e:
    print(f, "failed", e)
PY
done
