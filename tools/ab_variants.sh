#!/bin/bash
# A/B of engine builds under build/variants/*.so: same bench, interleaved
for rep in 1 2; do
for f in build/variants/*.so; do
  for s in 1 20; do
    RABE_HIP_LIB=$PWD/$f timeout 600 python bench.py --steps $((s*8)) --warmup 1 --no-cpu-baseline --inflight $s 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['roofline']['kernels_ms']
print('$f', 'inflight', $s, 'ops/s', d['value'], 'ms/step', d['ms_per_step'], 'ok', d['roundtrip_bit_exact'], {a.replace('k_ac17_', ''): round(b, 2) for a, b in k.items()})
"
  done
done
done
