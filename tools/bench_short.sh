#!/bin/bash
# quick GPU check: parity tests + short bench summary (used during kernel iteration)
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 16 --warmup 1 --min-time 0.3 --no-cpu-baseline --no-object-api --no-host-io-leg "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ops/s', d['value'], 'ms/step', d['ms_per_step'], 'roundtrip', d['roundtrip_bit_exact'])
print('kernels_ms', d['roofline']['kernels_ms'])
print('roofline frac', d['roofline']['frac'], 'peak', d['roofline']['peak'])
"
