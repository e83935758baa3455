#!/bin/bash
# variants of the other device units against the product library on ONE box: configs 2-5 (top kernels) and a lone batch
cd "$(dirname "$0")/.."
Q="--no-single-batch --no-configs-leg --no-host-io-leg --no-cpu-baseline --no-object-api --wide-window 0 --min-time 0.5"
for lib in "" "$@" ""; do
  echo "== ${lib:-product}"
  RABE_BENCH_FULL_LINE=1 RABE_HIP_LIB=$lib python bench.py --steps 16 --warmup 16 $Q 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['roofline']['kernels_ms'];print('  cfg2', d['value'], {a:round(b,3) for a,b in k.items()})"
  RABE_BENCH_FULL_LINE=1 RABE_HIP_LIB=$lib python bench.py --steps 6 --warmup 2 --group 1 $Q 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['roofline']['kernels_ms'];print('  lone', d['value'], d['ms_per_step'], {a:round(b,3) for a,b in list(k.items())[:4]})"
  for c in 3 4 5; do
  RABE_BENCH_FULL_LINE=1 RABE_HIP_LIB=$lib python bench.py --config $c $Q 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['roofline']['kernels_ms'];print('  cfg$c', d['value'], {a:round(b,2) for a,b in list(k.items())[:5]})"
  done
done
