#!/bin/bash
# a lone 4096-item AC17 batch (--group 1) and config 5 (small launches) per library, product library first and last, ONE box
cd "$(dirname "$0")/.."
Q="--no-single-batch --no-configs-leg --no-host-io-leg --no-cpu-baseline --no-object-api --wide-window 0 --min-time 0.5"
for lib in "" "$@" ""; do
  a=$(RABE_BENCH_FULL_LINE=1 RABE_HIP_LIB=$lib python bench.py --steps 6 --warmup 2 --group 1 $Q 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['roofline']['kernels_ms'];print(d['value'], d['ms_per_step'], d['roundtrip_bit_exact'], {a:round(b,3) for a,b in list(k.items())[:3]})")
  b=$(RABE_BENCH_FULL_LINE=1 RABE_HIP_LIB=$lib python bench.py --config 5 $Q 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['roofline']['kernels_ms'];print(d['value'], d['roundtrip_bit_exact'], {a:round(b,2) for a,b in k.items() if 'c6' in a})")
  echo "${lib:-product}: lone $a | cfg5 $b"
done
