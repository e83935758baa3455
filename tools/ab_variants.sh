#!/bin/bash
# compiler-flag variants of engine_rr.hip (build/variants/lib<name>.so) against the product library on ONE box: config 2's full group and config 4
cd "$(dirname "$0")/.."
Q="--no-single-batch --no-configs-leg --no-host-io-leg --no-cpu-baseline --no-object-api --wide-window 0 --min-time 0.5"
for lib in "" "$@" ""; do
  a=$(RABE_BENCH_FULL_LINE=1 RABE_HIP_LIB=$lib python bench.py --steps 16 --warmup 16 $Q 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['roofline']['kernels_ms'];print(d['value'], k['k_miller_multi_rr'], k['k_final_exp_rr'], d['roundtrip_bit_exact'])")
  b=$(RABE_BENCH_FULL_LINE=1 RABE_HIP_LIB=$lib python bench.py --config 4 $Q 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['roofline']['kernels_ms'];print(d['value'], k['k_miller_multi_rr'], d['roundtrip_bit_exact'])")
  echo "${lib:-product}: cfg2 $a | cfg4 $b"
done
