#!/bin/bash
# A/B of engine builds under build/variants/*.so: same bench, interleaved
for rep in 1 2; do
for f in build/variants/*.so; do
  for s in 1 8; do
    RABE_HIP_LIB=$PWD/$f timeout 600 python bench.py --steps 48 --warmup 1 --no-cpu-baseline --inflight $s 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['roofline']['kernels_ms']
print('$f', 'inflight', $s, 'ops/s', d['value'], 'ms/step', d['ms_per_step'], 'ok', d['roundtrip_bit_exact'], 'miller', k.get('k_ac17_dec_miller'), 'fe', k.get('k_final_exp'), 'rows', k.get('k_ac17_enc_rows'), 'cp', k.get('k_ac17_enc_cp'), 'c0', k.get('k_ac17_enc_c0'))
"
  done
done
done
