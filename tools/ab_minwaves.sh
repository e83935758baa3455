for rep in 1 2; do
for f in build/variants/base.so build/variants/mw2.so; do
  for s in 1 8; do
    GPU_MAX_HW_QUEUES=8 RABE_HIP_LIB=$PWD/$f timeout 600 python bench.py --steps $((s*12)) --warmup 1 --no-cpu-baseline --no-host-io-leg --inflight $s 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['roofline']['kernels_ms']
print('$f', 'inflight', $s, 'ops/s', d['value'], 'ms/step', d['ms_per_step'], 'ok', d['roundtrip_bit_exact'], {a.replace('k_ac17_', ''): round(b, 2) for a, b in k.items()})
"
  done
done
done
