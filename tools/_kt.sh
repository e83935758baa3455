python -m pytest tests/test_gpu_rr2.py -x -q -k "pairing_jobs or infinity" 2>&1 | tail -3
for m in "$@"; do
python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-object-api --no-configs-leg --no-host-io-leg --no-single-batch --wide-window 0 --pairing-mode $m 2>&1 >/dev/null | python -c "
import sys, json
t=sys.stdin.read().split('bench_detail: ',1)[1].splitlines()[0]
d=json.loads(t); print('mode $m', d['value'], d['roundtrip_bit_exact'], d['roofline']['kernels_ms'])"
done
