#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03ai_pytest.txt 2>&1; tail -6 gpurun_out/r03ai_pytest.txt
F="--no-cpu-baseline --no-object-api --no-host-io-leg --no-single-batch --no-configs-leg --wide-window 0"
echo "== cfg2 $(timeout 300 python bench.py $F --steps 64 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roundtrip_bit_exact'], d['roofline']['kernels_ms'], d['roofline']['frac'])")"
echo "== cfg2 driver form $(timeout 300 python bench.py $F --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roundtrip_bit_exact'])")"
for c in 3 4 5; do
echo "== cfg$c $(timeout 300 python bench.py --config $c --no-cpu-baseline --no-object-api --inflight 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roundtrip_bit_exact'], d['roofline']['kernels_ms'].get('k_miller_multi'), d['roofline']['frac'])")"
done
timeout 600 python tools/bench_schemes.py --only ghw11 --batch 1024 2>/dev/null | cut -c1-330
