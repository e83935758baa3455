#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-object-api --no-host-io-leg --no-configs-leg --wide-window 0"
for v in "--steps 20" "--steps 16"; do
timeout 300 python bench.py $F $v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['single_batch']
print('$v', d['value'], d['roundtrip_bit_exact'], s['latency_ms'], s['ops_per_s_2_in_flight'], s['ops_per_s_4_in_flight'], s['roundtrip_bit_exact'])"
done
