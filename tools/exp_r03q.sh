#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-object-api --no-host-io-leg --no-configs-leg --wide-window 0"
for v in "--steps 16" "--steps 20 --no-tail-overlap" "--steps 20"; do
timeout 300 python bench.py $F $v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['single_batch']
print('$v', d['value'], s['latency_ms'], s['ops_per_s_2_in_flight'], s['ops_per_s_4_in_flight'])"
done
for v in "--inflight 2" "--inflight 3" "--inflight 4"; do
timeout 300 python bench.py $F --no-single-batch --group 1 --steps 16 $v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('main loop $v', d['value'])"
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
