#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-object-api --no-host-io-leg --no-single-batch --no-configs-leg --wide-window 0"
for w in 20 21 22 23 24; do
echo "== w=$w $(timeout 300 python bench.py $F --steps 20 --warmup 5 --g-window $w 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernels_ms'].get('k_ac17_enc_rows'), d['tables'])")"
done
