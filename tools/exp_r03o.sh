#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-object-api --no-host-io-leg --no-configs-leg --wide-window 0 --no-single-batch"
for lib in ""; do
  RABE_HIP_LIB=$lib timeout 300 python bench.py $F --steps 32 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('lib=$lib', d['value'], d['ms_per_step'], d['roundtrip_bit_exact'], d['roofline']['kernels_ms'])"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-object-api --no-host-io-leg --no-configs-leg --wide-window 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('single', d['value'], d['single_batch'])"
