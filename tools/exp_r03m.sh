#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ac17.py tests/test_gpu_bsw_dev.py tests/test_gpu_lsw_aw11_dev.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -4
F="--no-cpu-baseline --no-object-api --no-host-io-leg --no-configs-leg --wide-window 0"
timeout 300 python bench.py $F --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('drv', d['value'], d['ms_per_step'], d['config']['steps_per_launch_set'], d['roundtrip_bit_exact'], d['roofline']['kernels_ms'], d.get('single_batch'))"
for v in "--inflight 1" "--inflight 2"; do
timeout 300 python bench.py $F --no-single-batch --steps 64 $v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('64 $v', d['value'], d['ms_per_step'], d['config']['steps_per_launch_set'], d['roundtrip_bit_exact'])"
done
for v in "--group 1 --inflight 4 --steps 16" "--group 2 --inflight 4 --steps 16" "--group 4 --inflight 2 --steps 16" "--group 8 --inflight 2 --steps 16"; do
timeout 300 python bench.py $F --no-single-batch $v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', d['value'], d['ms_per_step'], d['roundtrip_bit_exact'])"
done
for c in 3 4 5; do
  timeout 300 python bench.py --config $c --steps 4 --min-time 0.3 --no-cpu-baseline --no-object-api 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print($c, d['value'], d['roofline']['kernels_ms'])"
done
