#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_validation.py tests/test_gpu_ac17.py tests/test_gpu_bsw_dev.py tests/test_gpu_lsw_aw11_dev.py tests/test_gpu_fullsize_parity.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -6
F="--no-cpu-baseline --no-object-api --no-host-io-leg --no-configs-leg --wide-window 0"
for v in "" "--no-tail-overlap"; do
  timeout 300 python bench.py $F --steps 20 --warmup 5 $v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('drv $v', d['value'], d['ms_per_step'], d['config']['steps_per_launch_set'], d['roundtrip_bit_exact'], d.get('single_batch'))"
done
timeout 300 python bench.py $F --no-single-batch --steps 36 --warmup 4 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('36', d['value'], d['ms_per_step'], d['config']['steps_per_launch_set'], d['roundtrip_bit_exact'])"
for c in 3 4 5; do
  timeout 300 python bench.py --config $c --steps 4 --min-time 0.3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print($c, d['value'], d['roofline']['kernels_ms'], json.dumps(d.get('object_api'))[:330])"
done
