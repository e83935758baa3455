#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_validation.py tests/test_gpu_packed_schemes.py tests/test_gpu_packed.py tests/test_gpu_ac17.py -x -q -m gpu 2>&1 | tail -8
F="--no-cpu-baseline --no-object-api --no-host-io-leg --no-single-batch --no-configs-leg --wide-window 0"
for v in "" "--no-tail-overlap"; do
  python bench.py $F --steps 20 --warmup 5 $v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('drv $v', d['value'], d['ms_per_step'], d['config']['steps_per_launch_set'], d['roundtrip_bit_exact'], d['roofline']['kernels_ms'])"
done
python bench.py $F --steps 36 --warmup 4 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('36', d['value'], d['ms_per_step'], d['config']['steps_per_launch_set'], d['roundtrip_bit_exact'])"
python bench.py $F --steps 64 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('64', d['value'], d['ms_per_step'], d['config']['steps_per_launch_set'], d['roundtrip_bit_exact'])"
for c in 3 5; do
  python bench.py --config $c --steps 4 --min-time 0.3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print($c, d['value'], json.dumps(d.get('object_api'))[:400])"
done
