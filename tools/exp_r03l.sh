#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ac17.py tests/test_gpu_bsw_dev.py tests/test_gpu_lsw_aw11_dev.py tests/test_gpu_elements.py tests/test_gpu_batches.py -x -q -m gpu 2>&1 | tail -4
F="--no-cpu-baseline --no-object-api --no-host-io-leg --no-configs-leg --wide-window 0 --no-single-batch"
for v in "" "--no-tail-overlap"; do
  timeout 300 python bench.py $F --steps 20 --warmup 5 $v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('drv $v', d['value'], d['ms_per_step'], d['config']['steps_per_launch_set'], d['roundtrip_bit_exact'], d['roofline']['kernels_ms'])"
done
timeout 300 python bench.py $F --steps 64 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('64', d['value'], d['ms_per_step'], d['config']['steps_per_launch_set'], d['roundtrip_bit_exact'])"
bash tools/exp_r03k.sh 2>&1 | tail -22
