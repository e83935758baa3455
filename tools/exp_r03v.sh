#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F="--steps 16 --no-cpu-baseline --no-single-batch --no-configs-leg --wide-window 0 --no-host-io-leg"
run() { # name, env...
  name=$1; shift
  env "$@" RABE_HOST_TIMING=1 timeout 300 python bench.py $F > gpurun_out/r03v_$name.out 2> gpurun_out/r03v_$name.err
  echo "== $name rc=$? $(tail -1 gpurun_out/r03v_$name.out | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], json.dumps(d['object_api']['packed'])[:400])
except Exception as e: print('no line')")"
  grep -a "fault\|error\|Error" gpurun_out/r03v_$name.err | head -3
}
run lanes1_noarena_w16 RABE_PACKED_LANES=1 RABE_NO_ARENA=1 RABE_G_WINDOW=16
run lanes1_w16 RABE_PACKED_LANES=1 RABE_G_WINDOW=16
run lanes1 RABE_PACKED_LANES=1
run lanes2 RABE_PACKED_LANES=2
run lanes3 RABE_PACKED_LANES=3
run lanes4 RABE_PACKED_LANES=4
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_packed.py tests/test_gpu_packed_schemes.py -x -q -m gpu > gpurun_out/r03v_pytest.txt 2>&1; tail -15 gpurun_out/r03v_pytest.txt
