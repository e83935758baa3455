#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_carry_interlock.py tests/test_gpu_packed.py tests/test_gpu_elements.py tests/test_gpu_hostops.py -x -q -m gpu 2>&1 | tail -15
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r03c_drv.json 2> gpurun_out/r03c_drv.err ) 2>&1 | tail -3
tail -c 600 gpurun_out/r03c_drv.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03c_drv.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","single_batch","value_wide_tables","configs","object_api"):
    print(k, json.dumps(d.get(k))[:1500])
print("roofline", {k:d["roofline"][k] for k in ("kernel","kernel_ms","frac","frac_survey","peak")})
P
