#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ac17.py tests/test_gpu_packed.py tests/test_gpu_elements.py tests/test_gpu_fullsize_parity.py tests/test_gpu_hostops.py tests/test_gpu_validation.py -x -q -m gpu 2>&1 | tail -15
F="--no-cpu-baseline --no-object-api --no-host-io-leg"
run() { tag=$1; shift; python bench.py $F "$@" 2>gpurun_out/r03b_$tag.err | tail -1 > gpurun_out/r03b_$tag.json; python - <<P
import json
d=json.loads(open("gpurun_out/r03b_$tag.json").read())
print("$tag", d["value"], d["ms_per_step"], d["config"].get("steps_per_launch_set"), d.get("roofline",{}).get("kernels_ms"), d["tables"]["build_ms_per_public_key"])
P
}
run gw26 --steps 64
run gw16 --steps 64 --g-window 16
run gw18 --steps 64 --g-window 18
run gw20 --steps 64 --g-window 20
run gw22 --steps 64 --g-window 22
run gw24 --steps 64 --g-window 24
run drv26 --steps 20 --warmup 5
run drv20 --steps 20 --warmup 5 --g-window 20
bash tools/pmc_traffic.sh > gpurun_out/r03b_pmc_traffic.txt 2>&1
grep -i "enc_rows" gpurun_out/r03b_pmc_traffic.txt | head
