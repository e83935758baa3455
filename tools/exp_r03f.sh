#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in 3 4 5; do
  RABE_HOST_TIMING=1 python bench.py --config $c --steps 4 --min-time 0.3 --no-cpu-baseline > gpurun_out/r03f_cfg$c.json 2> gpurun_out/r03f_cfg$c.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r03f_cfg$c.json").read().strip().splitlines()[-1])
print($c, d["value"], json.dumps(d.get("object_api")))
P
  grep "host-timing" gpurun_out/r03f_cfg$c.err | tail -28
done
build/ubench_issue 2>&1 | tail -45
