#!/bin/bash
# stage timings of the packed entry points (RABE_HOST_TIMING=1) for configs 3-5
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in 3 4 5; do
echo "== config $c"
RABE_HOST_TIMING=1 timeout 400 python bench.py --config $c --no-cpu-baseline --steps 4 --min-time 0.2 2> gpurun_out/r03t_cfg$c.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], json.dumps(d.get('object_api')))"
grep host-timing gpurun_out/r03t_cfg$c.err | tail -40
done
