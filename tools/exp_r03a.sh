#!/bin/bash
# round-3 first GPU session: baseline numbers for the driver form and launch-granularity experiments
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
F="--no-cpu-baseline --no-object-api --no-host-io-leg"
run() { tag=$1; shift; echo "== $tag: $*"; env "${ENVV[@]}" python bench.py $F "$@" 2>gpurun_out/r03a_$tag.err | tail -1 > gpurun_out/r03a_$tag.json; python - <<P
import json
d=json.loads(open("gpurun_out/r03a_$tag.json").read())
print("$tag", d["value"], d["ms_per_step"], d["config"].get("steps_per_launch_set"), d["config"].get("launch_sets_in_flight"), d.get("roofline",{}).get("kernels_ms"))
P
}
ENVV=(X=1)
run drv --steps 20 --warmup 5
run drv_if2 --steps 20 --warmup 5 --inflight 2
ENVV=(RABE_BENCH_SPLIT=4,16); run drv_4_16_if2 --steps 20 --warmup 5 --inflight 2
ENVV=(RABE_BENCH_SPLIT=10,10); run drv_10_10 --steps 20 --warmup 5
ENVV=(RABE_BENCH_SPLIT=10,10); run drv_10_10_if2 --steps 20 --warmup 5 --inflight 2
ENVV=(X=1)
run g1 --steps 16 --group 1
run g1_if4 --steps 16 --group 1 --inflight 4
run g2_if4 --steps 16 --group 2 --inflight 4
run gw16 --steps 64 --g-window 16
run gw16_drv --steps 20 --warmup 5 --g-window 16
run def --steps 64
