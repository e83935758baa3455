#!/bin/bash
# a round's profile session (usage: profile_all.sh r04a): driver-form line, then per config the bench line + rocprofv3 kernel stats + SQ / HBM counters,
# the issue-rate micro-benchmark and the host-layer stage timings of the packed entry points
TAG=${1:-rXXa}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# stdout = the ONE compact line the driver parses; the full object is bench_detail.json (benchkit/lib.py: emit_line)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_driver_form.json 2> gpurun_out/${TAG}_driver_form.err
cp bench_detail.json gpurun_out/${TAG}_driver_form_detail.json
bash tools/profile_round.sh ${TAG}
for c in 3 4 5; do
  bash tools/profile_round.sh ${TAG}_cfg$c --config $c --inflight 1
done
bash tools/profile_round.sh ${TAG}_cfg3_ragged --config 3 --ragged --inflight 1
RABE_BENCH_FULL_LINE=1 RABE_AW11_ATTR_W16=1 python bench.py --config 5 --no-cpu-baseline --no-object-api > gpurun_out/${TAG}_cfg5_w16_bench.json 2>/dev/null
RABE_BENCH_FULL_LINE=1 python bench.py --steps 16 --group 1 --inflight 4 --no-cpu-baseline --no-object-api --no-host-io-leg --no-single-batch --no-configs-leg --wide-window 0 > gpurun_out/${TAG}_group1_inflight4.json 2>/dev/null
RABE_HOST_TIMING=1 python tools/bench_packed_pipeline.py 65536 1073741824 1 > gpurun_out/${TAG}_ac17_packed_65536.json 2> gpurun_out/${TAG}_ac17_packed_65536_stages.txt
python tools/pcie_probe.py 256 > gpurun_out/${TAG}_pcie_probe.json 2>/dev/null
[ -x build/ubench_issue ] && build/ubench_issue > gpurun_out/${TAG}_ubench_issue.txt 2>&1
build/ubench_addc > gpurun_out/${TAG}_ubench_addc.txt 2>&1
ls gpurun_out | grep ${TAG} | head -60
