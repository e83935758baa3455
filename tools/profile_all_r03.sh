#!/bin/bash
# round-3 profile session: driver-form line, then per config the bench line + rocprofv3 kernel stats + SQ / HBM counters
TAG=${1:-r03a}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_driver_form.json 2> gpurun_out/${TAG}_driver_form.err
bash tools/profile_round.sh ${TAG}
for c in 3 4 5; do
  bash tools/profile_round.sh ${TAG}_cfg$c --config $c --inflight 1
done
ls -la gpurun_out | grep ${TAG} | head -40
