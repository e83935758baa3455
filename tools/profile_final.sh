#!/bin/bash
# the evidence set of a build (usage: profile_final.sh r05r): driver-form line, config 2's bench + rocprofv3 stats + SQ / HBM counters, the same for
# configs 3-5, the remainder-group timelines, the routine micro-benchmark and the region profile of the Miller kernel
TAG=${1:-rXXa}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_driver_form.json 2> gpurun_out/${TAG}_driver_form.err
bash tools/profile_round.sh ${TAG}
for c in 3 4 5; do
  bash tools/profile_round.sh ${TAG}_cfg$c --config $c --inflight 1
done
bash tools/trace_tail.sh "pairing final-exp"
cp gpurun_out/timeline_tail_pairing.txt gpurun_out/${TAG}_timeline_tail_miller_resident.txt
cp gpurun_out/timeline_tail_final-exp.txt gpurun_out/${TAG}_timeline_tail_final_exp.txt
bash tools/ab_tail.sh > gpurun_out/${TAG}_ab_tail.txt 2>&1
python tools/ubench_cores.py 2000 > gpurun_out/${TAG}_ubench_cores.txt 2>&1
[ -f build/variants/libdiag.so ] && bash tools/prof_miller.sh > gpurun_out/${TAG}_prof_miller.txt 2>&1
ls gpurun_out | grep ${TAG}
