#!/bin/bash
# one GPU session that produces everything profiles/ holds for a round: bench JSONs, rocprofv3 kernel stats
# (default pipelining and one group at a time), SQ issue counters and HBM traffic counters.
# usage: profile_round.sh r02a [extra bench.py args, e.g. --config 3]
TAG=${1:-rXX}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --inflight 1 --no-cpu-baseline --no-object-api --no-host-io-leg "$@" > gpurun_out/${TAG}_bench_inflight1.json 2>> gpurun_out/${TAG}_bench.err
for mode in pipelined inflight1; do
  extra=""; [ $mode = inflight1 ] && extra="--inflight 1"
  rm -rf gpurun_out/prof_${TAG}_$mode
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_$mode -o x -- python bench.py --no-cpu-baseline --no-object-api --no-host-io-leg $extra "$@" > gpurun_out/${TAG}_bench_prof_$mode.json 2> gpurun_out/${TAG}_prof_$mode.err
  db=$(ls gpurun_out/prof_${TAG}_$mode/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/${TAG}_kernel_stats_$mode.csv
done
bash tools/pmc_sq.sh "$@" > gpurun_out/${TAG}_pmc_sq.txt 2>&1
bash tools/pmc_traffic.sh "$@" > gpurun_out/${TAG}_pmc_traffic.txt 2>&1
tail -2 gpurun_out/${TAG}_bench.json | cut -c1-600
