#!/bin/bash
# one GPU session that produces everything profiles/ holds for a round and a config: the default bench line, the same command
# under rocprofv3 --kernel-trace --stats (per-kernel averages; must agree with roofline.kernels_ms of the line it printed),
# SQ issue counters and HBM traffic counters (separate --pmc passes).
# usage: profile_round.sh r02e [extra bench.py args, e.g. --config 3]
TAG=${1:-rXX}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cp bench_detail.json gpurun_out/${TAG}_bench_detail.json
rm -rf gpurun_out/prof_${TAG}
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG} -o x -- python bench.py --no-cpu-baseline --no-object-api --no-host-io-leg --no-single-batch --no-configs-leg --wide-window 0 "$@" > gpurun_out/${TAG}_bench_prof.json 2> gpurun_out/${TAG}_prof.err
cp bench_detail.json gpurun_out/${TAG}_bench_prof_detail.json
db=$(ls gpurun_out/prof_${TAG}/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/${TAG}_kernel_stats.csv
bash tools/pmc_sq.sh "$@" > gpurun_out/${TAG}_pmc_sq.txt 2>&1
bash tools/pmc_traffic.sh "$@" > gpurun_out/${TAG}_pmc_traffic.txt 2>&1
tail -2 gpurun_out/${TAG}_bench.json | cut -c1-600
