#!/bin/bash
# one-off counter passes (instruction cache, L2 hit rates, VMEM stalls) of the bench kernels; usage: pmc_extra.sh "<counters>" [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SET=$1; shift
rm -rf gpurun_out/px
timeout 900 rocprofv3 --pmc $SET --kernel-trace -d gpurun_out/px -o t --output-format csv -- python bench.py --steps 16 --warmup 16 --inflight 1 --min-time 0 --no-cpu-baseline --no-object-api --no-host-io-leg "$@" > /dev/null 2> gpurun_out/px.err
python - <<'PY'
import csv, collections, glob
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob("gpurun_out/px/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for (kk, c), v in sorted(agg.items()):
    if kk.startswith("k_ac17") or kk.startswith("k_miller") or kk.startswith("k_final"):
        print("%-28s %-28s %.4g per launch (%d launches)" % (kk, c, v / cnt[(kk, c)], cnt[(kk, c)]))
PY
tail -3 gpurun_out/px.err
