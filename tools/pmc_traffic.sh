#!/bin/bash
# HBM traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limits), one full group
# (one launch set that fills the chip) at a time.  usage: pmc_traffic.sh [extra bench.py args, e.g. --config 3]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o t --output-format csv -- python bench.py --steps 16 --warmup 16 --inflight 1 --min-time 0 --no-cpu-baseline --no-object-api --no-host-io-leg --no-single-batch --no-configs-leg --wide-window 0 "$@" > /dev/null 2> gpurun_out/pmc_$c.err
done
python - <<'PY'
import csv, collections, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_%s/*counter_collection.csv" % c)[0]
    agg = collections.defaultdict(float); cnt = collections.Counter(); mx = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if r["Counter_Name"] == c:
            v = float(r["Counter_Value"])
            agg[k] += v; cnt[k] += 1; mx[k] = max(mx[k], v)
    for k in sorted(agg):
        if k.startswith("k_") and not k.startswith("k_table_build") and not k.startswith("k_calib"):
            # launches differ in size (set-up launches of a few items beside the full groups): report the largest
            print("%s %s per launch: %.1f KB-units (x1024 B = %.2f MB; largest of %d launches, mean %.1f)" % (c, k, mx[k], mx[k] * 1024 / 1e6, cnt[k], agg[k] / cnt[k]))
PY
