#!/bin/bash
# SQ (issue/stall) counters of the bench kernels, one full group (one launch set that fills the chip) at a time;
# counters in separate passes of <= 4.  usage: pmc_sq.sh [extra bench.py args, e.g. --config 3]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rm -rf gpurun_out/sq_$i
  timeout 900 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/sq_$i -o t --output-format csv -- python bench.py --steps 16 --warmup 16 --inflight 1 --min-time 0 --no-cpu-baseline --no-object-api --no-host-io-leg --no-single-batch --no-configs-leg --wide-window 0 "$@" > /dev/null 2> gpurun_out/sq_$i.err
done
python - <<'PY'
import csv, collections, glob
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob("gpurun_out/sq_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
ks = sorted({k for k, _ in agg if k.startswith("k_") and not k.startswith("k_table_build") and not k.startswith("k_calib")})
for k in ks:
    print(k)
    for (kk, c), v in sorted(agg.items()):
        if kk == k:
            print("   %-28s %.4g per launch (%d launches)" % (c, v / cnt[(kk, c)], cnt[(kk, c)]))
PY
