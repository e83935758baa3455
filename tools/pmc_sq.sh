#!/bin/bash
# SQ (issue/stall) counters of the bench kernels, one batch at a time; counters in separate passes of <= 4
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/sq_$i -o t --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 "$@" > /dev/null 2> gpurun_out/sq_$i.err
done
python - <<'PY'
import csv, collections, glob
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob("gpurun_out/sq_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
ks = sorted({k for k, _ in agg if k.startswith("k_ac17") or k.startswith("k_final")})
for k in ks:
    print(k)
    for (kk, c), v in sorted(agg.items()):
        if kk == k:
            print("   %-28s %.4g per launch" % (c, v / cnt[(kk, c)]))
PY
