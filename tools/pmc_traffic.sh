#!/bin/bash
# HBM traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limits), one batch at a time
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o t --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> gpurun_out/pmc_$c.err
done
python - <<'PY'
import csv, collections, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_%s/*counter_collection.csv" % c)[0]
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if r["Counter_Name"] == c:
            agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in ("k_ac17_dec_miller2", "k_ac17_dec_miller", "k_final_exp", "k_ac17_enc_rows", "k_ac17_enc_cp", "k_ac17_enc_c0"):
        if cnt[k]:
            print("%s %s per launch: %.1f KB-units (x1024 B = %.2f MB)" % (c, k, agg[k] / cnt[k], agg[k] / cnt[k] * 1024 / 1e6))
PY
