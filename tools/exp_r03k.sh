#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/trace_tail
F="--no-cpu-baseline --no-object-api --no-host-io-leg --no-configs-leg --wide-window 0 --no-single-batch"
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/trace_tail -o t --output-format csv -- python bench.py $F --steps 20 --warmup 20 --min-time 0 > /dev/null 2> gpurun_out/trace_tail.err
python - <<'P'
import csv, glob
f = glob.glob("gpurun_out/trace_tail/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if r["Kernel_Name"].startswith("k_") and not r["Kernel_Name"].startswith("k_table_build") and not r["Kernel_Name"].startswith("k_calib")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 2 regions' worth of kernels
t0 = int(rows[-40]["Start_Timestamp"])
for r in rows[-40:]:
    print("%-22s q=%s  start %9.3f ms  end %9.3f ms  dur %7.3f  grid %s" % (r["Kernel_Name"][:22], r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size")))
P
