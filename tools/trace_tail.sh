#!/bin/bash
# kernel timeline of the last timed region of the driver's form, per remainder-group schedule (rocprofv3 --kernel-trace; tools/timeline.py)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
Q="--no-single-batch --no-configs-leg --no-host-io-leg --no-cpu-baseline --no-object-api --wide-window 0 --min-time 0.2"
for m in ${1:-pairing final-exp}; do
  rm -rf /tmp/tr_$m
  rocprofv3 --kernel-trace -d /tmp/tr_$m -o x -- python bench.py --steps 20 --warmup 5 --tail-mode $m $Q > gpurun_out/trace_$m.log 2>&1
  db=$(find /tmp/tr_$m -name "*.db" | head -1)
  python tools/timeline.py "$db" 75 k_final_exp_c6 ${TL_BACK:-20} > gpurun_out/timeline_tail_$m.txt 2>&1
done
