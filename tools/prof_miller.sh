#!/bin/bash
# region profile of k_miller_multi_rr (tools/build_diag.sh: -DRB_MILLER_PROF, see bn254/pairing29.h)
# usage: tools/prof_miller.sh [extra bench.py args, e.g. --config 3 --steps 8 --warmup 8]
cd "$(dirname "$0")/.."
Q="--no-single-batch --no-configs-leg --no-host-io-leg --no-cpu-baseline --no-object-api --wide-window 0"
A="${@:---steps 16 --warmup 16}"
RABE_MILLER_PROF=1 RABE_HIP_LIB=${PROF_LIB:-build/variants/libdiag.so} python bench.py $A $Q 2>&1 >/dev/null | grep -A6 "k_miller_multi_rr regions"
