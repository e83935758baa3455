#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in 20480 40960 65536 131072; do
python tools/bench_packed_pipeline.py $n 1073741824 1 $((n/2)) 2 16384 2 16384 3 8192 3 2>/dev/null
done
