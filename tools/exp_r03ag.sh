#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ghw11.py -x -q -m gpu > gpurun_out/r03ag_pytest.txt 2>&1; tail -25 gpurun_out/r03ag_pytest.txt
RABE_HOST_TIMING=1 timeout 900 python tools/bench_schemes.py --only ghw11 --batch 1024 > gpurun_out/r03ag_ghw11.txt 2> gpurun_out/r03ag_ghw11.err; cat gpurun_out/r03ag_ghw11.txt; grep host-timing gpurun_out/r03ag_ghw11.err | grep ghw11 | tail -8
