#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python tools/bench_packed_pipeline.py 20480 1073741824 1 10240 2 > gpurun_out/r03w_20k.txt 2> gpurun_out/r03w_20k.err; cat gpurun_out/r03w_20k.txt
timeout 900 python tools/bench_packed_pipeline.py 131072 1073741824 1 65536 2 32768 2 32768 3 16384 3 > gpurun_out/r03w_131k.txt 2> gpurun_out/r03w_131k.err; cat gpurun_out/r03w_131k.txt
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_packed.py tests/test_gpu_packed_schemes.py -x -q -m gpu > gpurun_out/r03w_pytest.txt 2>&1; tail -15 gpurun_out/r03w_pytest.txt
