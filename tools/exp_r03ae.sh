#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lsw_aw11_dev.py -x -q -m gpu > gpurun_out/r03ae_pytest.txt 2>&1; tail -4 gpurun_out/r03ae_pytest.txt
bash tools/profile_round.sh r03d_cfg5 --config 5 --inflight 1
python bench.py --config 5 --no-cpu-baseline > gpurun_out/r03d_cfg5_inflight2_bench.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03d_driver_form.json 2> gpurun_out/r03d_driver_form.err
tail -1 gpurun_out/r03d_driver_form.json | cut -c1-200
