#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_bsw_dev.py tests/test_gpu_packed_schemes.py tests/test_gpu_fullsize_parity.py tests/test_gpu_packed_fuzz.py tests/test_gpu_pipeline.py tests/test_gpu_schemes.py -x -q -m gpu > gpurun_out/r03az_pytest.txt 2>&1; tail -12 gpurun_out/r03az_pytest.txt | cut -c1-250
for v in "" "RABE_BSW_GENERAL_DECRYPT=1"; do
echo "== cfg3 [$v] $(env $v timeout 300 python bench.py --config 3 --no-cpu-baseline --inflight 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); o = d['object_api']; print(d['value'], d['roundtrip_bit_exact'], d['roofline']['kernels_ms'], o['ops_per_s'], o['ops_per_s_trusted'])")"
done
