#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_packed.py tests/test_gpu_packed_schemes.py tests/test_gpu_validation.py -x -q -m gpu > gpurun_out/r03aa_pytest.txt 2>&1; tail -4 gpurun_out/r03aa_pytest.txt
for n in 20480 65536 131072; do
python tools/bench_packed_pipeline.py $n 1073741824 1 $((n/2)) 2 2>/dev/null
done
for c in 3 4 5; do
echo "== config $c $(timeout 400 python bench.py --config $c --no-cpu-baseline --steps 4 --min-time 0.2 2> /dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); o = d.get('object_api'); print(d['value'], o['ops_per_s'], o['ops_per_s_trusted'], o.get('encrypt_s', o.get('keygen_s')), o['decrypt_s'], o['decrypt_trusted_s'], o['plaintexts_match'])")"
done
