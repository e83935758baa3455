#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_packed.py tests/test_gpu_packed_schemes.py tests/test_gpu_validation.py tests/test_wire_compat.py -x -q -m gpu > gpurun_out/r03y_pytest.txt 2>&1; tail -5 gpurun_out/r03y_pytest.txt
for inl in 0 1; do
if [ $inl = 1 ]; then export RABE_MEMBER_INLINE=1; else unset RABE_MEMBER_INLINE; fi
echo "== config 2 inline=$inl $(timeout 500 python bench.py --steps 16 --no-cpu-baseline --no-single-batch --no-configs-leg --wide-window 0 --no-host-io-leg 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); o = d['object_api']['packed']; print(d['value'], o['ops_per_s'], o['ops_per_s_trusted'], o['encrypt_s'], o['decrypt_s'], o['decrypt_trusted_s'], o['plaintexts_match'])")"
for c in 3 4 5; do
echo "== config $c inline=$inl $(timeout 400 python bench.py --config $c --no-cpu-baseline --steps 4 --min-time 0.2 2> /dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); o = d.get('object_api'); print(d['value'], o['ops_per_s'], o['ops_per_s_trusted'], o.get('encrypt_s', o.get('keygen_s')), o['decrypt_s'], o['decrypt_trusted_s'], o['plaintexts_match'])")"
done
done
