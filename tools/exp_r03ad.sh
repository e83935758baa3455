#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lsw_aw11_dev.py tests/test_gpu_packed_schemes.py tests/test_gpu_fullsize_parity.py tests/test_gpu_schemes.py -x -q -m gpu > gpurun_out/r03ad_pytest.txt 2>&1; tail -4 gpurun_out/r03ad_pytest.txt
for b in 8 9 10 11 12 13; do
echo "== bits $b $(RABE_AW11_ATTR_BITS=$b timeout 400 python bench.py --config 5 --no-cpu-baseline --no-object-api --min-time 0.5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['kernels_ms']; print(d['value'], d['roundtrip_bit_exact'], k.get('k_aw11_enc_c1'), k.get('k_aw11_enc_c3'), d.get('tables'))")"
done
