#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lanes in 1 2; do
for c in 3 4 5; do
echo "== config $c lanes $lanes $(RABE_PACKED_LANES=$lanes timeout 400 python bench.py --config $c --no-cpu-baseline --steps 4 --min-time 0.2 2> gpurun_out/r03x_cfg${c}_$lanes.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); o = d.get('object_api'); print(d['value'], o['ops_per_s'], o['ops_per_s_trusted'], o.get('encrypt_s', o.get('keygen_s')), o['decrypt_s'], o['decrypt_trusted_s'], o['plaintexts_match'])")"
done
done
