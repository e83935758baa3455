"""bench.py on the GPU prints ONE stdout line -- the compact object the driver parses (benchkit/lib.py) -- and leaves the full object in
bench_detail.json; a short run of the default configuration with the informational legs switched off."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_compact_line_on_stdout_and_the_detail_beside_it():
    env = dict(os.environ, RABE_PAIRING_MODE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RABE_BENCH_FULL_LINE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--min-time", "0", "--cpu-sample", "2",
                          "--no-object-api", "--no-host-io-leg", "--no-single-batch", "--no-configs-leg", "--wide-window", "0"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = out.stdout.decode().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096, (len(lines), [len(l) for l in lines])
    d = json.loads(lines[0])
    assert d["metric"].startswith("ABE ops/sec") and d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 4 and d["roundtrip_bit_exact"] is True
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["kernel"] and rf["kernel_ms"] > 0 and 0 < rf["frac"] < 1 and rf["peak"] > 0 and rf["traffic"] and os.path.exists(os.path.join(ROOT, rf["traffic_source"]))
    assert 0 < rf["frac_valu_issue"] < 1
    assert cb["value"] > 0 and cb["cores"] == 1 and cb["kind"] == "port"
    full = json.load(open(os.path.join(ROOT, "bench_detail.json")))
    assert full["value"] == d["value"] and full["roofline"]["kernels_ms"] and full["timed_regions"]["count"] >= 1
    assert out.stderr.decode().count("bench_detail: ") == 1
