"""bench.py's N > 1 path on a one-GPU box: `python bench.py --gpus 2` spawns its two ranks itself (they share GPU 0 and
rendezvous over gloo there; on a multi-GPU node the same code runs one rank per GPU over RCCL), shards the global batch,
gathers the result records with one all_gather_into_tensor and checks them against the unsharded order."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--warmup", "1", "--min-time", "0", "--no-cpu-baseline",
                          "--no-object-api", "--no-host-io-leg"] + list(extra), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    return json.loads(out.stdout.decode().strip().splitlines()[-1])


def test_two_ranks_config2():
    d = run_bench("--steps", "4", "--g-window", "20")
    assert d["n_gpus"] == 2 and d["roundtrip_bit_exact"] is True
    assert d["gather"]["matches_unsharded_order"] is True and d["gather"]["collective"] == "all_gather_into_tensor"
    assert d["value"] > 0 and d["steps"] == 4


def test_two_ranks_config4_lsw():
    d = run_bench("--config", "4", "--steps", "2", "--batch", "64", "--attrs", "24", "--policies", "4")
    assert d["n_gpus"] == 2 and d["roundtrip_bit_exact"] is True
    assert d["gather"]["matches_unsharded_order"] is True


def test_two_ranks_under_torch_distributed_run():
    """the driver's form for N > 1: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N (ranks from the environment)"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           "29531", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--g-window", "20", "--min-time", "0",
           "--no-cpu-baseline", "--no-object-api", "--no-host-io-leg"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    d = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["roundtrip_bit_exact"] is True and d["gather"]["matches_unsharded_order"] is True
