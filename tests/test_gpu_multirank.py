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
    env["RABE_PAIRING_MODE"] = "0"          # these tests are about sharding and the gather: the ranks run the automatic kernel selection, not the suite's cross-check
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


def run_bench_n(n, *extra, env_extra=None, timeout=1800):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    env["RABE_PAIRING_MODE"] = "0"          # (as above: several ranks share this box's one GPU; the cross-check would run every launch five times in each)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--warmup", "1", "--min-time", "0", "--no-cpu-baseline",
                          "--no-object-api", "--no-host-io-leg", "--no-single-batch", "--no-configs-leg", "--wide-window", "0"] + list(extra), env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]          # RCCL may print its own lines around rank 0's
    assert lines, out.stdout.decode()[-1500:] + out.stderr.decode()[-1500:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("extra", [("--steps", "4", "--g-window", "20"), ("--config", "4", "--steps", "1", "--batch", "128", "--attrs", "24", "--policies", "4")])
def test_rccl_collective_path_with_one_rank(extra):
    """the branch the 8-GPU run takes -- init_process_group("nccl"), barrier, all_reduce and all_gather_into_tensor on DEVICE tensors --
    executed here with a one-rank RCCL group (RABE_FORCE_DIST=1), so that the first multi-GPU run is not its first execution"""
    d = run_bench_n(1, *extra, env_extra={"RABE_FORCE_DIST": "1", "RABE_DIST_BACKEND": "nccl"})
    assert d["n_gpus"] == 1 and d["roundtrip_bit_exact"] is True
    assert d["gather"]["backend"] == "nccl" and d["gather"]["collective"] == "all_gather_into_tensor"
    assert d["gather"]["matches_unsharded_order"] is True


@pytest.mark.parametrize("config", [4, 5])
def test_eight_ranks_at_the_real_splits_of_configs_4_and_5(config):
    """BASELINE configs 4 / 5 as `--gpus 8` launches them: 16384 / 8 = 2048 and 8192 / 8 = 1024 items per rank, 200 attributes, eight
    ranks (sharing GPU 0 over gloo on this box), one step: shard_range's eight blocks, the gather and the unsharded-order check"""
    d = run_bench_n(8, "--config", str(config), "--steps", "1", timeout=3000)
    assert d["n_gpus"] == 8 and d["roundtrip_bit_exact"] is True
    assert d["gather"]["matches_unsharded_order"] is True
    assert d["config"]["batch_per_gpu"] == (2048 if config == 4 else 1024)
