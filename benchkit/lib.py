"""Shared plumbing of bench.py's configurations: group splitting, repeated exactly-K-step timed regions, rank-consistent
loop control, torch-owned device buffers.  Measurement only -- no group arithmetic here."""
import ctypes
import json
import sys
import time

G1_GEN = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
G2_GEN = b"".join(int(v).to_bytes(32, "little") for v in (
    10857046999023057135944570762232829481370756359578518086990519993285655852781,
    11559732032986387107991004021392285783925812861821192530917403151452391805634,
    8495653923123431417604973247489272438418190587263600148770280649306958101930,
    4082367875863433681332203403145435568316851327593401208105741076214120093531))
MAC_PER_FPMUL = 136          # 8-limb Montgomery multiplication: 2 * 8^2 + 8 32x32 multiply-adds


class ExtBuf:
    """A device buffer owned by a torch tensor (so that torch.distributed can gather it), seen as an engine buffer."""

    def __init__(self, torch_mod, nbytes, device):
        self.t = torch_mod.empty(max(int(nbytes), 4), dtype=torch_mod.uint8, device=device)
        self.ptr = ctypes.c_void_p(self.t.data_ptr())
        self.nbytes = int(nbytes)


def split_steps(k, gmax):
    """K steps -> group sizes: full groups of gmax first, the remainder last (a full group fills the chip exactly; for the
    driver's 20 steps 16 + 4 measures 982 k ops/s against 965 k for 10 + 10 and 959 k for one group of 20)."""
    sizes = [gmax] * (k // gmax)
    if k % gmax:
        sizes.append(k % gmax)
    return sizes


def same_on_all_ranks(flag):
    """rank 0's decision, everywhere (loop control of the repeated timed regions)"""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return bool(flag)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.broadcast(t, src=0)
    return bool(t.item())


def timed_regions(run_steps, sync_all, min_time, max_regions=200):
    """Repeats the region [barrier, run_steps() = exactly K steps, synchronize, barrier] until min_time seconds are covered;
    returns the list of max-over-ranks region times."""
    import torch
    import torch.distributed as dist
    from rabe_amd import shard

    def barrier():
        if dist.is_available() and dist.is_initialized():
            dist.barrier()

    regions = []
    while True:
        barrier()
        t0 = time.perf_counter()
        run_steps()
        sync_all()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        regions.append(shard.max_over_ranks(t1 - t0))
        if not same_on_all_ranks(sum(regions) < min_time and len(regions) < max_regions):
            break
    return regions


def regions_summary(regions, steps):
    mean = sum(regions) / len(regions)
    return {"count": len(regions), "steps_each": steps, "ms_min": round(1e3 * min(regions), 3), "ms_mean": round(1e3 * mean, 3),
            "ms_max": round(1e3 * max(regions), 3),
            "note": "every region times exactly --steps steps between barrier + synchronize; repeated until --min-time s are covered; value uses the mean"}


def cpu_model():
    """model name of the host CPU (stated with every CPU number)"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def emit_line(result):
    """rank 0's ONE JSON line, as the LAST line of stdout: RCCL writes a version banner through C stdio, which would otherwise be flushed
    at exit -- after the line (found on the GPU box, RABE_FORCE_DIST run) -- so everything buffered is flushed first."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(result), flush=True)
