"""Compiles the HIP engine for gfx950 in-tree: rabe_amd/csrc/engine.hip -> rabe_amd/librabe_hip.so.

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librabe_hip.so")
SOURCES = [os.path.join(CSRC, "engine.hip"), os.path.join(CSRC, "host", "schemes.cpp"), os.path.join(CSRC, "host", "host_abi.cpp")]


def _deps():
    inc = os.path.join(os.path.dirname(HERE), "include")
    deps = list(SOURCES) + [os.path.join(inc, "rabe_hip.h"), os.path.join(inc, "rabe_host.h")]
    for root, _dirs, files in os.walk(CSRC):
        deps += [os.path.join(root, f) for f in files]
    return deps


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False):
    if not (force or stale()):
        return LIB
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + SOURCES
    extra = os.environ.get("RABE_HIPCC_FLAGS")          # kernel-tuning experiments, e.g. -DRB_MIN_WAVES=3
    if extra:
        cmd += extra.split()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, timeout=1800)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
