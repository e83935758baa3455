"""Builds tests/hostsim/libhostsim.so (TEST INFRASTRUCTURE: the engine's host+device math run on the CPU)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostsim.hip")
LIB = os.path.join(HERE, "libhostsim.so")
HDR_DIR = os.path.join(os.path.dirname(os.path.dirname(HERE)), "rabe_amd", "csrc", "bn254")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [SRC] + [os.path.join(HDR_DIR, f) for f in os.listdir(HDR_DIR)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False):
    if force or _stale():
        subprocess.run(["hipcc", "-O2", "-std=c++17", "-DRB_COUNT_MULS", "-DRB29_CHECK", "--cuda-host-only", "-shared", "-fPIC", "-pthread", "-o", LIB, SRC],
                       check=True, timeout=600)
    return LIB


def load():
    return ctypes.CDLL(build())
