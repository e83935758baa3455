"""rabe_amd: MI355X-native batched ABE pairing / scalar-multiplication engine.

The product is the HIP library `rabe_amd/librabe_hip.so` (C ABI: include/rabe_hip.h) built from
rabe_amd/csrc/.  The Python modules here are thin ctypes plumbing over that ABI; nothing in this
package computes group arithmetic on the CPU, and nothing here imports the CPU checker (tests/test_package_isolation.py).
"""
from .engine import Engine, EngineError, lib_path  # noqa: F401
