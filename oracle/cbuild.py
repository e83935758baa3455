"""Builds oracle/c/rabe_ref.c -> oracle/_build/librabe_ref.so with gcc (TEST ORACLE / CPU baseline)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c", "rabe_ref.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "librabe_ref.so")


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(OUT_DIR, exist_ok=True)
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", LIB, SRC], check=True, timeout=300)
    return LIB
