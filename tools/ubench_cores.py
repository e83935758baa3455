#!/usr/bin/env python3
"""Shader cycles per call of the reduced-radix kernels' out-of-line routines at their own occupancy (one wave per SIMD, every CU busy):
rhip_debug_ubench_cores (engine_rr.hip: k_ubench_cores) of a DIAGNOSTIC build -- the product library does not contain the kernel:
    RABE_HIPCC_FLAGS=-DRB_UBENCH_CORES python -m rabe_amd.build --force && python tools/ubench_cores.py [iters=2000]
(int32_t rhip_debug_ubench_cores(rhip_ctx*, uint32_t iters, int32_t which, uint32_t blocks, uint64_t* d_out): d_out = blocks x 4 uint64 of device
memory, one per wave; which: 0 / 1 the line products' dot products (general / unit-y form), 2 the Fq2 multiplication, 3 the empty loop)"""
import ctypes
import struct
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from rabe_amd import Engine

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
eng = Engine(0)
blocks = 256
out = eng.alloc(blocks * 4 * 8)
names = {0: "rr_dot3_core", 1: "rr_dot3s_core", 2: "mul2_core", 3: "empty loop"}
res = {}
for rep in range(2):
    for which in (3, 0, 1, 2):
        eng._check(eng.lib.rhip_debug_ubench_cores(eng.ctx, ctypes.c_uint32(iters), ctypes.c_int32(which), ctypes.c_uint32(blocks), out.ptr))
        raw = eng.download(out, blocks * 4 * 8)
        v = struct.unpack("<%dQ" % (blocks * 4), raw)
        res[which] = (sum(v) / len(v) / iters, min(v) / iters, max(v) / iters)
for which in (3, 0, 1, 2):
    m, lo, hi = res[which]
    print("%-14s %8.1f cycles per call (min %.1f, max %.1f over %d waves)" % (names[which], m, lo, hi, blocks * 4))
