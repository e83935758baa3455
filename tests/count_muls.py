#!/usr/bin/env python3
"""Instrumented Montgomery-multiplication counts of the engine's own code paths (run on the CPU build of
the same headers, tests/hostsim).  These are the "algorithmic work per unit" figures DESIGN.md and
bench.py's roofline use (SURVEY.md 8d asks for counts from an instrumented engine, not formulas)."""
import ctypes
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.hostsim import build as hs_build  # noqa: E402
from oracle import bn254 as bn  # noqa: E402

HS = hs_build.load()
HS.hs_mul_counter_reset.restype = ctypes.c_ulonglong
rnd = random.Random(1)


def b2c(b):
    return (ctypes.c_uint32 * (len(b) // 4)).from_buffer_copy(b)


def count(fn, *args, out=384):
    o = (ctypes.c_uint32 * (out // 4))()
    HS.hs_mul_counter_reset()
    getattr(HS, fn)(*[b2c(a) if isinstance(a, bytes) else a for a in args], o)
    return HS.hs_mul_counter_reset()


def le(x):
    return int(x).to_bytes(32, "little")


p = bn.g1_mul(bn.G1_GEN, rnd.randrange(1, bn.R))
q = bn.g2_mul(bn.G2_GEN, rnd.randrange(1, bn.R))
P, Q = bn.g1_to_le(p), bn.g2_to_le(q)
res = {}
base_io = count("hs_fp_add", le(1), le(2), out=32)           # load/store conversions of a trivial op
res["io_fp_roundtrip(2 loads + 1 store)"] = base_io
res["miller_loop(affine P) incl. 6 loads 12 stores"] = count("hs_miller", P, Q)
res["miller_loop(jacobian P) + final_exp"] = count("hs_pairing_jac", P, le(rnd.randrange(bn.P)), Q)
res["pairing via prepared lines (affine P) + final_exp incl. line preparation"] = count("hs_pairing_prepared", P, Q)
res["paired miller (A prepared incl. preparation, B jacobian) + final_exp"] = count("hs_pairing_pair", P, Q, P, Q)
res["paired miller, parked + merged lines (as k_ac17_dec_miller2) incl. preparation + final_exp"] = count("hs_pairing_pair_parked", P, Q, P, Q)
res["g2_prepare_lines (88 line triples) incl. io"] = count("hs_g2_prepare", Q, out=192)
m = (ctypes.c_uint32 * 96)()
HS.hs_miller(b2c(P), b2c(Q), m)
res["final_exponentiation incl. 12 loads 12 stores"] = count("hs_final_exp", bytes(m))
res["final_exponentiation over workspace slots, wNAF(3) chain (as k_final_exp) incl. 12 loads 12 stores"] = count("hs_final_exp_ws", bytes(m))
f = bn.gt_to_le(bn.pairing(p, q))
res["fp12_mul incl. 24 loads 12 stores"] = count("hs_fp12_mul", f, f)
res["fp12_cyclotomic_sqr incl. 12 loads 12 stores"] = count("hs_fp12_cyclotomic_sqr", f)
res["g1_mixed_add + to_affine(inv) incl. io"] = count("hs_g1_add", P, bn.g1_to_le(bn.G1_GEN), out=64)
res["fp_inv incl. io"] = count("hs_fp_inv", le(12345), out=32)
res["g1_mul_binary(254-bit) + to_affine incl. io"] = count("hs_g1_mul", P, le(rnd.randrange(bn.R)), out=64)
res["g2_mul_binary(254-bit) + to_affine incl. io"] = count("hs_g2_mul", Q, le(rnd.randrange(bn.R)), out=128)
res["gt_pow_binary(254-bit) incl. io"] = count("hs_gt_pow", f, le(rnd.randrange(bn.R)))
print(json.dumps(res, indent=1))
