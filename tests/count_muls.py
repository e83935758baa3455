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
# the same 88 line triples by AFFINE steps, the shared inversion itself not charged, Montgomery's trick (3 Fq2
# products per element and step) charged -- to be compared with the projective preparation above
res["g2 lines by affine steps + 3 Fq2 products per step for a simultaneous inversion (inversion itself not counted) incl. io"] = \
    count("hs_g2_prepare_affine", Q, out=192)
_a, _b = (ctypes.c_uint32 * 96)(), (ctypes.c_uint32 * 96)()
HS.hs_pairing_affine_lines(b2c(P), b2c(Q), _a)
HS.hs_pairing_prepared(b2c(P), b2c(Q), _b)
assert bytes(_a) == bytes(_b), "affine lines give another pairing value"
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

# ---- the job kernels' code paths (engine_jobs.hip): counts per lane; conversions of the harness (loads / stores / line preparation)
# are subtracted so that the figures are what a device lane executes
def count_multi(kinds):
    n = len(kinds)
    ps = b"".join(bn.g1_to_le(bn.g1_mul(bn.G1_GEN, rnd.randrange(1, bn.R))) for _ in range(n))
    qs = b"".join(bn.g2_to_le(bn.g2_mul(bn.G2_GEN, rnd.randrange(1, bn.R))) for _ in range(n))
    o = (ctypes.c_uint32 * 96)()
    HS.hs_mul_counter_reset()
    HS.hs_pairing_multi(n, (ctypes.c_int * n)(*kinds), b2c(ps), b2c(qs), o)
    total = HS.hs_mul_counter_reset()
    prep = res["g2_prepare_lines (88 line triples) incl. io"] - 4 - 6            # one preparation without its loads (4) / stores (6)
    return total - n * (2 + 4) - sum(1 for k in kinds if k == 1) * prep - res["final_exponentiation incl. 12 loads 12 stores"] + 12


jobs = {}
jobs["miller_loop_multi, 14 pairs: 7 prepared + 7 walking (a bsw chunk with a prepared key)"] = count_multi([1, 0] * 7)
jobs["miller_loop_multi, 14 walking pairs (bsw / lsw / aw11 chunk, nothing prepared)"] = count_multi([0] * 14)
jobs["miller_loop_multi, 6 pairs: 3 prepared + 3 walking (an ac17 item with a prepared key)"] = count_multi([1, 0] * 3)
jobs["miller_loop_multi, 14 prepared pairs (a ghw11 transform chunk: every G2 argument is the transform key's)"] = count_multi([1] * 14)
jobs["miller_loop_multi, 6 walking pairs (an ac17 item, nothing prepared)"] = count_multi([0] * 6)
jobs["miller_loop_multi, 2 walking pairs"] = count_multi([0, 0])
jobs["miller_loop_multi, 1 walking pair"] = count_multi([0])
jobs["miller_loop_multi, 1 prepared pair"] = count_multi([1])
k_full = rnd.randrange(bn.R)
jobs["jac_mul_naf G1 (254-bit) + to_affine incl. io"] = count("hs_g1_mul_naf", P, le(k_full), out=64)
jobs["jac_mul_naf G2 (254-bit) + to_affine incl. io"] = count("hs_g2_mul_naf", Q, le(k_full), out=128)
jobs["jac_mul_naf G1 (k = 2) + to_affine incl. io"] = count("hs_g1_mul_naf", P, le(2), out=64)
n = 16
pts = b"".join(bn.g1_to_le(bn.g1_mul(bn.G1_GEN, rnd.randrange(1, bn.R))) for _ in range(n))
ks = b"".join(le(rnd.randrange(bn.R)) for _ in range(n))
o = (ctypes.c_uint32 * 16)()
HS.hs_mul_counter_reset()
HS.hs_g1_msm(n, b2c(pts), b2c(ks), o)
jobs["jac_msm_naf G1, 16 terms (254-bit) + to_affine incl. io"] = HS.hs_mul_counter_reset()
print(json.dumps(jobs, indent=1))

# ---- the reduced-radix kernels (engine_rr.hip): multiply-add INSTRUCTIONS per lane (81 per schoolbook product of two 9-limb elements,
# 81 per reduction) -- what k_miller_multi_rr / k_final_exp_rr issue; the harness's conversion of prepared lines (done once per key
# handle on the device: rhip_lines_to_rr) is subtracted
HS.hs_rr_mad_counter_reset.restype = ctypes.c_ulonglong


def rr_mads_multi(kinds):
    n = len(kinds)
    ps = b"".join(bn.g1_to_le(bn.g1_mul(bn.G1_GEN, rnd.randrange(1, bn.R))) for _ in range(n))
    qs = b"".join(bn.g2_to_le(bn.g2_mul(bn.G2_GEN, rnd.randrange(1, bn.R))) for _ in range(n))
    o = (ctypes.c_uint32 * 96)()
    HS.hs_rr_mad_counter_reset()
    HS.hs_rr_miller_multi(n, (ctypes.c_int * n)(*kinds), b2c(ps), b2c(qs), o, None)
    total = HS.hs_rr_mad_counter_reset()
    lines_conv = sum(1 for k in kinds if k == 1) * 88 * 6 * 162          # from_fp = one product + one reduction
    # the host accessor converts P (and Q) at every fetch; the device converts them once (begin) -- count one conversion per argument
    fetches_p = 88 * n * 2 * 162
    fetches_q = sum(1 for k in kinds if k == 0) * (23 + 1) * 4 * 162        # 21 additions + 2 Frobenius steps + begin
    once = n * 2 * 162 + sum(1 for k in kinds if k == 0) * 4 * 162
    return total - lines_conv - fetches_p - fetches_q + once


rrj = {}
rrj["rr miller_loop_multi, 6 pairs: 3 prepared + 3 walking (an ac17 item with a prepared key): mad instructions"] = rr_mads_multi([1, 0] * 3)
rrj["rr miller_loop_multi, 6 walking pairs: mad instructions"] = rr_mads_multi([0] * 6)
rrj["rr miller_loop_multi, 14 walking pairs: mad instructions"] = rr_mads_multi([0] * 14)
rrj["rr miller_loop_multi, 14 pairs: 7 prepared + 7 walking: mad instructions"] = rr_mads_multi([1, 0] * 7)
rrj["rr miller_loop_multi, 1 walking pair: mad instructions"] = rr_mads_multi([0])
HS.hs_rr_mad_counter_reset()
_o = (ctypes.c_uint32 * 96)()
HS.hs_rr_final_exp(b2c(bytes(m)), _o)
rrj["rr final_exponentiation over workspace slots incl. 12 + 12 conversions: mad instructions (the one inversion runs on the 8 x 32-bit core: + its Fp multiplications x 136)"] = HS.hs_rr_mad_counter_reset()
print(json.dumps(rrj, indent=1))
