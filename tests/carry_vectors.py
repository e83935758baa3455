"""Helper of tests/test_gpu_carry_interlock.py: runs a fixed set of engine operations -- chosen so that every multiply-accumulate
and carry-chain form of bn254/fp.h executes on adversarial limb patterns and on whole pairings -- through the library named by
RABE_HIP_LIB and prints one JSON object {group: sha256 of the output bytes}.  Run once per build; the digests must agree."""
import hashlib
import json
import random
import sys


def main():
    from oracle import bn254 as bn
    from rabe_amd import Engine
    from tests.test_gpu_field_edge import RINV_P, RINV_R, canon, fp12_of, mont_patterns
    rnd = random.Random(0xCA221)
    le = lambda x: int(x).to_bytes(32, "little")
    eng = Engine(0)
    out = {}

    def put(name, chunks):
        out[name] = hashlib.sha256(b"".join(chunks)).hexdigest()

    # Fr: single-chain products, additive chains, inversion
    ms = mont_patterns(bn.R, 150)
    xs = [canon(m, RINV_R, bn.R) for m in ms]
    a = [x for x in xs for _ in range(3)]
    b = [rnd.choice(xs) for _ in a]
    A, B = [le(x) for x in a], [le(x) for x in b]
    for op, nm in ((0, "fr_add"), (1, "fr_sub"), (2, "fr_mul")):
        put(nm, eng.fr_op(op, A, B))
    put("fr_neg", eng.fr_op(3, A))
    put("fr_inv", eng.fr_op(4, A[:64]))
    # Fq12: two- and three-chain products, lazy Fq2 products with their two-chain reduction, xi-reductions, Fp inversion
    mp = mont_patterns(bn.P, 160)
    elems = [fp12_of(mp[i:i + 12]) for i in range(0, len(mp) - 12, 4)] + [fp12_of([m] * 12) for m in mp[:16]]
    elems = [e for e in elems if e != bn.FP12_ZERO]
    G = [bn.gt_to_le(e) for e in elems]
    H = [G[(7 * i + 3) % len(G)] for i in range(len(G))]
    put("gt_mul", eng.gt_mul(G, H))
    put("gt_inv", eng.gt_inv(G[:48]))
    # curve arithmetic (G1: paired products; G2: Fq2 products) and whole pairings (Miller loop + final exponentiation)
    ks = [rnd.randrange(1, bn.R) for _ in range(24)] + [1, 2, bn.R - 1, (1 << 253) + 1]
    P0 = bn.g1_to_le(bn.g1_mul(bn.G1_GEN, 0x1234567))
    Q0 = bn.g2_to_le(bn.g2_mul(bn.G2_GEN, 0x7654321))
    g1 = eng.g1_mul([P0] * len(ks), [le(k) for k in ks])
    g2 = eng.g2_mul([Q0] * len(ks), [le(k) for k in ks])
    put("g1_mul", g1)
    put("g2_mul", g2)
    put("g1_add", eng.g1_add(g1, g1[1:] + g1[:1]))
    put("g2_add", eng.g2_add(g2, g2[1:] + g2[:1]))
    e = eng.pairing(g1[:12], g2[:12])
    put("pairing", e)
    put("gt_pow", eng.gt_pow(e[:8], [le(k) for k in ks[:8]]))
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
