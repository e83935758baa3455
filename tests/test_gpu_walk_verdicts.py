"""G2 membership as a by-product of a decrypt's Miller loops (include/rabe_hip.h: rhip_ctx_collect_walk_verdicts; engine_jobs.hip:
k_walk_verdicts; tests/test_walk_relation.py has the argument): after k_miller_multi the running point of a walking pair holds
[6u+2]Q + psi(Q) - psi^2(Q), which is -psi^3(Q) exactly for the members of G2.

Engine level: the verdicts and counts of pair lists whose G2 arguments are members, twist points outside G2, members shifted by a
cofactor-torsion point and the point at infinity, on the planned (ragged) and the static (uniform) launch geometry -- against
rhip_g2_in_subgroup_by_order (the definition r * Q = O).
Host level: a packed decrypt of records in which ONE group element was replaced by a point of the twist outside G2 (on the curve,
canonical coordinates: only a subgroup test can see it) fails that item with the decoder's error and leaves its neighbours alone --
for an element the decrypt walks, for a leaf the policy did not select (stand-alone test), and the same with the fused checks switched
off (RABE_NO_WALK_CHECKS)."""
import ctypes
import os
import random
import struct
import subprocess
import sys

import numpy as np
import pytest

from rabe_amd import hostlib as hl
from tests.test_gpu_validation import fp2_sqrt

pytestmark = pytest.mark.gpu
PT = b"dance like no one's watching, encrypt like everyone is!"


def _points(rnd, n_members=6, n_twist=4):
    from oracle import bn254 as bn
    b2 = bn.fp2_mul((3, 0), bn.fp2_inv(bn.XI))
    twist = []
    while len(twist) < n_twist:
        x = (rnd.randrange(bn.P), rnd.randrange(bn.P))
        y = fp2_sqrt(bn.fp2_add(bn.fp2_mul(bn.fp2_mul(x, x), x), b2))
        if y is not None:
            twist.append((x, y))
    members = [bn.g2_mul(bn.G2_GEN, rnd.randrange(1, bn.R)) for _ in range(n_members)]
    cof = [bn.g2_add(bn.ec_mul(bn.FP2, t, bn.R - 1), t) for t in twist[:2]]
    shifted = [bn.g2_add(m, c) for m, c in zip(members, cof)]
    return members, twist, cof, shifted


@pytest.mark.parametrize("shape", ["ragged", "uniform", "ragged_static"])
def test_walk_verdicts_of_pair_lists(shape):
    from rabe_amd import Engine
    from rabe_amd.engine import _sz
    from oracle import bn254 as bn
    rnd = random.Random(31)
    members, twist, cof, shifted = _points(rnd)
    good = [bn.g2_to_le(q) for q in members]
    bad = [bn.g2_to_le(q) for q in twist + cof + shifted]
    inf_q = bn.g2_to_le(None)
    ps = [bn.g1_to_le(bn.g1_mul(bn.G1_GEN, rnd.randrange(1, bn.R))) for _ in range(5)]
    inf_p = bn.g1_to_le(None)
    n_items = 40
    counts = [7] * n_items if shape == "uniform" else [rnd.choice([1, 2, 5, 9, 70]) for _ in range(n_items)]
    off, P, Q, want_fail, want_count = [0], [], [], [], []
    for i, c in enumerate(counts):
        fail, cnt = 0, 0
        for j in range(c):
            kind = rnd.random()
            p = ps[rnd.randrange(5)]
            if i % 3 == 0 and kind < 0.15:
                q, is_bad = bad[rnd.randrange(len(bad))], True
            else:
                q, is_bad = good[rnd.randrange(len(good))], False
            if kind > 0.95:
                q, is_bad = inf_q, False                 # an argument at infinity: the pair is skipped, not counted
            if 0.90 < kind <= 0.95:
                p = inf_p
            skipped = p == inf_p or q == inf_q
            if not skipped:
                cnt += 1
                fail |= int(is_bad)
            P.append(p)
            Q.append(q)
        off.append(off[-1] + c)
        want_fail.append(fail)
        want_count.append(cnt)
    assert any(want_fail) and not all(want_fail)
    if shape == "ragged_static":
        os.environ["RABE_NO_MILLER_PLAN"] = "1"
    eng = Engine(0)
    try:
        v = eng.alloc(8 * n_items)
        eng._check(eng.lib.rhip_memset_async(eng.ctx, v.ptr, 0, _sz(8 * n_items)))
        eng._check(eng.lib.rhip_ctx_collect_walk_verdicts(eng.ctx, ctypes.c_void_p(v.ptr.value), ctypes.c_void_p(v.ptr.value + 4 * n_items)))
        eng.pairing_jobs(off, P, Q)
        got = struct.unpack("<%dI" % (2 * n_items), eng.download(v))
        assert list(got[:n_items]) == want_fail
        assert list(got[n_items:]) == want_count
        # one-shot: the next launch collects nothing
        eng._check(eng.lib.rhip_memset_async(eng.ctx, v.ptr, 0, _sz(8 * n_items)))
        eng.pairing_jobs(off, P, Q)
        assert not any(struct.unpack("<%dI" % (2 * n_items), eng.download(v)))
        # the listed form of the stand-alone test agrees with the definition
        allq = good + bad + [inf_q]
        d = eng.upload(b"".join(allq))
        idx = [len(allq) - 1 - k for k in range(len(allq))]
        o1, o2 = eng.alloc(4 * len(allq)), eng.alloc(4 * len(allq))
        eng._check(eng.lib.rhip_g2_in_subgroup_by_order(eng.ctx, _sz(len(allq)), d.ptr, o1.ptr))
        eng._check(eng.lib.rhip_g2_in_subgroup_at(eng.ctx, _sz(len(allq)), eng.upload_u32(idx).ptr, d.ptr, o2.ptr))
        by_order = struct.unpack("<%dI" % len(allq), eng.download(o1))
        assert list(struct.unpack("<%dI" % len(allq), eng.download(o2))) == [by_order[k] for k in idx]
        assert list(by_order) == [1] * len(good) + [0] * len(bad) + [1]
    finally:
        os.environ.pop("RABE_NO_MILLER_PLAN", None)
        eng.close()


def _offsets(items):
    return np.concatenate([[0], np.cumsum([len(p) for p in items])]).astype(np.uint64)


def _split(out, off, n):
    return [bytes(out[int(off[i]):int(off[i + 1])]) for i in range(n)]


def _bsw_leaf_g2_offset(rec, leaf):
    n = int.from_bytes(rec[:4], "little")
    at = 4 + n + 1 + 64 + 384
    leaves = int.from_bytes(rec[at:at + 4], "little")
    assert leaf < leaves
    at += 4
    for y in range(leaves):
        ln = int.from_bytes(rec[at:at + 4], "little")
        at += 4 + ln + 64
        if y == leaf:
            return at
        at += 128
    raise AssertionError


CHILD = r"""
import sys
sys.path.insert(0, %r)
from tests.test_gpu_walk_verdicts import run_host_cases
run_host_cases()
print("host cases ok")
"""


def run_host_cases():
    from oracle import bn254 as bn
    from rabe_amd.schemes import ac17, bsw
    rnd = random.Random(77)
    members, twist, cof, shifted = _points(rnd, 2, 2)
    outside = [bn.g2_to_le(twist[0]), bn.g2_to_le(shifted[0])]
    host = hl.Host(0)
    n = 10
    pts = [PT + bytes([i]) for i in range(n)]

    def expect(fn, sk, recs, victim, msg_part, want=None):
        want = want or pts
        b = b"".join(recs)
        out, oo, st = fn(host, sk, b, _offsets(recs))
        got = _split(out, oo, n)
        for i in range(n):
            if i == victim:
                assert st[i] == -1, (i, st)
            else:
                assert st[i] == 0 and got[i] == want[i], (i, st)
        err = (host.lib.rabe_host_last_error(host.h) or b"").decode()
        assert msg_part in err, err

    # ---- AC17: c_0[t] is walked by every decrypt
    pk, msk = ac17.setup(host)
    sk = ac17.cp_keygen(host, msk, ["A", "B", "C"])
    pols = ['"A" and "B"', '"A" or ("B" and "C")']
    blob, off = ac17.cp_encrypt_packed(host, pk, pols, [i % 2 for i in range(n)], b"".join(pts), _offsets(pts), hl.HUMAN_POLICY)
    recs = _split(blob, off, n)
    out, oo, st = ac17.cp_decrypt_packed(host, sk, blob, off)
    assert not st.any() and _split(out, oo, n) == pts
    for victim, t, q in ((3, 0, outside[0]), (6, 2, outside[1]), (0, 1, outside[0])):
        r = bytearray(recs[victim])
        at = 4 + int.from_bytes(r[:4], "little") + 1 + 4 + 128 * t
        r[at:at + 128] = q
        expect(ac17.cp_decrypt_packed, sk, recs[:victim] + [bytes(r)] + recs[victim + 1:], victim, "not a member of G2")
    # ---- BSW: a selected leaf (walked) and a leaf the policy did not select (stand-alone test of the complement)
    bpk, bmsk = bsw.setup(host)
    bsk = bsw.keygen(host, bpk, bmsk, ["A", "B", "C"])          # no "D": the second policy is satisfied through "A" alone
    bpols = ['"A" and "B" and "C"', '"A" or ("B" and "D")']
    blob, off = bsw.encrypt_packed(host, bpk, bpols, [i % 2 for i in range(n)], b"".join(pts), _offsets(pts), hl.HUMAN_POLICY)
    recs = _split(blob, off, n)
    out, oo, st = bsw.decrypt_packed(host, bsk, blob, off)
    assert not st.any() and _split(out, oo, n) == pts
    for victim, leaf, q in ((2, 1, outside[0]), (4, 2, outside[1]), (5, 0, outside[0]), (5, 2, outside[1]), (7, 1, outside[0])):
        r = bytearray(recs[victim])
        at = _bsw_leaf_g2_offset(r, leaf)
        r[at:at + 128] = q
        expect(bsw.decrypt_packed, bsk, recs[:victim] + [bytes(r)] + recs[victim + 1:], victim, "not a group member")
    # an all-AND batch (every leaf walked: no complement) with one damaged record
    blob, off = bsw.encrypt_packed(host, bpk, bpols[:1], [0] * n, b"".join(pts), _offsets(pts), hl.HUMAN_POLICY)
    recs = _split(blob, off, n)
    r = bytearray(recs[8])
    at = _bsw_leaf_g2_offset(r, 2)
    r[at:at + 128] = outside[1]
    expect(bsw.decrypt_packed, bsk, recs[:8] + [bytes(r)] + recs[9:], 8, "not a group member")
    # a selected leaf at infinity: the pair is skipped, the walk counts one argument less, the item is re-examined stand-alone --
    # infinity IS a member, so the decoder accepts the element and the item fails later, at the AES layer
    r = bytearray(recs[1])
    at = _bsw_leaf_g2_offset(r, 0)
    r[at:at + 128] = bn.g2_to_le(None)
    b = b"".join(recs[:1] + [bytes(r)] + recs[2:])
    out, oo, st = bsw.decrypt_packed(host, bsk, b, _offsets(recs))
    assert st[1] == -1 and all(st[i] == 0 for i in range(n) if i != 1)
    # ---- LSW (n keys, one ciphertext): D2 of a selected key leaf is walked, D2 of a leaf the selection left out is not
    from rabe_amd.schemes import aw11, lsw
    lpk, lmsk = lsw.setup(host)
    lpols = ['{"name": "and", "children": [{"name": "A"}, {"name": "B"}]}', '{"name": "or", "children": [{"name": "A"}, {"name": "C"}]}']
    blob, off = lsw.keygen_packed(host, lpk, lmsk, lpols, [i % 2 for i in range(n)], hl.JSON_POLICY)
    recs = _split(blob, off, n)
    ct = lsw.encrypt(host, lpk, ["A", "B"], PT)
    out, oo, st = lsw.decrypt_packed(host, ct, blob, off)
    assert not st.any() and _split(out, oo, n) == [PT] * n
    same = [PT] * n

    def key_leaf_d2(rec, leaf):
        ln = int.from_bytes(rec[:4], "little")
        at = 4 + ln + 1
        leaves = int.from_bytes(rec[at:at + 4], "little")
        assert leaf < leaves
        at += 4
        for y in range(leaves):
            nl = int.from_bytes(rec[at:at + 4], "little")
            at += 4 + nl + 64
            if y == leaf:
                return at
            at += 128 + 192
        raise AssertionError
    for victim, leaf, q in ((2, 0, outside[0]), (4, 1, outside[1]), (3, 0, outside[1]), (5, 1, outside[0])):
        r = bytearray(recs[victim])
        at = key_leaf_d2(r, leaf)
        r[at:at + 128] = q
        expect(lambda h, c, b, o: lsw.decrypt_packed(h, c, b, o), ct, recs[:victim] + [bytes(r)] + recs[victim + 1:], victim, "not a group member", same)
    # ---- AW11: C2 of a selected row is walked; C3 enters its pairing as a sum and keeps the stand-alone test
    gk = aw11.setup(host)
    pk1, msk1 = aw11.authgen(host, gk, ["A", "B"])
    pk2, msk2 = aw11.authgen(host, gk, ["C"])
    apols = ['{"name": "and", "children": [{"name": "A"}, {"name": "C"}]}', '{"name": "or", "children": [{"name": "B"}, {"name": "C"}]}']
    blob, off = aw11.encrypt_packed(host, gk, [pk1, pk2], apols, [i % 2 for i in range(n)], b"".join(pts), _offsets(pts), hl.JSON_POLICY)
    recs = _split(blob, off, n)
    ask = aw11.keygen(host, gk, msk1, "alice", ["A", "B"])
    aw11.add_to_attribute(host, gk, msk2, "C", ask)
    out, oo, st = aw11.decrypt_packed(host, gk, ask, blob, off)
    assert not st.any() and _split(out, oo, n) == pts

    def row_elem(rec, row, which):          # which: 2 = C2, 3 = C3
        ln = int.from_bytes(rec[:4], "little")
        at = 4 + ln + 1 + 384
        rows = int.from_bytes(rec[at:at + 4], "little")
        assert row < rows
        at += 4
        for y in range(rows):
            nl = int.from_bytes(rec[at:at + 4], "little")
            at += 4 + nl + 384
            if y == row:
                return at + (128 if which == 3 else 0)
            at += 256
        raise AssertionError
    for victim, row, which, q in ((2, 0, 2, outside[0]), (4, 1, 3, outside[1]), (3, 0, 2, outside[1]), (3, 1, 2, outside[0]), (5, 1, 3, outside[0])):
        r = bytearray(recs[victim])
        at = row_elem(r, row, which)
        r[at:at + 128] = q
        expect(lambda h, k, b, o: aw11.decrypt_packed(h, gk, k, b, o), ask, recs[:victim] + [bytes(r)] + recs[victim + 1:], victim, "not a group member")
    host.close()


@pytest.mark.parametrize("fused", [True, False], ids=["walk_checks", "stand_alone"])
def test_a_point_outside_g2_fails_its_item(fused):
    env = dict(os.environ)
    if not fused:
        env["RABE_NO_WALK_CHECKS"] = "1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pr = subprocess.run([sys.executable, "-c", CHILD % root], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert pr.returncode == 0 and b"host cases ok" in pr.stdout, pr.stdout.decode()[-3000:]
