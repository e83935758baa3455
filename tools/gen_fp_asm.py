#!/usr/bin/env python3
"""Generates rabe_amd/csrc/bn254/fp_gfx950_gen.h: the straight-line gfx950 forms of the three multiplication routines the
pairing kernels live in -- wide_mul3 (three 256x256 products in lockstep), redc2 (two Montgomery reductions in lockstep) and
mont_mul2_raw over Fp (two Montgomery products in lockstep).

Why generated: hipcc pads every `asm` statement with one wait state before the next VALU instruction that touches its outputs
(cdna_hip_programming.md 5.7 item 2), and a kernel that runs one wave per SIMD pays each of them (tools/ubench_mac.hip: ~1.7
cycles).  The hand-written loops in fp.h issue one statement per product group -- ~120 pads per Fq2 multiplication.  Here
every statement carries as many product groups as the 30-operand limit of an asm statement allows, in a fixed instruction
order: per group the N mads, then the N carry captures (N = chains), so a carry is read two instructions after it is written.

Which products issue WITHOUT a carry capture is the plan of fp.h (ColumnPlan; tests/test_mac_plan.py proves its bounds): the
generator recomputes it with the same rule and emits static_asserts that tie the two together at compile time.

Run from the repository root:  python tools/gen_fp_asm.py
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = (1 << 32) - 1
LIMIT = 0xFFFFFF00


def fp_mod():
    text = open(os.path.join(ROOT, "rabe_amd", "csrc", "bn254", "constants.h")).read()
    m = re.search(r"#define RB_FP_MOD\s*\{([^}]*)\}", text)
    return [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]


def column_plan(mod, with_ab, top_a, top_b):
    """fp.h: ColumnPlan -- the same greedy rule"""
    safe, last_safe = [0] * 16, 0
    for k in range(16):
        used = 0
        if with_ab and 7 <= k <= 14:
            used = top_a if k == 14 else top_a + top_b
        lo, hi = (0, k - 1) if k < 8 else (k - 7, 7)
        taken = set()
        while True:
            best = None
            for i in range(lo, hi + 1):
                if i not in taken and (best is None or mod[k - i] < mod[k - best]):
                    best = i
            if best is None or used + mod[k - best] > LIMIT:
                break
            used += mod[k - best]
            taken.add(best)
            safe[k] |= 1 << best
        n_mp = hi - lo + 1
        n_ab_unsafe = 0 if not with_ab else (k + 1 if k < 7 else (0 if k == 14 else (15 - k) - 2))
        if k < 8 and len(taken) == n_mp and n_ab_unsafe == 0 and used + mod[0] <= LIMIT:
            last_safe |= 1 << k
    return safe, last_safe


class Stmt:
    """one asm statement over N chains: accumulators, carry words, carry SGPR pairs, then inputs"""

    def __init__(self, n, first_capture_pending):
        self.n = n
        self.macs = []                  # (xs[n], ys[n] or single y, y_kind, capture, defines_ovf)
        self.pending = first_capture_pending

    def operands(self, extra=None):
        macs = self.macs + ([extra] if extra else [])
        any_capture = any(m[3] for m in macs)
        fixed = self.n + (self.n if any_capture else 0) + (self.n if any_capture else 1)
        ins = set()
        for xs, ys, kind, cap, _ in macs:
            ins.update(("v", x) for x in xs)
            if kind == "v":
                ins.update(("v", y) for y in ys)
            elif kind == "s":
                ins.add(("s", ys))
        return fixed + len(ins)

    def fits(self, mac):
        return self.operands(mac) <= 30

    def add(self, xs, ys, kind, capture):
        defines = capture and self.pending
        if defines:
            self.pending = False
        self.macs.append((xs, ys, kind, capture, defines))

    def emit(self, out, accs, ovfs, carries):
        n = self.n
        any_capture = any(m[3] for m in self.macs)
        ovf_defined_here = any(m[4] for m in self.macs)
        ops, text = [], []
        idx = {}

        def op(name, constraint):
            if name not in idx:
                idx[name] = len(ops)
                ops.append((name, constraint))
            return "%%%d" % idx[name]

        for a in accs:
            op(a, "+v")
        if any_capture:
            for o in ovfs:
                op(o, "=&v" if ovf_defined_here else "+v")
            for c in carries:
                op(c, "=&s")
        else:
            op(carries[0], "=&s")
        n_out = len(ops)
        lines = []
        for xs, ys, kind, cap, defines in self.macs:
            for ch in range(n):
                y = op(ys[ch], "v") if kind == "v" else (op(ys, "s") if kind == "s" else str(ys))
                c = op(carries[ch] if any_capture else carries[0], "=&s")
                lines.append("v_mad_u64_u32 %s, %s, %s, %s, %s" % (op(accs[ch], "+v"), c, op(xs[ch], "v"), y, op(accs[ch], "+v")))
            if cap:
                for ch in range(n):
                    c = op(carries[ch], "=&s")
                    o = op(ovfs[ch], "")
                    # RB_CPAD (fp.h): empty in the fast build; "s_nop 1" in the RB_SAFE_CARRY build, between the carry-writing
                    # multiply-adds of the group and their first reader (the later readers are further from their writers still)
                    lines.append(("\" RB_CPAD \"" if ch == 0 else "") + "v_addc_co_u32_e64 %s, %s, 0, %s, %s" % (o, c, "0" if defines else o, c))
        outs = ", ".join('"%s"(%s)' % (cst, name) for name, cst in ops[:n_out])
        ins = ", ".join('"%s"(%s)' % (cst, name) for name, cst in ops[n_out:])
        assert len(ops) <= 30, len(ops)
        body = "\\n\\t".join(lines)
        out.append('  asm("%s"\n      : %s\n      : %s);' % (body, outs, ins))


def pack(out, n, macs, accs, ovfs, carries, have_ref):
    """macs: list of (xs, ys, kind, capture); have_ref: [bool] -- whether the column's carry word is already defined"""
    st = Stmt(n, not have_ref[0])
    for xs, ys, kind, cap in macs:
        mac = (xs, ys, kind, cap, False)
        if st.macs and not st.fits(mac):
            st.emit(out, accs, ovfs, carries)
            st = Stmt(n, not have_ref[0])
        st.add(xs, ys, kind, cap)
        if cap:
            have_ref[0] = True
    if st.macs:
        st.emit(out, accs, ovfs, carries)


def shift(out, n, have):
    for ch in range(n):
        if have:
            out.append("  acc%d = (acc%d >> 32) | ((uint64_t)ovf%d << 32);" % (ch, ch, ch))
        else:
            out.append("  acc%d >>= 32;" % ch)


def gen_wide_mul3(out):
    out.append("// three plain 256 x 256 -> 512-bit products in lockstep (operands: a0, b0, a1, b1 < p; a2, b2 < 2p)")
    out.append("template <class A>")
    out.append("RB_HD void wide_mul3(uint32_t* T0, uint32_t* T1, uint32_t* T2, const A& a0, const A& b0, const A& a1, const A& b1, const uint32_t* a2,")
    out.append("                     const uint32_t* b2) {")
    out.append("  uint64_t acc0 = 0, acc1 = 0, acc2 = 0, c0_, c1_, c2_;")
    out.append("  uint32_t ovf0, ovf1, ovf2;")
    for i in range(8):
        out.append("  const uint32_t x0_%d = a0[%d], y0_%d = b0[%d], x1_%d = a1[%d], y1_%d = b1[%d], x2_%d = a2[%d], y2_%d = b2[%d];" % ((i,) * 12))
    accs, ovfs, cars = ["acc0", "acc1", "acc2"], ["ovf0", "ovf1", "ovf2"], ["c0_", "c1_", "c2_"]
    for k in range(15):
        lo, hi = (0, k) if k < 8 else (k - 7, 7)
        tops = [i for i in range(lo, hi + 1) if k >= 7 and (i == 7 or k - i == 7)]
        rest = [i for i in range(lo, hi + 1) if i not in tops]
        macs = [(["x%d_%d" % (c, i) for c in range(3)], ["y%d_%d" % (c, k - i) for c in range(3)], "v", False) for i in tops]
        macs += [(["x%d_%d" % (c, i) for c in range(3)], ["y%d_%d" % (c, k - i) for c in range(3)], "v", True) for i in rest]
        have = [False]
        out.append("  // column %d" % k)
        pack(out, 3, macs, accs, ovfs, cars, have)
        out.append("  T0[%d] = (uint32_t)acc0; T1[%d] = (uint32_t)acc1; T2[%d] = (uint32_t)acc2;" % (k, k, k))
        shift(out, 3, have[0])
    out.append("  T0[15] = (uint32_t)acc0; T1[15] = (uint32_t)acc1; T2[15] = (uint32_t)acc2;")
    out.append("}")


def gen_wide_mac3(out):
    out.append("// the same three products ADDED to the running 512-bit sums T0, T1, T2 (mod 2^512; bn254/coop6.h keeps the true sums below it):")
    out.append("// every column starts with the sum's word (a multiply-add by the constant 1, as in redc2_fp), then the products of wide_mul3")
    out.append("template <class A>")
    out.append("RB_HD void wide_mac3(uint32_t* T0, uint32_t* T1, uint32_t* T2, const A& a0, const A& b0, const A& a1, const A& b1, const uint32_t* a2,")
    out.append("                     const uint32_t* b2) {")
    out.append("  uint64_t acc0 = 0, acc1 = 0, acc2 = 0, c0_, c1_, c2_;")
    out.append("  uint32_t ovf0, ovf1, ovf2;")
    for i in range(8):
        out.append("  const uint32_t x0_%d = a0[%d], y0_%d = b0[%d], x1_%d = a1[%d], y1_%d = b1[%d], x2_%d = a2[%d], y2_%d = b2[%d];" % ((i,) * 12))
    accs, ovfs, cars = ["acc0", "acc1", "acc2"], ["ovf0", "ovf1", "ovf2"], ["c0_", "c1_", "c2_"]
    for k in range(16):
        lo, hi = (0, k) if k < 8 else (k - 7, 7)
        tops = [i for i in range(lo, hi + 1) if k >= 7 and (i == 7 or k - i == 7)]
        rest = [i for i in range(lo, hi + 1) if i not in tops]
        out.append("  // column %d" % k)
        out.append("  const uint32_t t0_%d = T0[%d], t1_%d = T1[%d], t2_%d = T2[%d];" % ((k,) * 6))
        macs = [(["t0_%d" % k, "t1_%d" % k, "t2_%d" % k], 1, "const", False)]
        macs += [(["x%d_%d" % (c, i) for c in range(3)], ["y%d_%d" % (c, k - i) for c in range(3)], "v", False) for i in tops]
        macs += [(["x%d_%d" % (c, i) for c in range(3)], ["y%d_%d" % (c, k - i) for c in range(3)], "v", True) for i in rest]
        have = [False]
        pack(out, 3, macs, accs, ovfs, cars, have)
        out.append("  T0[%d] = (uint32_t)acc0; T1[%d] = (uint32_t)acc1; T2[%d] = (uint32_t)acc2;" % (k, k, k))
        if k < 15:
            shift(out, 3, have[0])
    out.append("}")


def gen_redc2(out, mod, safe, last_safe):
    out.append("// two Montgomery reductions over Fp of 512-bit values (< 2^256 p) in lockstep; results < p")
    out.append("RB_HD void redc2_fp(uint32_t* r0, uint32_t* r1, const uint32_t* W0, const uint32_t* W1) {")
    out.append("  uint64_t acc0 = 0, acc1 = 0, c0_, c1_;")
    out.append("  uint32_t ovf0, ovf1;")
    for j in range(8):
        out.append("  const uint32_t p%d = 0x%08xu;" % (j, mod[j]))
    out.append("  const uint32_t inv_ = FpParams::INV;")
    accs, ovfs, cars = ["acc0", "acc1"], ["ovf0", "ovf1"], ["c0_", "c1_"]
    for k in range(16):
        lo, hi = (0, k - 1) if k < 8 else (k - 7, 7)
        out.append("  // column %d" % k)
        out.append("  const uint32_t w0_%d = W0[%d], w1_%d = W1[%d];" % (k, k, k, k))
        macs = [(["w0_%d" % k, "w1_%d" % k], 1, "const", False)]
        order = sorted([i for i in range(lo, hi + 1) if (safe[k] >> i) & 1], key=lambda i: mod[k - i])
        macs += [(["m0_%d" % i, "m1_%d" % i], "p%d" % (k - i), "s", False) for i in order]
        macs += [(["m0_%d" % i, "m1_%d" % i], "p%d" % (k - i), "s", True) for i in range(lo, hi + 1) if not (safe[k] >> i) & 1]
        have = [False]
        pack(out, 2, macs, accs, ovfs, cars, have)
        if k < 8:
            out.append("  const uint32_t m0_%d = (uint32_t)acc0 * inv_, m1_%d = (uint32_t)acc1 * inv_;" % (k, k))
            closing_safe = (not have[0]) and ((last_safe >> k) & 1)
            pack(out, 2, [(["m0_%d" % k, "m1_%d" % k], "p0", "s", not closing_safe)], accs, ovfs, cars, have)
        else:
            out.append("  r0[%d] = (uint32_t)acc0; r1[%d] = (uint32_t)acc1;" % (k - 8, k - 8))
        shift(out, 2, have[0])
    out.append("  cond_sub_mod<FpParams>(r0, 0);")
    out.append("  cond_sub_mod<FpParams>(r1, 0);")
    out.append("}")


def gen_mul2(out, mod, safe, last_safe):
    out.append("// two Montgomery products over Fp in lockstep; operands < p")
    out.append("template <class A>")
    out.append("RB_HD void mont_mul2_fp(uint32_t* r0, uint32_t* r1, const A& a0, const A& b0, const A& a1, const A& b1) {")
    out.append("  uint64_t acc0 = 0, acc1 = 0, c0_, c1_;")
    out.append("  uint32_t ovf0, ovf1;")
    for j in range(8):
        out.append("  const uint32_t p%d = 0x%08xu;" % (j, mod[j]))
    out.append("  const uint32_t inv_ = FpParams::INV;")
    for i in range(8):
        out.append("  const uint32_t x0_%d = a0[%d], y0_%d = b0[%d], x1_%d = a1[%d], y1_%d = b1[%d];" % ((i,) * 8))
    accs, ovfs, cars = ["acc0", "acc1"], ["ovf0", "ovf1"], ["c0_", "c1_"]
    for k in range(16):
        lo, hi = (0, k) if k < 8 else (k - 7, 7)
        tops = [i for i in range(lo, hi + 1) if k >= 7 and (i == 7 or k - i == 7)]
        mp = [i for i in range(lo, hi + 1) if not (k < 8 and i == k)]
        macs = [(["x0_%d" % i, "x1_%d" % i], ["y0_%d" % (k - i), "y1_%d" % (k - i)], "v", False) for i in tops]
        macs += [(["m0_%d" % i, "m1_%d" % i], "p%d" % (k - i), "s", False) for i in sorted([i for i in mp if (safe[k] >> i) & 1], key=lambda i: mod[k - i])]
        macs += [(["x0_%d" % i, "x1_%d" % i], ["y0_%d" % (k - i), "y1_%d" % (k - i)], "v", True) for i in range(lo, hi + 1) if i not in tops]
        macs += [(["m0_%d" % i, "m1_%d" % i], "p%d" % (k - i), "s", True) for i in mp if not (safe[k] >> i) & 1]
        have = [False]
        out.append("  // column %d" % k)
        pack(out, 2, macs, accs, ovfs, cars, have)
        if k < 8:
            out.append("  const uint32_t m0_%d = (uint32_t)acc0 * inv_, m1_%d = (uint32_t)acc1 * inv_;" % (k, k))
            closing_safe = (not have[0]) and ((last_safe >> k) & 1)
            pack(out, 2, [(["m0_%d" % k, "m1_%d" % k], "p0", "s", not closing_safe)], accs, ovfs, cars, have)
        else:
            out.append("  r0[%d] = (uint32_t)acc0; r1[%d] = (uint32_t)acc1;" % (k - 8, k - 8))
        shift(out, 2, have[0])
    out.append("  cond_sub_mod<FpParams>(r0, 0);")
    out.append("  cond_sub_mod<FpParams>(r1, 0);")
    out.append("}")


def main():
    mod = fp_mod()
    top = mod[7] + 1
    redc_safe, redc_last = column_plan(mod, False, 0, 0)
    mul_safe, mul_last = column_plan(mod, True, top, top)
    out = ["// GENERATED by tools/gen_fp_asm.py -- do not edit; regenerate with `python tools/gen_fp_asm.py`.",
           "// Straight-line gfx950 forms of wide_mul3 / redc2 / mont_mul2_raw over Fp (see the generator's header for the why).",
           "// Included by fp.h inside its device-only block.", ""]
    out.append("// the plan these statements were laid out for is the one ColumnPlan computes (fp.h; tests/test_mac_plan.py proves its bounds)")
    for name, safe, last, args in (("redc", redc_safe, redc_last, "false, 0, 0"), ("mul", mul_safe, mul_last, "true, FpParams::mod(7) + 1, FpParams::mod(7) + 1")):
        out.append("namespace gen_check_%s {" % name)
        out.append("constexpr ColumnPlan<FpParams> plan(%s);" % args)
        out.append("static_assert(" + " && ".join("plan.safe[%d] == 0x%x" % (k, safe[k]) for k in range(16)) + ", \"regenerate fp_gfx950_gen.h\");")
        out.append("static_assert(plan.last_safe == 0x%x, \"regenerate fp_gfx950_gen.h\");" % last)
        out.append("}")
    out.append("")
    gen_wide_mul3(out)
    out.append("")
    gen_wide_mac3(out)
    out.append("")
    gen_redc2(out, mod, redc_safe, redc_last)
    out.append("")
    gen_mul2(out, mod, mul_safe, mul_last)
    out.append("")
    path = os.path.join(ROOT, "rabe_amd", "csrc", "bn254", "fp_gfx950_gen.h")
    with open(path, "w") as f:
        f.write("\n".join(out))
    n_stmt = sum(1 for l in out if l.startswith("  asm("))
    print("wrote", path, len(out), "lines,", n_stmt, "asm statements")


if __name__ == "__main__":
    main()
