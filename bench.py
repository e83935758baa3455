#!/usr/bin/env python3
"""bench.py -- AC17 CP-ABE encrypt+decrypt throughput on MI355X (BASELINE.json metric, config 2).

One step = one pass of the hot path over one batch: `--batch` (4096) independent
ac17::cp_encrypt + ac17::cp_decrypt group-arithmetic calls at `--attrs` (50) attributes, 16 distinct
random binary AND/OR policies x 256 items, one public key, one secret key holding all attributes
(SURVEY.md 8d config 2).  Inputs (randomness s0,s1 and the Gt message per item, policy tables, key
material, selection lists) are resident in HBM before the timed region; ciphertexts go
encrypt -> HBM -> decrypt without touching the host.  N>1: the batch definition is per rank
(weak scaling), ranks are independent (no data-path collective); the only collective is the
max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (integer-VALU bound: the
path is modular big-integer arithmetic, neither HBM- nor MFMA-bound -- DESIGN.md section 5);
`cpu_baseline` times the oracle's reference-order restatement on the host CPU (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# One hardware queue per in-flight batch (HIP's default is 4); must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

G1_GEN = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
G2_GEN = b"".join(int(v).to_bytes(32, "little") for v in (
    10857046999023057135944570762232829481370756359578518086990519993285655852781,
    11559732032986387107991004021392285783925812861821192530917403151452391805634,
    8495653923123431417604973247489272438418190587263600148770280649306958101930,
    4082367875863433681332203403145435568316851327593401208105741076214120093531))

# Algorithmic work, in Fp multiplications (1 Fp mul = 136 32x32 multiply-adds: 8-limb CIOS), per lane:
#   SURVEY.md 8d constants (the "algorithmic minimum" the roofline is priced against) and the
#   instrumented counts of this engine's own code (tests/count_muls.py).
MAC_PER_FPMUL = 136
SURVEY_MILLER_FPMUL = 8000          # "Miller loop (optimal ate, 65-bit loop) ~ 8 kM"
SURVEY_MIXED_ADD_FPMUL = 11
IMPL_MILLER_FPMUL = 8983            # tests/count_muls.py: miller_loop (NAF chain) with Jacobian P
IMPL_MILLER2_FPMUL = 12170          # tests/count_muls.py: miller_loop_pair_parked (A replays prepared lines, B Jacobian, merged lines), two pairings
IMPL_FINAL_EXP_FPMUL = 7553         # tests/count_muls.py: final_exponentiation_ws (width-3 NAF exponent chain)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=800)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="items per step per GPU")
    ap.add_argument("--attrs", type=int, default=50)
    ap.add_argument("--policies", type=int, default=16)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--inflight", type=int, default=20,
                    help="independent steps (batches) in flight on separate HIP streams; 1 = strictly one batch at a time")
    ap.add_argument("--pairing-mode", type=int, default=0, choices=[0, 1, 3],
                    help="0 auto, 1 one lane per pairing, 3 three cooperating lanes per pairing (identical results)")
    ap.add_argument("--g-window", type=int, default=26,
                    help="window width (bits) of the fixed-base table of g: 16 (67 MB), or 17..27 signed digits (24: 5.4 GB, 26: 19 GB)")
    ap.add_argument("--no-host-io-leg", action="store_true",
                    help="skip the informational second timed region in which every step also moves its inputs and outputs over PCIe")
    ap.add_argument("--only-encrypt", action="store_true", help="diagnostic: skip the decrypt half of every step (value is then not the metric)")
    ap.add_argument("--no-prepared-sk", action="store_true",
                    help="decrypt without the per-key prepared lines (6 independent Miller loops per item)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="items for the CPU baseline (0 = auto)")
    return ap.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl (= RCCL) on the GPUs; RABE_DIST_BACKEND=gloo lets several ranks share one GPU for a functional check
        dist.init_process_group(backend=os.environ.get("RABE_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU fallback)"
    torch.cuda.set_device(local_rank)

    from rabe_amd import Engine
    from rabe_amd import engine as E
    from rabe_amd import hostprep as hp

    eng = Engine(local_rank)
    n_cu, dev_name = eng.device_info()
    stream = torch.cuda.Stream(device=local_rank)
    eng.set_stream(stream.cuda_stream)
    eng.set_pairing_mode(args.pairing_mode)

    rnd = random.Random(args.seed * 1000003 + rank)
    R = hp.R_ORDER
    le = hp.fr_le

    def rfr():
        return rnd.randrange(1, R)

    # ---------------------------------------------------------------- key material (ac17::setup, :141-182) via Level E ops
    a = [rfr(), rfr()]
    b = [rfr(), rfr()]
    k = [rfr(), rfr(), rfr()]
    g = eng.g1_mul([G1_GEN], [le(rfr())])[0]
    h = eng.g2_mul([G2_GEN], [le(rfr())])[0]
    h_a = eng.g2_mul([h, h], [le(a[0]), le(a[1])]) + [h]
    g_k = eng.g1_mul([g] * 3, [le(x) for x in k])
    e_gh = eng.pairing([g], [h])[0]
    e_gh_ka = eng.gt_pow([e_gh] * 2, [le(k[i] * a[i] + k[2]) for i in range(2)])
    pk = E.Ac17Pk(eng, g, h_a, e_gh_ka)
    if args.g_window > 16:
        pk.set_g_window(args.g_window)

    # ---------------------------------------------------------------- secret key with all attributes (ac17::cp_keygen, :191-264)
    attrs = ["a%d" % (i + 1) for i in range(args.attrs)]
    g_tab, h_tab = eng.g1_table(g), eng.g2_table(h)
    H, H01 = hp.ac17_keygen_tables(attrs)
    dk0, dk, dkp = eng.alloc(3 * 128), eng.alloc(len(attrs) * 3 * 64), eng.alloc(3 * 64)
    E.ac17_keygen_dev(eng, g_tab, h_tab, eng.upload(b"".join(g_k)), eng.upload(b"".join(le(pow(x, R - 2, R)) for x in a)),
                      eng.upload(b"".join(le(x) for x in b)), 1, len(attrs), eng.upload(H), eng.upload(H01),
                      eng.upload(le(rfr()) + le(rfr())), eng.upload(b"".join(le(rfr()) for _ in attrs)), eng.upload(le(rfr())),
                      dk0, dk, dkp)

    # ---------------------------------------------------------------- policies (host: parse/MSP/prune are string work)
    prnd = random.Random(args.seed)          # the same policies on every rank
    trees = [hp.random_binary_tree(attrs, prnd) for _ in range(args.policies)]
    tables, sels, pol_rows, nnz = [], [], [], []
    for t in trees:
        pi, A, z = hp.ac17_policy_table(t)
        ok, ct_sel, sk_sel = hp.ac17_decrypt_selection(attrs, pi, t)
        assert ok
        tables.append(A)
        sels.append((ct_sel, sk_sel))
        pol_rows.append(len(pi))
        nnz.append(z)
    A_off = [0]
    for r_ in pol_rows:
        A_off.append(A_off[-1] + r_)
    dA = eng.upload(b"".join(tables))

    B = args.batch
    item_pol = [i % args.policies for i in range(B)]
    ct_row_off = [0]
    ct_sel_all, sk_sel_all, ct_sel_off, sk_sel_off = [], [], [0], [0]
    for i in range(B):
        p_ = item_pol[i]
        ct_row_off.append(ct_row_off[-1] + pol_rows[p_])
        ct_sel_all += sels[p_][0]
        sk_sel_all += sels[p_][1]
        ct_sel_off.append(len(ct_sel_all))
        sk_sel_off.append(len(sk_sel_all))
    total_rows = ct_row_off[-1]
    d_item_A_off = eng.upload_u32([A_off[p_] for p_ in item_pol])
    d_ct_row_off = eng.upload_u32(ct_row_off)
    d_ct_sel, d_ct_sel_off = eng.upload_u32(ct_sel_all), eng.upload_u32(ct_sel_off)
    d_sk_sel, d_sk_sel_off = eng.upload_u32(sk_sel_all), eng.upload_u32(sk_sel_off)
    d_sk_idx = eng.upload_u32([0] * B)
    d_sk_row_off = eng.upload_u32([0, len(attrs)])

    # per-item randomness: s0, s1 and the Gt message msg = e_gh^rho (computed on the GPU, untimed)
    ds = eng.upload(b"".join(le(rfr()) for _ in range(2 * B)))
    e_tab = eng.gt_table(e_gh)
    dmsg = eng.alloc(B * 384)
    drho = eng.upload(b"".join(le(rfr()) for _ in range(B)))
    eng._check(eng.lib.rhip_gt_table_pow(eng.ctx, e_tab.h, E._sz(B), drho.ptr, dmsg.ptr))

    # Steps are independent batches: up to `inflight` of them are pipelined on separate HIP streams (each
    # lane = its own engine context, stream, scratch and output buffers; inputs and key tables are shared,
    # read-only).  The kernels of one batch are latency-bound at 4096 items (384 Miller waves on 1024
    # SIMDs), so overlapping batches is what fills the chip.
    S = max(1, min(args.inflight, args.steps))
    lanes_ctx = [eng]
    streams = [stream]
    for _ in range(S - 1):
        e2 = Engine(local_rank)
        st2 = torch.cuda.Stream(device=local_rank)
        e2.set_stream(st2.cuda_stream)
        e2.set_pairing_mode(args.pairing_mode)
        lanes_ctx.append(e2)
        streams.append(st2)
    bufs = [(e_.alloc(B * 3 * 128), e_.alloc(total_rows * 3 * 64), e_.alloc(B * 384), e_.alloc(B * 384)) for e_ in lanes_ctx]
    dc0, dc, dcp, dout = bufs[0]
    step_no = [0]

    # the key is loaded once: Miller-loop lines of k_0 (rhip_ac17_sk_prepare), reused by every decryption with it
    sk_lines = None if args.no_prepared_sk else E.Ac17SkLines(eng, 1, dk0)
    eng.sync()

    def step():
        i = step_no[0] % S
        step_no[0] += 1
        e_ = lanes_ctx[i]
        c0_, c_, cp_, out_ = bufs[i]
        E.ac17_encrypt_dev(e_, pk, B, dA, d_item_A_off, d_ct_row_off, total_rows, ds, dmsg, c0_, c_, cp_)
        if args.only_encrypt:
            return
        if sk_lines is None:
            E.ac17_decrypt_dev(e_, B, c0_, c_, d_ct_row_off, cp_, dk0, dk, d_sk_row_off, dkp, d_sk_idx,
                               d_ct_sel, d_ct_sel_off, d_sk_sel, d_sk_sel_off, out_)
        else:
            E.ac17_decrypt_prepared_dev(e_, B, c0_, c_, d_ct_row_off, cp_, sk_lines, dk, d_sk_row_off, dkp, d_sk_idx,
                                        d_ct_sel, d_ct_sel_off, d_sk_sel, d_sk_sel_off, out_)

    def sync_all():
        for e_ in lanes_ctx:
            e_.sync()

    def barrier():
        if world > 1:
            dist.barrier()

    # ---------------------------------------------------------------- timed region
    for _ in range(max(args.warmup, 1) * S if args.warmup else 0):
        step()
    sync_all()
    torch.cuda.synchronize()
    barrier()
    step_no[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    from rabe_amd import shard
    elapsed = shard.max_over_ranks(t1 - t0)

    # ---------------------------------------------------------------- size-independent correctness property on the FULL batch:
    # decrypt(encrypt(msg)) == msg, bit for bit, for every item (oracle parity at small sizes is in tests/)
    want = eng.download(dmsg)
    ok = all(lanes_ctx[i].download(bufs[i][3]) == want for i in range(S))
    if world > 1:
        f = torch.tensor([1 if ok else 0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        ok = bool(f.item())

    value = world * B * args.steps / elapsed
    result = {
        "metric": "ABE ops/sec (AC17 CP-ABE encrypt+decrypt)", "value": round(value, 2), "unit": "ops/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (BN254 Fp/Fr Montgomery, 8x32)",
        "data": "synthetic", "roundtrip_bit_exact": ok,
        "config": {"workload": "AC17 CP-ABE, %d-attribute random binary AND/OR MSP policies (%d distinct), batch %d encrypt+decrypt per GPU"
                               % (args.attrs, args.policies, B),
                   "batch_per_gpu": B, "attrs": args.attrs, "policies": args.policies, "rows": total_rows // B,
                   "pruned_leaves_avg": round(len(ct_sel_all) / B, 2), "msp_nnz_avg": round(sum(nnz) / len(nnz), 1),
                   "steps_in_flight": S, "pairing_mode": args.pairing_mode,
                   "parallelism": "batch-sharded x%d (no data-path collective)" % world, "device": dev_name},
    }

    # ---------------------------------------------------------------- informational: the same steps with host buffers
    # (inputs s, msg uploaded; ciphertext c_0, c, c_p and the decrypted Gt downloaded; pinned memory, copies ordered on
    # each batch's own stream so that they overlap the other batches' kernels).  Never `value`.
    if not args.no_host_io_leg and not args.only_encrypt:
        n_s, n_msg = 2 * B * 32, B * 384
        sizes_out = (B * 3 * 128, total_rows * 3 * 64, B * 384, B * 384)
        io = []
        for e_ in lanes_ctx:
            h_in = (e_.host_alloc(n_s), e_.host_alloc(n_msg))
            h_out = tuple(e_.host_alloc(z) for z in sizes_out)
            io.append((e_.alloc(n_s), e_.alloc(n_msg), h_in, h_out))
        hs, hm = eng.download(ds), eng.download(dmsg)
        for (_, _, h_in, _) in io:
            ctypes.memmove(h_in[0], hs, n_s)
            ctypes.memmove(h_in[1], hm, n_msg)

        # one copy context per batch in flight: it drains the outputs while the compute context runs on
        copy_ctx = []
        for _ in lanes_ctx:
            c2 = Engine(local_rank)
            st3 = torch.cuda.Stream(device=local_rank)
            c2.set_stream(st3.cuda_stream)
            copy_ctx.append((c2, st3))

        def step_io():
            i = step_no[0] % S
            step_no[0] += 1
            e_ = lanes_ctx[i]
            cpy = copy_ctx[i][0]
            c0_, c_, cp_, out_ = bufs[i]
            ds_, dmsg_, h_in, h_out = io[i]
            e_.wait_for(cpy)                               # the previous round's downloads of these buffers are done
            e_.upload_async(ds_, h_in[0], n_s)
            e_.upload_async(dmsg_, h_in[1], n_msg)
            E.ac17_encrypt_dev(e_, pk, B, dA, d_item_A_off, d_ct_row_off, total_rows, ds_, dmsg_, c0_, c_, cp_)
            cpy.wait_for(e_)
            cpy.download_async(h_out[0], c0_, sizes_out[0])
            cpy.download_async(h_out[1], c_, sizes_out[1])
            cpy.download_async(h_out[2], cp_, sizes_out[2])
            if sk_lines is None:
                E.ac17_decrypt_dev(e_, B, c0_, c_, d_ct_row_off, cp_, dk0, dk, d_sk_row_off, dkp, d_sk_idx,
                                   d_ct_sel, d_ct_sel_off, d_sk_sel, d_sk_sel_off, out_)
            else:
                E.ac17_decrypt_prepared_dev(e_, B, c0_, c_, d_ct_row_off, cp_, sk_lines, dk, d_sk_row_off, dkp, d_sk_idx,
                                            d_ct_sel, d_ct_sel_off, d_sk_sel, d_sk_sel_off, out_)
            cpy.wait_for(e_)
            cpy.download_async(h_out[3], out_, sizes_out[3])

        def sync_io():
            sync_all()
            for c2, _ in copy_ctx:
                c2.sync()

        step_no[0] = 0
        for _ in range(S):
            step_io()
        sync_io()
        barrier()
        step_no[0] = 0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_io()
        sync_io()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        el_io = shard.max_over_ranks(t1 - t0)
        ok_io = all(ctypes.string_at(io[i][3][3], sizes_out[3]) == want for i in range(S))
        per_step = n_s + n_msg + sum(sizes_out)
        result["host_io_leg"] = {"ops_per_s": round(world * B * args.steps / el_io, 2), "ms_per_step": round(1e3 * el_io / args.steps, 3),
                                 "pcie_bytes_per_step": per_step, "pcie_GBps": round(per_step * args.steps / el_io / 1e9, 2),
                                 "roundtrip_bit_exact": ok_io,
                                 "note": "inputs uploaded and all outputs downloaded every step (pinned host memory; downloads on a copy stream per batch in flight, ordered by events)"}
        for i, e_ in enumerate(lanes_ctx):
            for hp_ in io[i][2] + io[i][3]:
                e_.host_free(hp_)
        for c2, _ in copy_ctx:
            c2.close()

    if rank == 0:
        # ------------------------------------------------------------ roofline of the dominant kernel (HIP events on the launch stream)
        eng.timing(True)
        eng.timing_read()
        reps = 3
        for _ in range(reps):          # one batch at a time on lane 0: per-kernel durations without overlap
            step_no[0] = 0
            step()
            eng.sync()
        tim = eng.timing_read()
        eng.timing(False)
        per_kernel = {kname: ms / cnt for kname, (ms, cnt) in tim.items()}
        # "dominant" = the kernel that consumes the most SIMD time (duration x SIMDs it occupies): the pairing
        # kernels run one 64-lane wave per SIMD, so a launch of n lanes occupies min(n/64, #SIMDs) of them.
        n_simd = n_cu * 4
        lane_count = {"k_ac17_dec_miller": B * 6, "k_ac17_dec_miller2": B * 3, "k_final_exp": B, "k_ac17_dec_miller_c3": B * 18, "k_final_exp_c3": B * 3,
                      "k_ac17_enc_rows": total_rows, "k_ac17_enc_c0": B * 3, "k_ac17_enc_cp": B}
        waves_per_simd = {"k_ac17_enc_rows": 4}
        simd_ms = {kk: v * min(n_simd, lane_count.get(kk, 0) / 64.0 / waves_per_simd.get(kk, 1)) for kk, v in per_kernel.items()}
        dom = max(simd_ms, key=lambda kk: simd_ms[kk])
        dom_ms = per_kernel[dom]
        # peak: dependent-free v_mad_u64_u32 issue rate measured live on this chip
        ms_c, ops_c = eng.calibrate(0, 20000)
        peak_tmac = ops_c / (ms_c * 1e-3) / 1e12
        m_avg = len(ct_sel_all) / B
        lanes = {"k_ac17_dec_miller": B * 6, "k_ac17_dec_miller2": B * 3, "k_final_exp": B, "k_ac17_dec_miller_c3": B * 6, "k_final_exp_c3": B, "k_ac17_enc_rows": total_rows,
                 "k_ac17_enc_c0": B * 3, "k_ac17_enc_cp": B}
        alg = {"k_ac17_dec_miller": SURVEY_MILLER_FPMUL + m_avg * SURVEY_MIXED_ADD_FPMUL,
               "k_ac17_dec_miller2": 2 * (SURVEY_MILLER_FPMUL + m_avg * SURVEY_MIXED_ADD_FPMUL),
               "k_ac17_dec_miller_c3": SURVEY_MILLER_FPMUL + m_avg * SURVEY_MIXED_ADD_FPMUL, "k_final_exp_c3": 9000 + 6 * 54,
               "k_final_exp": 9000 + 6 * 54, "k_ac17_enc_rows": 3 * 352, "k_ac17_enc_c0": 1056, "k_ac17_enc_cp": 2 * 1700 + 54}
        impl = {"k_ac17_dec_miller": IMPL_MILLER_FPMUL + m_avg * 11, "k_ac17_dec_miller2": IMPL_MILLER2_FPMUL + 2 * m_avg * 11, "k_final_exp": IMPL_FINAL_EXP_FPMUL + 6 * 54,
                "k_ac17_dec_miller_c3": IMPL_MILLER_FPMUL + m_avg * 11, "k_final_exp_c3": IMPL_FINAL_EXP_FPMUL + 6 * 54}
        macs = lanes.get(dom, 0) * alg.get(dom, 0) * MAC_PER_FPMUL
        achieved = macs / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        # HBM side (reported, not binding): algorithmic bytes of the whole step
        alg_bytes = B * (2 * 32 + 384) + total_rows * 192 * 2 + B * (384 + 384) * 2 + len(ct_sel_all) * 4 * 2 + B * 384
        traffic, traffic_src = pmc_traffic(dom)
        result["roofline"] = {
            "bound": "valu_int (v_mad_u64_u32 issue rate; not hbm, not mfma)", "kernel": dom,
            "kernel_ms": round(dom_ms, 4), "achieved": round(achieved, 4), "peak": round(peak_tmac, 3), "unit": "TMAC32/s",
            "frac": round(achieved / peak_tmac, 4) if peak_tmac else None,
            "work": "algorithmic Fp-muls/lane (SURVEY 8d) x 136 MAC32 x lanes = %.3e MAC32 per launch" % macs,
            "achieved_impl_count": round(lanes.get(dom, 0) * impl.get(dom, alg.get(dom, 0)) * MAC_PER_FPMUL / (dom_ms * 1e-3) / 1e12, 4)
            if dom_ms > 0 else None,
            "traffic": traffic, "traffic_unit": "bytes of HBM fetch + write per launch of the dominant kernel (PMC FETCH_SIZE + WRITE_SIZE, separate passes)",
            "traffic_source": traffic_src,
            "valu_issue": pmc_valu_issue(dom, lanes.get(dom, 0), impl.get(dom, alg.get(dom, 0))),
            "hbm": {"algorithmic_bytes_per_step": alg_bytes, "GBps_at_measured_step": round(alg_bytes / (elapsed / args.steps) / 1e9, 3),
                    "peak_GBps": 8000},
            "kernels_ms": {kk: round(v, 4) for kk, v in sorted(per_kernel.items(), key=lambda x: -x[1])},
            "kernels_simd_share": {kk: round(v / sum(simd_ms.values()), 3) for kk, v in sorted(simd_ms.items(), key=lambda x: -x[1])},
            "dominant_by": "SIMD time = duration x occupied SIMDs (one batch at a time, HIP events on the launch stream)",
        }
        # ------------------------------------------------------------ CPU baseline (oracle = reference-order restatement; checker only)
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(args, trees[0])
            except Exception as ex:  # the GPU number must still be reported
                result["cpu_baseline"] = {"error": repr(ex)}
            try:
                mc = cpu_baseline_multicore(args, trees[0])
                if mc:
                    result["cpu_baseline_multicore"] = mc
            except Exception as ex:
                result["cpu_baseline_multicore"] = {"error": repr(ex)}
        print(json.dumps(result), flush=True)

    pk.destroy()
    for e_ in lanes_ctx[1:]:
        e_.close()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (tools/pmc_traffic.sh writes them; rocprofv3
    cannot run inside the bench).  None when no summary names the kernel."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.txt")))
    for f in reversed(files):
        tot, seen = 0.0, 0
        for line in open(f):
            m = re.match(r"(FETCH_SIZE|WRITE_SIZE) (\S+) per launch: ([0-9.]+) KB-units", line)
            if m and m.group(2) == kernel:
                tot += float(m.group(3)) * 1024.0
                seen += 1
        if seen == 2:
            return round(tot), "profiles/" + os.path.basename(f)
    return None, None


def pmc_valu_issue(kernel, n_lanes, fpmul_per_lane):
    """VALU issue-slot occupancy of `kernel` from the newest committed SQ-counter summary (tools/pmc_sq.sh): VALU
    instructions per wave issue slot (SQ_WAVE_CYCLES counts 4-cycle slots), and the same with the second slot of every
    half-rate v_mad_u64_u32 added (64 + 43 MADs per Fp multiplication-equivalent of the lazy Fq2 arithmetic)."""
    import glob
    import re
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_sq.txt")))):
        vals, cur = {}, None
        for line in open(f):
            if not line.startswith(" "):
                cur = line.strip()
                continue
            m = re.match(r"\s+(SQ_\w+)\s+([0-9.e+]+) per launch", line)
            if m and cur == kernel:
                vals[m.group(1)] = float(m.group(2))
        if "SQ_INSTS_VALU" in vals and vals.get("SQ_WAVE_CYCLES"):
            mads = n_lanes * fpmul_per_lane * 107.0 / 64.0          # per-wave MAD count x waves
            return {"valu_per_slot": round(vals["SQ_INSTS_VALU"] / vals["SQ_WAVE_CYCLES"], 3),
                    "valu_slots_incl_mad_second_slot": round((vals["SQ_INSTS_VALU"] + mads) / vals["SQ_WAVE_CYCLES"], 3),
                    "source": "profiles/" + os.path.basename(f)}
    return None


def cpu_baseline_multicore(args, tree):
    """The same C restatement on several host cores at once (independent processes, a few items each): what a
    multi-threaded caller of the single-threaded reference would get.  Informational, beside cpu_baseline.
    Plain subprocesses with a hard deadline -- nothing here can hold up the GPU result."""
    import subprocess
    from rabe_amd import hostprep as hp
    from oracle import cport
    if not cport.available():
        return None
    procs = max(2, min(32, (os.cpu_count() or 2) // 2))
    per = 6
    policy = hp.to_json(tree)
    code = ("import sys, json; sys.path.insert(0, %r); from oracle import cport; "
            "o, dt = cport.ac17_encdec(%r, %d, %d, seed=int(sys.argv[1])); print(json.dumps([len(o), dt]))" % (ROOT, policy, args.attrs, per))
    env = dict(os.environ)
    env.pop("RANK", None)
    t0 = time.perf_counter()
    ps = [subprocess.Popen([sys.executable, "-c", code, str(args.seed + i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
          for i in range(procs)]
    deadline = t0 + 120.0
    done = []
    for pr in ps:
        try:
            out, _ = pr.communicate(timeout=max(1.0, deadline - time.perf_counter()))
            done.append(json.loads(out.decode().strip().splitlines()[-1]))
        except Exception:
            pr.kill()
    wall = time.perf_counter() - t0
    if not done:
        return None
    n = sum(d[0] for d in done)
    busy = max(d[1] for d in done)
    return {"value": round(n / busy, 3), "unit": "ops/s", "cores": len(done), "kind": "port",
            "sample": "%d processes x %d AC17 encrypt+decrypt at %d attributes, slowest process %.1f s in its timed loop (%.1f s wall with "
                      "interpreter start-up and key set-up); same C restatement as cpu_baseline" % (len(done), per, args.attrs, busy, wall)}


def cpu_baseline(args, tree):
    """Times the oracle (reference operation order) on the host CPU on a bounded sample of the same workload."""
    from rabe_amd import hostprep as hp
    policy = hp.to_json(tree)
    try:
        from oracle import cport
        have_c = cport.available()
    except Exception:
        have_c = False
    if have_c:
        n = args.cpu_sample or 48
        _outs, dt = cport.ac17_encdec(policy, args.attrs, n, seed=args.seed)
        return {"value": round(n / dt, 4), "unit": "ops/s", "cores": 1, "kind": "port",
                "sample": "%d AC17 encrypt+decrypt (policy parse + MSP + group loops) at %d attributes in %.1f s; C restatement "
                          "of the reference's operation order (oracle/c/rabe_ref.c: binary double-and-add for every G*Fr, "
                          "per-row hash-to-group, one final exponentiation per pairing), single thread like the reference"
                          % (n, args.attrs, dt)}
    # pure-Python big-int oracle: one item at a reduced attribute count scaled linearly in the encrypt part
    from oracle import bn254 as bn
    from oracle import policy as pol
    from oracle import schemes as sch
    from oracle.tape import SeededRng
    n_attr = min(args.attrs, 6)
    names = ["a%d" % (i + 1) for i in range(n_attr)]
    rnd = random.Random(args.seed)
    sub = hp.random_binary_tree(names, rnd)
    rng = SeededRng(args.seed)
    pk, msk = sch.ac17_setup(rng)
    sk = sch.ac17_cp_keygen(msk, names, rng)
    msg = bn.gt_pow(pk["e_gh_ka"][0], 12345)
    t0 = time.perf_counter()
    ct = sch.ac17_cp_encrypt(pk, hp.to_json(sub), pol.JSON, rng, msg)
    t_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = sch.ac17_cp_decrypt(sk, ct)
    t_dec = time.perf_counter() - t0
    assert out == msg
    est = t_enc * args.attrs / n_attr + t_dec
    return {"value": round(1.0 / est, 6), "unit": "ops/s", "cores": 1, "kind": "port",
            "sample": "pure-Python big-int oracle (reference operation order): 1 encrypt+decrypt at %d attributes measured "
                      "(%.1f s + %.1f s), encrypt scaled linearly to %d attributes" % (n_attr, t_enc, t_dec, args.attrs)}


if __name__ == "__main__":
    main()
