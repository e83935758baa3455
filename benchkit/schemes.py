"""bench.py --config 3 / 4 / 5: BASELINE.json configs[2..4] on the device-resident Level B paths.

  3  BSW CP-ABE, 100-attribute access tree, batch 4096 encrypt+decrypt               (bsw/mod.rs:217-318)
  4  LSW KP-ABE keygen+decrypt, 200 attributes, batch 16384 over 8 GPUs (2048 / GPU)   (lsw/mod.rs:121-290)
  5  AW11 multi-authority, 10 authorities x 20 attributes, batch 8192 over 8 GPUs      (aw11/mod.rs:241-366)

Same contract and JSON schema as config 2 (bench.py): inputs resident in HBM (key material, flattened policy tables,
selection tables, per-item explicit randomness), one step = one pass over one batch, steps submitted in groups that are
contiguous in HBM, exactly-K-step timed regions, decrypt(encrypt(msg)) == msg checked bit for bit on every item."""
import json
import os
import random
import time

from benchkit.lib import G1_GEN, G2_GEN, MAC_PER_FPMUL, ExtBuf, cpu_model, regions_summary, split_steps, timed_regions


def run(args, world, rank, local_rank, dist_on=None):
    cls = {3: BswBench, 4: LswBench, 5: Aw11Bench}.get(args.config)
    if cls is None:
        raise SystemExit("bench.py --config %d: the device-resident path of this scheme is not built yet" % args.config)
    r = SchemeRunner(cls, args, world, rank, local_rank)
    r.dist_on = (world > 1) if dist_on is None else dist_on          # RABE_FORCE_DIST: the collective path with one rank (bench.py)
    return r.run()


class SchemeRunner:
    def __init__(self, cls, args, world, rank, local_rank):
        import torch
        from rabe_amd import Engine
        self.torch = torch
        self.args, self.world, self.rank, self.local_rank = args, world, rank, local_rank
        self.eng = Engine(local_rank)
        self.n_cu, self.dev_name = self.eng.device_info()
        self.stream = torch.cuda.Stream(device=local_rank)
        self.eng.set_stream(self.stream.cuda_stream)
        self.dev = torch.device("cuda", local_rank)
        self.bench = cls(self)

    def new_lane(self):
        from rabe_amd import Engine
        e2 = Engine(self.local_rank)
        st = self.torch.cuda.Stream(device=self.local_rank)
        e2.set_stream(st.cuda_stream)
        self._streams.append(st)
        return e2

    def run(self):
        import torch
        import torch.distributed as dist
        args, world, rank, eng, b = self.args, self.world, self.rank, self.eng, self.bench
        sizes = split_steps(args.steps, max(1, min(b.default_group if args.group == 16 else args.group, args.steps)))
        G = max(sizes)
        S = max(1, min(args.inflight, len(sizes)))
        self._streams = [self.stream]
        lanes = [eng] + [self.new_lane() for _ in range(S - 1)]
        b.prepare(G, lanes)
        launch_no = [0]

        def run_steps():
            launch_no[0] = 0
            for g_ in sizes:
                b.submit(launch_no[0] % S, g_)
                launch_no[0] += 1

        def sync_all():
            for e_ in lanes:
                e_.sync()

        if args.warmup:
            for _ in range((args.warmup + args.steps - 1) // args.steps):
                run_steps()
        sync_all()
        torch.cuda.synchronize()
        regions = timed_regions(run_steps, sync_all, args.min_time)
        elapsed = sum(regions) / len(regions)
        used = {}
        for j, g_ in enumerate(sizes):
            used[j % S] = g_
        ok = all(b.check(i, g_) for i, g_ in used.items())
        if self.dist_on:
            f = torch.tensor([1 if ok else 0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            ok = bool(f.item())
        B = b.B
        gather = None
        if self.dist_on:
            # the trivial gather: every rank's first-slot result records (B x 384 B) in ONE all_gather_into_tensor on device (RCCL);
            # rank 0 recomputes the unsharded batch's messages from the global randomness stream and compares in global order
            mine = b.out_tensor(0)[:B * 384]
            src = mine if dist.get_backend() == "nccl" else mine.cpu()
            allrec = torch.empty(world * B * 384, dtype=torch.uint8, device=src.device)
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            dist.all_gather_into_tensor(allrec, src)
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg0
            if rank == 0:
                gather = {"collective": "all_gather_into_tensor", "backend": dist.get_backend(), "bytes_per_rank": B * 384, "ms": round(1e3 * tg, 3),
                          "matches_unsharded_order": allrec.cpu().numpy().tobytes() == b.expected_first_slot_all_ranks()}
        value = world * B * args.steps / elapsed
        result = {
            "metric": b.metric, "value": round(value, 2), "unit": "ops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "dtype_note": "exact BN254 integers in 32-bit limbs (Montgomery: 8 x 32 bits; 9 signed x 29 bits in the pairing kernels of full launches)", "data": "synthetic", "roundtrip_bit_exact": ok,
            "timed_regions": regions_summary(regions, args.steps),
            "config": dict(b.describe(), batch_per_gpu=B, steps_per_launch_set=sizes, launch_sets_in_flight=S,
                           hw_queues=os.environ.get("GPU_MAX_HW_QUEUES", "default"),
                           parallelism="batch-sharded x%d (no data-path collective)" % world, device=self.dev_name),
        }
        if gather is not None:
            result["gather"] = gather
        if rank == 0:
            eng.timing(True)
            eng.timing_read()
            for _ in range(2):
                b.submit(0, G)
                eng.sync()
            tim = eng.timing_read()
            eng.timing(False)
            per_kernel = {k: ms / cnt * b.launches_per_submit.get(k, 1) for k, (ms, cnt) in tim.items()}
            dom = max(per_kernel, key=lambda k: per_kernel[k])
            ms_c, ops_c = min((eng.calibrate(0, 20000) for _ in range(2)), key=lambda r_: r_[0])          # best of two: the first may see clocks ramping
            peak = ops_c / (ms_c * 1e-3) / 1e12
            alg = b.algorithmic_fpmul_per_item()          # {kernel: SURVEY 8d Fp-mul per item carried by that kernel}
            impl_all = b.impl_fpmul_per_item()
            # the reduced-radix kernels (engine_rr.hip) carry the same units of work; what they execute is counted in multiply-add instructions
            # (tests/count_muls.py: 1.077 M per walking pair of a 14-pair chunk, 0.806 M per pair when half of them replay prepared lines),
            # expressed here in units of 136 like the Fp multiplications of the 8 x 32-bit kernels
            if "k_miller_multi" in alg:
                alg["k_miller_multi_rr"] = alg["k_miller_multi"]
                if "k_miller_multi" in impl_all:
                    pairs = impl_all["k_miller_multi"] / (4766.0 if getattr(b, "sk_lines", None) else 5976.0)
                    impl_all["k_miller_multi_rr"] = pairs * ((805505.0 if getattr(b, "sk_lines", None) else 1077381.0) / 136.0)
            if "k_final_exp" in alg:
                alg["k_final_exp_rr"] = alg["k_final_exp"]
            macs = alg.get(dom, 0) * MAC_PER_FPMUL * G * B
            achieved = macs / (per_kernel[dom] * 1e-3) / 1e12 if per_kernel[dom] > 0 else 0.0
            traffic, tsrc = pmc_traffic(dom, args.config)
            impl = impl_all.get(dom)
            achieved_impl = (impl or alg.get(dom, 0)) * MAC_PER_FPMUL * G * B / (per_kernel[dom] * 1e-3) / 1e12 if per_kernel[dom] > 0 else 0.0
            result["roofline"] = {
                "bound": "valu_int (v_mad_u64_u32 issue rate; not hbm, not mfma)", "kernel": dom, "kernel_ms": round(per_kernel[dom], 4),
                "items_per_launch": G * B, "steps_per_launch": G, "kernel_ms_per_step": round(per_kernel[dom] / G, 4),
                "achieved": round(achieved_impl, 4), "peak": round(peak, 3), "unit": "TMAC32/s", "frac": round(achieved_impl / peak, 4) if peak else None,
                "work": "Fp multiplications this engine executes per item in this kernel (instrumented, tests/count_muls.py: %.3g) x 136 MAC32 x items "
                        "= %.3e MAC32 per launch" % (impl or alg.get(dom, 0), (impl or alg.get(dom, 0)) * MAC_PER_FPMUL * G * B),
                "achieved_survey": round(achieved, 4), "frac_survey": round(achieved / peak, 4) if peak else None,
                "work_survey": "SURVEY 8d algorithmic Fp-muls per item carried by this kernel (%.3g) x 136 MAC32 x items = %.3e MAC32 per launch; the engine "
                               "executes fewer (shared squarings, merged / prepared lines): `frac` is the utilisation of the multiplier" % (alg.get(dom, 0), macs),
                "whole_step_frac": round(b.survey_fpmul_per_item * MAC_PER_FPMUL * B / (elapsed / args.steps) / 1e12 / peak, 4) if peak else None,
                "whole_step_work": "SURVEY 8d: %.3g M Fp-mul per op" % (b.survey_fpmul_per_item / 1e6),
                "traffic": traffic, "traffic_source": tsrc,
                "traffic_unit": "bytes of HBM fetch + write per launch of the dominant kernel (PMC FETCH_SIZE + WRITE_SIZE, separate passes)",
                "hbm": {"algorithmic_bytes_per_step": b.algorithmic_bytes_per_step(),
                        "GBps_at_measured_step": round(b.algorithmic_bytes_per_step() / (elapsed / args.steps) / 1e9, 3), "peak_GBps": 8000},
                "kernels_ms": {k: round(v, 4) for k, v in sorted(per_kernel.items(), key=lambda x: -x[1])},
                "kernels_ms_sum_per_step": round(sum(per_kernel.values()) / G, 4),
                "dominant_by": "duration of the kernel's launches for one full group, HIP events on the launch stream, nothing else running",
            }
            if world == 1 and not args.no_object_api:
                try:
                    result["object_api"] = packed_api_leg(args, b)
                except Exception as ex:
                    result["object_api"] = {"error": repr(ex)[:300]}
            if world == 1 and not args.no_cpu_baseline:
                try:
                    result["cpu_baseline"] = b.cpu_baseline()
                except Exception as ex:
                    result["cpu_baseline"] = {"error": repr(ex)}
            from benchkit.lib import emit_line
            emit_line(result)
        b.close()
        for e_ in lanes[1:]:
            e_.close()
        eng.close()
        if self.dist_on:
            dist.destroy_process_group()


def packed_api_leg(args, b):
    """The same configuration through the reference-shaped host layer's PACKED entry points (rabe_{bsw,lsw,aw11}_*_packed): policy
    text and plaintext bytes in, canonical records out and back, with parse / flattening / pruning / KDF + AES-GCM / record assembly
    and the PCIe copies inside the timed region.  These feed the same device-resident kernels the headline times."""
    import numpy as np
    from rabe_amd import hostlib as hl
    from rabe_amd import hostprep as hp
    host = hl.Host(0)
    try:
        n = b.B * b.default_group          # one packed call carries a launch set's worth of items, like the device-level groups
        pols = [hp.to_json(t) for t in b.trees]
        item_pol = np.arange(n, dtype=np.uint32) % len(pols)
        pts = [b"dance like no one's watching, encrypt like everyone is!" + i.to_bytes(4, "little") for i in range(n)]
        pt_blob = np.frombuffer(b"".join(pts), dtype=np.uint8)
        pt_off = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.uint64)
        out = {"batch": n}

        def best_of(fn, reps=3):
            best = None
            for rep in range(reps):
                t0 = time.perf_counter()
                r_ = fn()
                dt = time.perf_counter() - t0
                if rep and (best is None or dt < best[0]):
                    best = (dt, r_)
            return best

        def io_buffers(first_call):
            """caller-allocated, RE-USED output buffers, as a server keeps them (the first call also builds tables and staging areas:
            untimed): fresh pages cost a page fault each when the device's copy lands in them (185 MB: 11.5 instead of 3.3 ms)"""
            blob0, _ = first_call()
            ct_buf = np.zeros(blob0.size, dtype=np.uint8)
            return ct_buf, np.zeros(blob0.size, dtype=np.uint8)
        if args.config == 3:
            from rabe_amd.schemes import bsw
            pk, msk = bsw.setup(host)
            sk = bsw.keygen(host, pk, msk, b.attrs)
            ct_buf, pt_buf = io_buffers(lambda: bsw.encrypt_packed(host, pk, pols, item_pol, pt_blob, pt_off))
            te, (blob, off) = best_of(lambda: bsw.encrypt_packed(host, pk, pols, item_pol, pt_blob, pt_off, out=ct_buf))
            td, (o, _, st) = best_of(lambda: bsw.decrypt_packed(host, sk, blob, off, out=pt_buf))
            tt_, (o2, _, st2) = best_of(lambda: bsw.decrypt_packed(host, sk, blob, off, out=pt_buf, trusted=True))
            ok = o.tobytes() == pt_blob.tobytes() and not st.any() and o2.tobytes() == pt_blob.tobytes() and not st2.any()
        elif args.config == 4:
            from rabe_amd.schemes import lsw
            pk, msk = lsw.setup(host)
            ct = lsw.encrypt(host, pk, b.attrs, pts[0])
            ct_buf, pt_buf = io_buffers(lambda: lsw.keygen_packed(host, pk, msk, pols, item_pol))
            te, (blob, off) = best_of(lambda: lsw.keygen_packed(host, pk, msk, pols, item_pol, out=ct_buf))
            td, (o, _, st) = best_of(lambda: lsw.decrypt_packed(host, ct, blob, off, out=pt_buf))
            tt_, (o2, _, st2) = best_of(lambda: lsw.decrypt_packed(host, ct, blob, off, out=pt_buf, trusted=True))
            ok = o.tobytes() == pts[0] * n and not st.any() and o2.tobytes() == pts[0] * n and not st2.any()
        else:
            from rabe_amd.schemes import aw11
            gk = aw11.setup(host)
            per = b.n_attr // b.n_auth
            auth = [aw11.authgen(host, gk, b.attrs[a * per:(a + 1) * per]) for a in range(b.n_auth)]
            sk = aw11.keygen(host, gk, auth[0][1], "alice", b.attrs[:per])
            for a in range(1, b.n_auth):
                for nm in b.attrs[a * per:(a + 1) * per]:
                    aw11.add_to_attribute(host, gk, auth[a][1], nm, sk)
            pks = [a[0] for a in auth]
            ct_buf, pt_buf = io_buffers(lambda: aw11.encrypt_packed(host, gk, pks, pols, item_pol, pt_blob, pt_off))
            te, (blob, off) = best_of(lambda: aw11.encrypt_packed(host, gk, pks, pols, item_pol, pt_blob, pt_off, out=ct_buf))
            td, (o, _, st) = best_of(lambda: aw11.decrypt_packed(host, gk, sk, blob, off, out=pt_buf))
            tt_, (o2, _, st2) = best_of(lambda: aw11.decrypt_packed(host, gk, sk, blob, off, out=pt_buf, trusted=True))
            ok = o.tobytes() == pt_blob.tobytes() and not st.any() and o2.tobytes() == pt_blob.tobytes() and not st2.any()
        first = "keygen_s" if args.config == 4 else "encrypt_s"
        out.update({"ops_per_s": round(n / (te + td), 1), first: round(te, 4), "decrypt_s": round(td, 4), "decrypt_trusted_s": round(tt_, 4),
                    "ops_per_s_trusted": round(n / (te + tt_), 1), "record_bytes": int(blob.size), "plaintexts_match": bool(ok),
                    "note": "packed entry points of the host layer (one blob of canonical records + offsets per side); decrypt_s includes the batched "
                            "group-membership pass over every decoded element (G2 elements: r * P = O), *_trusted skips it (RABE_PACKED_TRUSTED)"})
        return out
    finally:
        host.close()


def pmc_traffic(kernel, config):
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in reversed(sorted(glob.glob(os.path.join(root, "profiles", "*_cfg%d_pmc_traffic.txt" % config)))):
        tot, seen = 0.0, 0
        for line in open(f):
            m = re.match(r"(FETCH_SIZE|WRITE_SIZE) (\S+) per launch: ([0-9.]+) KB-units", line)
            if m and m.group(2) == kernel:
                tot += float(m.group(3)) * 1024.0
                seen += 1
        if seen == 2:
            return round(tot), "profiles/" + os.path.basename(f)
    return None, None


# ====================================================================================================================== BSW
class BswBench:
    """config 3: bsw::encrypt + bsw::decrypt, n-leaf access tree, one public key, one secret key with all attributes."""
    metric = "ABE ops/sec (BSW CP-ABE encrypt+decrypt)"
    default_group = 8                           # steps per launch set: 72.1 / 77.4 / 80.4 k ops/s at 2 / 4 / 8 (the final exponentiation's fixed 6.8 ms is shared by more items)
    survey_fpmul_per_item = 2.3e6               # SURVEY.md 8d, config 3 restructured work
    launches_per_submit = {"k_table_mul_g1": 2}

    def __init__(self, r):
        from rabe_amd import engine as E
        from rabe_amd import hostprep as hp
        self.r, self.E, self.hp = r, E, hp
        args, eng = r.args, r.eng
        self.B = args.batch or 4096
        self.n_attr = args.attrs or 100
        R, le = hp.R_ORDER, hp.fr_le
        krnd = random.Random(args.seed * 1000003 + 3)

        def kfr():
            return krnd.randrange(1, R)
        # ---- setup (bsw/mod.rs:92-114) and keygen (:125-152) through Level E / table kernels (untimed input generation)
        g1 = eng.g1_mul([G1_GEN], [le(kfr())])[0]
        g2 = eng.g2_mul([G2_GEN], [le(kfr())])[0]
        beta, alpha = kfr(), kfr()
        h = eng.g1_mul([g1], [le(beta)])[0]
        e_gg_alpha = eng.pairing([g1], [eng.g2_mul([g2], [le(alpha)])[0]])[0]
        eng.sync()
        t0 = time.perf_counter()
        self.pk = E.BswPk(eng, g1, g2, h, e_gg_alpha)
        eng.sync()
        self.table_build_ms = 1e3 * (time.perf_counter() - t0)
        self.attrs = ["b%d" % i for i in range(self.n_attr)]
        t1, t2 = eng.g1_table(g1), eng.g2_table(g2)
        rr = kfr()
        rj = [kfr() for _ in self.attrs]
        beta_inv = pow(beta, R - 2, R)
        self.d_sk_d = eng.upload(t2.mul([le((alpha + rr) * beta_inv)])[0])
        self.d_sk_g1 = eng.upload(b"".join(t1.mul([le(x) for x in rj])))
        self.d_sk_g2 = eng.upload(b"".join(t2.mul([le(rr + hp.h_fr(a) * x) for a, x in zip(self.attrs, rj)])))
        self.d_sk_attr_off = eng.upload_u32([0, self.n_attr])
        t1.destroy()
        t2.destroy()
        self.e_tab = eng.gt_table(e_gg_alpha)
        self.sk_lines = None if args.no_prepared_sk else E.BswSkLines(eng, 1, self.n_attr, self.d_sk_d, self.d_sk_g2)
        # ---- policies
        prnd = random.Random(args.seed)
        self.trees = [self.make_tree(prnd) for _ in range(args.policies)]
        t0 = time.perf_counter()
        self.tt = hp.TreeTables(self.trees)
        self.sel = []
        for t in self.trees:
            ok, idx = hp.pruned_leaf_indices(self.attrs, t)
            assert ok
            z = hp.leaf_coefficients(t)
            names = hp.flatten_tree(t)["names"]
            self.sel.append((idx, [self.attrs.index(names[y]) for y in idx], [z[y] for y in idx]))
        self.host_prep_ms_per_policy = 1e3 * (time.perf_counter() - t0) / len(self.trees)
        self.dtt = E.DevTreeTables(eng, self.tt)

    def make_tree(self, prnd):
        names = list(self.attrs)
        prnd.shuffle(names)
        if self.r.args.ragged:                   # a batch of mixed shapes: 10 .. n_attr leaves per policy (an even count for the mixed tree)
            k = prnd.randrange(10, self.n_attr + 1)
            names = names[:k - (k % 2)]
        return make_tree(self.r.args.tree, names)

    def prepare(self, G, lanes):
        r, eng, E, hp, tt = self.r, self.r.eng, self.E, self.hp, self.tt
        R, le = hp.R_ORDER, hp.fr_le
        B, P = self.B, len(self.trees)
        GB = G * B
        self.G, self.lanes = G, lanes
        # items of one policy are interleaved (uniform shapes) or -- ragged -- contiguous runs of B / P items inside every step's batch:
        # a caller that batches mixed shapes groups them (the host layer's packed decrypt does), so that a wave sees one shape
        pol = [((i % B) * P // B) if r.args.ragged else i % P for i in range(GB)]
        sel_start_p, so = [], 0
        sel_ct, sel_sk, sel_z = [], [], []
        for idx, ska, z in self.sel:
            sel_start_p.append(so)
            sel_ct += idx
            sel_sk += ska
            sel_z += z
            so += len(idx)
        self.n_sel = so
        leaf_off, pair_off, coef_off = [0], [0], [0]
        for p in pol:
            leaf_off.append(leaf_off[-1] + tt.n_leaves(p))
            pair_off.append(pair_off[-1] + 2 * len(self.sel[p][0]) + 1)
            coef_off.append(coef_off[-1] + tt.n_coef(p))
        assert all(leaf_off[(j + 1) * B] == (j + 1) * leaf_off[B] for j in range(G)), "batch must be a multiple of the policy count"
        self.leaves_per_batch, self.pairs_per_batch = leaf_off[B], pair_off[B]
        self.max_pairs = max(2 * len(s[0]) + 1 for s in self.sel)
        self.d_leaf_off = eng.upload_u32(leaf_off)
        self.d_pair_off = eng.upload_u32(pair_off)
        self.d_item_tree_leaf = eng.upload_u32([tt.first_leaf[p] for p in pol])
        self.d_item_tree_gate = eng.upload_u32([tt.first_gate[p] for p in pol])
        self.d_item_coef_off = eng.upload_u32(coef_off[:-1])
        self.d_sel_start = eng.upload_u32([sel_start_p[p] for p in pol])
        self.d_sel_ct, self.d_sel_sk = eng.upload_u32(sel_ct), eng.upload_u32(sel_sk)
        self.d_sel_z = eng.upload(b"".join(le(z) for z in sel_z))
        self.d_sk_idx = eng.upload_u32([0] * GB)
        # per-item explicit randomness (bsw::encrypt draw order: secret, msg, gate coefficients)
        irnd = random.Random(r.args.seed * 7919 + 17 + r.rank)
        self.d_secret = eng.upload(b"".join(le(irnd.randrange(1, R)) for _ in range(GB)))
        self.rho_all, rho_grp = global_rho(r, B, G, R)
        rho = eng.upload(b"".join(le(x) for x in rho_grp))
        self.d_msg = eng.alloc(GB * 384)
        eng._check(eng.lib.rhip_gt_table_pow(eng.ctx, self.e_tab.h, E._sz(GB), rho.ptr, self.d_msg.ptr))
        n_coef = coef_off[-1]
        # 31 random bytes per coefficient (< 2^248 < r): plenty for a polynomial coefficient of a throughput run
        raw = irnd.randbytes(31 * n_coef)
        self.d_coef = eng.upload(b"".join(raw[31 * i:31 * i + 31] + b"\0" for i in range(n_coef)) or bytes(32))
        total_leaves = leaf_off[-1]
        self.bufs = [(e_.alloc(GB * 64), e_.alloc(GB * 384), e_.alloc(total_leaves * 64), e_.alloc(total_leaves * 128),
                      ExtBuf(r.torch, GB * 384, r.dev)) for e_ in lanes]
        eng.sync()

    def submit(self, lane, g):
        E, e_ = self.E, self.lanes[lane]
        c, cp, g1, g2, out = self.bufs[lane]
        n = g * self.B
        E.bsw_encrypt_dev(e_, self.pk, n, g * self.leaves_per_batch, self.d_leaf_off, self.d_item_tree_leaf, self.d_item_tree_gate, self.dtt,
                          self.d_secret, self.d_coef, self.d_item_coef_off, self.d_msg, c, cp, g1, g2)
        if self.r.args.only_encrypt:
            return
        if os.environ.get("RABE_BSW_GENERAL_DECRYPT"):        # A/B: the general entry point (a key index per item)
            E.bsw_decrypt_dev(e_, n, self.max_pairs, g * self.pairs_per_batch, self.d_pair_off, self.d_sel_start, self.d_sel_ct, self.d_sel_sk,
                              self.d_sel_z, c, cp, g1, g2, self.d_leaf_off, self.d_sk_d, self.d_sk_g1, self.d_sk_g2, self.d_sk_attr_off,
                              self.d_sk_idx, self.sk_lines, out)
            return
        # one key decrypts every ciphertext of the batch: the one-key entry point (the key's scaled Dj.g1 once per selection entry)
        E.bsw_decrypt_one_sk_dev(e_, n, self.max_pairs, g * self.pairs_per_batch, self.n_sel, self.d_pair_off, self.d_sel_start, self.d_sel_ct, self.d_sel_sk,
                                 self.d_sel_z, c, cp, g1, g2, self.d_leaf_off, self.d_sk_d, self.d_sk_g1, self.d_sk_g2, self.d_sk_attr_off, self.sk_lines, out)

    def check(self, lane, g):
        n = g * self.B * 384
        return self.bufs[lane][4].t[:n].cpu().numpy().tobytes() == self.r.eng.download(self.d_msg)[:n]

    def out_tensor(self, lane):
        return self.bufs[lane][4].t

    def expected_first_slot_all_ranks(self):
        eng, E = self.r.eng, self.E
        n = len(self.rho_all)
        d = eng.alloc(n * 384)
        eng._check(eng.lib.rhip_gt_table_pow(eng.ctx, self.e_tab.h, E._sz(n), eng.upload(b"".join(self.hp.fr_le(x) for x in self.rho_all)).ptr, d.ptr))
        return eng.download(d)

    def describe(self):
        m = sum(len(s[0]) for s in self.sel) / len(self.sel)
        return {"workload": "BSW CP-ABE, %d-leaf %s access tree (%d distinct policies), batch %d encrypt+decrypt per GPU"
                            % (self.n_attr, self.r.args.tree, len(self.trees), self.B),
                "attrs": self.n_attr, "policies": len(self.trees), "tree": self.r.args.tree, "ragged": bool(self.r.args.ragged),
                "leaves_per_policy": [self.tt.n_leaves(p) for p in range(len(self.trees))], "pruned_leaves_avg": round(m, 2),
                "pairings_per_item": round(2 * m + 1, 1), "prepared_key": self.sk_lines is not None,
                "table_build_ms_per_public_key": round(self.table_build_ms, 1), "host_prep_ms_per_policy": round(self.host_prep_ms_per_policy, 3)}

    def algorithmic_fpmul_per_item(self):
        # enc 101 fixed-base G1 (36 kM) + 100 fixed-base G2 (106 kM) + 1.7 kM; dec 200 var-base G1 (560 kM) +
        # 201 Miller (1.61 MM) + 1 final exponentiation (9 kM); scaled to the pruned leaf count m
        m = sum(len(s[0]) for s in self.sel) / len(self.sel)
        n = self.n_attr
        return {"k_miller_multi": (2 * m + 1) * 8000.0, "k_bsw_dec_pairs": 2 * m * 2800.0, "k_final_exp": 9000.0, "k_table_mul_g2": n * 1056.0,
                "k_table_mul_g1": (n + 1) * 352.0, "k_table_pow_gt_mul": 1700.0, "k_bsw_enc_scalars": 0.0}

    def impl_fpmul_per_item(self):
        # tests/count_muls.py: miller_loop_multi 4.77 kM per pair with half of the pairs replaying prepared lines, 5.98 kM all walking
        m = sum(len(s[0]) for s in self.sel) / len(self.sel)
        return {"k_miller_multi": (2 * m + 1) * (4766.0 if self.sk_lines else 5976.0), "k_bsw_dec_pairs": 2 * m * 2750.0, "k_final_exp": 7553.0 + 15 * 54}

    def algorithmic_bytes_per_step(self):
        B, n = self.B, self.n_attr
        return B * (32 + 384 + 64 + 384 + 384) + self.leaves_per_batch * (192 * 2 + 32) + self.pairs_per_batch * 4

    def cpu_baseline(self):
        from oracle import cport
        if not cport.available():
            return {"error": "oracle/c not built"}
        n = self.r.args.cpu_sample or 12
        dt = cport.bsw_encdec(self.n_attr, n, tree=self.r.args.tree, seed=self.r.args.seed)
        return {"value": round(n / dt, 4), "unit": "ops/s", "cores": 1, "cpu_model": cpu_model(), "kind": "port",
                "sample": "%d BSW encrypt+decrypt at %d leaves (%s tree) in %.1f s; the reference's operation order (bsw/mod.rs:217-318: binary "
                          "double-and-add for every G*Fr, two multiplications for (g2*h(name))*q, one full pairing per e(.,.), Gt::pow per leaf) "
                          "over the C primitives of oracle/c/rabe_ref.c, single thread like the reference" % (n, self.n_attr, self.r.args.tree, dt)}

    def close(self):
        if self.sk_lines:
            self.sk_lines.destroy()
        self.pk.destroy()


# ====================================================================================================================== shared helpers
def make_tree(kind, names, binary_and_only=False):
    """flat: one n-ary AND (binary-AND schemes: right-nested instead); nested: balanced binary ANDs; mixed: ANDs over two-leaf ORs"""
    def nest(nodes):
        if len(nodes) == 1:
            return nodes[0]
        return ("and", [nest(nodes[:len(nodes) // 2]), nest(nodes[len(nodes) // 2:])])
    leaves = [("leaf", x) for x in names]
    if kind == "mixed":
        ors = [("or", [leaves[2 * i], leaves[2 * i + 1]]) for i in range(len(leaves) // 2)]
        return nest(ors) if binary_and_only else ("and", ors)
    if kind == "flat" and not binary_and_only:
        return ("and", leaves)
    return nest(leaves)


def global_rho(r, B, G, R):
    """Message exponents of the GLOBAL batch (world x B items per step, one stream for all ranks): (all of them, this rank's
    group array): slot j of a group re-uses the step's exponents shifted by j."""
    from rabe_amd import shard
    grnd = random.Random(r.args.seed * 7919 + 29)
    rho_all = [grnd.randrange(1, R) for _ in range(r.world * B)]
    lo, hi = shard.shard_range(r.world * B, r.rank, r.world)
    loc = rho_all[lo:hi]
    return rho_all, [((loc[i % B] + i // B) % R) or 1 for i in range(G * B)]


def rand_fr_bytes(irnd, n):
    """n canonical Fr values of 248 random bits each (< r): bulk randomness for throughput runs"""
    raw = irnd.randbytes(31 * n)
    return b"".join(raw[31 * i:31 * i + 31] + b"\0" for i in range(n)) or bytes(32)


# ====================================================================================================================== LSW
class LswBench:
    """config 4: lsw::keygen + lsw::decrypt of a pre-made ciphertext, n positive leaves, a fresh key per item."""
    metric = "ABE ops/sec (LSW KP-ABE keygen+decrypt)"
    default_group = 8                           # 48.0 / 52.9 / 56.6 k ops/s at 2 / 4 / 8 steps per launch set
    survey_fpmul_per_item = 3.0e6               # SURVEY.md 8d, config 4 restructured work
    launches_per_submit = {}

    def __init__(self, r):
        from rabe_amd import engine as E
        from rabe_amd import hostprep as hp
        self.r, self.E, self.hp = r, E, hp
        args, eng = r.args, r.eng
        self.B = args.batch or 2048
        self.n_attr = args.attrs or 200
        R, le = hp.R_ORDER, hp.fr_le
        krnd = random.Random(args.seed * 1000003 + 4)

        def kfr():
            return krnd.randrange(1, R)
        g1 = eng.g1_mul([G1_GEN], [le(kfr())])[0]
        g2 = eng.g2_mul([G2_GEN], [le(kfr())])[0]
        self.alpha1, self.alpha2 = kfr(), kfr()
        e_gg_alpha = eng.gt_pow(eng.pairing([g1], [g2]), [le(self.alpha1 * self.alpha2)])[0]
        eng.sync()
        t0 = time.perf_counter()
        self.pk = E.LswPk(eng, g1, g2)
        eng.sync()
        self.table_build_ms = 1e3 * (time.perf_counter() - t0)
        self.attrs = ["c%d" % i for i in range(self.n_attr)]
        # the pre-made ciphertext (lsw::encrypt, lsw/mod.rs:180-219): E1_y = (g1*h(a))*s, e2 = g2*s, e1 = e_gg_alpha^s * msg
        s = kfr()
        t1 = eng.g1_table(g1)
        self.d_ct_e1j = eng.upload(b"".join(t1.mul([le(hp.h_fr(a) * s) for a in self.attrs])))
        t1.destroy()
        self.d_ct_e2 = eng.upload(eng.g2_mul([g2], [le(s)])[0])
        self.msg = eng.gt_pow([e_gg_alpha], [le(kfr())])[0]
        self.e1 = eng.gt_mul([eng.gt_pow([e_gg_alpha], [le(s)])[0]], [self.msg])[0]
        self.d_ct_attr_off = eng.upload_u32([0, self.n_attr])
        self.e2_lines = None if args.no_prepared_sk else E.G2Lines(eng, 1, self.d_ct_e2)
        self.d_alpha = eng.upload(le(self.alpha1) + le(self.alpha2))
        prnd = random.Random(args.seed)
        self.trees = []
        for _ in range(args.policies):
            names = list(self.attrs)
            prnd.shuffle(names)
            if args.ragged:                      # keys of mixed shapes: 10 .. n_attr leaves per policy (an even count for the mixed tree)
                k = prnd.randrange(10, self.n_attr + 1)
                names = names[:k - (k % 2)]
            self.trees.append(make_tree(args.tree, names))
        t0 = time.perf_counter()
        self.tt = hp.TreeTables(self.trees)
        self.sel = []
        for t in self.trees:
            ok, idx = hp.pruned_leaf_indices(self.attrs, t)
            assert ok
            z = hp.leaf_coefficients(t)
            names = hp.flatten_tree(t)["names"]
            self.sel.append((idx, [self.attrs.index(names[y]) for y in idx], [z[y] for y in idx]))
        self.host_prep_ms_per_policy = 1e3 * (time.perf_counter() - t0) / len(self.trees)
        self.dtt = E.DevTreeTables(eng, self.tt)

    def prepare(self, G, lanes):
        r, eng, hp, tt = self.r, self.r.eng, self.hp, self.tt
        le = hp.fr_le
        B, P = self.B, len(self.trees)
        GB = G * B
        self.G, self.lanes = G, lanes
        # ragged: the items of one policy are contiguous inside every step's batch (like config 3's mixed-shape batch)
        pol = [((i % B) * P // B) if r.args.ragged else i % P for i in range(GB)]
        sel_start_p, so, sel_sk, sel_ct, sel_z = [], 0, [], [], []
        for idx, cta, z in self.sel:
            sel_start_p.append(so)
            sel_sk += idx
            sel_ct += cta
            sel_z += z
            so += len(idx)
        self.n_sel = so
        leaf_off, pair_off, coef_off = [0], [0], [0]
        for p in pol:
            leaf_off.append(leaf_off[-1] + tt.n_leaves(p))
            pair_off.append(pair_off[-1] + len(self.sel[p][0]) + 1)
            coef_off.append(coef_off[-1] + tt.n_coef(p))
        assert all(leaf_off[(j + 1) * B] == (j + 1) * leaf_off[B] for j in range(G)), "batch must be a multiple of the policy count"
        self.leaves_per_batch, self.pairs_per_batch = leaf_off[B], pair_off[B]
        self.max_pairs = max(len(s[0]) + 1 for s in self.sel)
        self.d_leaf_off = eng.upload_u32(leaf_off)
        self.d_pair_off = eng.upload_u32(pair_off)
        self.d_item_tree_leaf = eng.upload_u32([tt.first_leaf[p] for p in pol])
        self.d_item_tree_gate = eng.upload_u32([tt.first_gate[p] for p in pol])
        self.d_item_coef_off = eng.upload_u32(coef_off[:-1])
        self.d_sel_start = eng.upload_u32([sel_start_p[p] for p in pol])
        self.d_sel_sk, self.d_sel_ct = eng.upload_u32(sel_sk), eng.upload_u32(sel_ct)
        self.d_sel_z = eng.upload(b"".join(le(z) for z in sel_z))
        self.d_ct_idx = eng.upload_u32([0] * GB)
        self.d_ct_e1 = eng.upload(self.e1 * GB)
        irnd = random.Random(r.args.seed * 7919 + 19 + r.rank)
        self.d_coef = eng.upload(rand_fr_bytes(irnd, coef_off[-1]))
        self.d_rand = eng.upload(rand_fr_bytes(irnd, leaf_off[-1]))
        total = leaf_off[-1]
        self.bufs = [(e_.alloc(total * 64), e_.alloc(total * 128), ExtBuf(r.torch, GB * 384, r.dev)) for e_ in lanes]
        eng.sync()

    def submit(self, lane, g):
        E, e_ = self.E, self.lanes[lane]
        d1, d2, out = self.bufs[lane]
        n = g * self.B
        E.lsw_keygen_dev(e_, self.pk, n, g * self.leaves_per_batch, self.d_leaf_off, self.d_item_tree_leaf, self.d_item_tree_gate, self.dtt,
                         self.d_alpha, self.d_coef, self.d_item_coef_off, self.d_rand, d1, d2)
        if self.r.args.only_encrypt:
            return
        if os.environ.get("RABE_LSW_GENERAL_DECRYPT"):        # A/B: the general entry point (a ciphertext index per item)
            E.lsw_decrypt_dev(e_, n, self.max_pairs, g * self.pairs_per_batch, self.n_sel, self.d_pair_off, self.d_sel_start, self.d_sel_sk, self.d_sel_ct,
                              self.d_sel_z, self.d_ct_e1, self.d_ct_e2, self.d_ct_e1j, self.d_ct_attr_off, self.d_ct_idx, d1, d2, self.d_leaf_off, None,
                              self.e2_lines, out)
            return
        # every fresh key decrypts the one pre-made ciphertext: the one-ciphertext entry point (scaled ciphertext rows once per selection entry)
        E.lsw_decrypt_one_ct_dev(e_, n, self.max_pairs, g * self.pairs_per_batch, self.n_sel, self.d_pair_off, self.d_sel_start, self.d_sel_sk, self.d_sel_ct,
                                 self.d_sel_z, self.d_ct_e1, self.d_ct_e2, self.d_ct_e1j, d1, d2, self.d_leaf_off, None, self.e2_lines, out)

    def check(self, lane, g):
        n = g * self.B
        return self.bufs[lane][2].t[:n * 384].cpu().numpy().tobytes() == self.msg * n

    def out_tensor(self, lane):
        return self.bufs[lane][2].t

    def expected_first_slot_all_ranks(self):
        return self.msg * (self.r.world * self.B)           # every item decrypts the one pre-made ciphertext

    def describe(self):
        m = sum(len(s[0]) for s in self.sel) / len(self.sel)
        return {"workload": "LSW KP-ABE, keygen under a %d-leaf %s policy (%d distinct) + decrypt of a pre-made %d-attribute ciphertext, batch %d per GPU"
                            % (self.n_attr, self.r.args.tree, len(self.trees), self.n_attr, self.B),
                "attrs": self.n_attr, "policies": len(self.trees), "tree": self.r.args.tree, "ragged": bool(self.r.args.ragged), "pruned_leaves_avg": round(m, 2),
                "pairings_per_item": round(m + 1, 1), "reference_pairings_per_item": round(2 * m, 1), "prepared_ciphertext_e2": self.e2_lines is not None,
                "table_build_ms_per_public_key": round(self.table_build_ms, 1), "host_prep_ms_per_policy": round(self.host_prep_ms_per_policy, 3)}

    def algorithmic_fpmul_per_item(self):
        # keygen 200 fixed-base G1 + 200 fixed-base G2; dec 400 var-base G1 (1.12 MM: 200 scaled E1, 200 in the
        # sum of D1) + 201 Miller (1.6 MM) + 1 final exponentiation
        m = sum(len(s[0]) for s in self.sel) / len(self.sel)
        n = self.n_attr
        return {"k_miller_multi": (m + 1) * 8000.0, "k_lsw_dec_pairs": m * 2800.0, "k_msm_partial_g1": m * 2800.0, "k_final_exp": 9000.0,
                "k_table_mul_g2": n * 1056.0, "k_table_mul_g1": n * 352.0}

    def impl_fpmul_per_item(self):
        m = sum(len(s[0]) for s in self.sel) / len(self.sel)
        return {"k_miller_multi": (m + 1) * 5976.0, "k_lsw_dec_pairs": m * 2750.0, "k_msm_partial_g1": m * 1080.0}

    def algorithmic_bytes_per_step(self):
        return self.leaves_per_batch * (192 + 32 + 32) + self.B * (384 + 384) + self.pairs_per_batch * 4

    def cpu_baseline(self):
        from oracle import cport
        if not cport.available():
            return {"error": "oracle/c not built"}
        n = self.r.args.cpu_sample or 6
        dt = cport.lsw_keygen_dec(self.n_attr, n, tree=self.r.args.tree, seed=self.r.args.seed)
        return {"value": round(n / dt, 4), "unit": "ops/s", "cores": 1, "cpu_model": cpu_model(), "kind": "port",
                "sample": "%d LSW keygen+decrypt at %d leaves (%s tree) in %.1f s; the reference's operation order (lsw/mod.rs:121-170,228-290: "
                          "three G1 and one G2 binary double-and-add multiplications per leaf in keygen, two full pairings and a Gt::pow per leaf "
                          "in decrypt) over the C primitives of oracle/c/rabe_ref.c, single thread like the reference"
                          % (n, self.n_attr, self.r.args.tree, dt)}

    def close(self):
        if self.e2_lines:
            self.e2_lines.destroy()
        self.pk.destroy()


# ====================================================================================================================== AW11
class Aw11Bench:
    """config 5: aw11::encrypt + aw11::decrypt, 10 authorities x 20 attributes, a user key with all attributes."""
    metric = "ABE ops/sec (AW11 multi-authority CP-ABE encrypt+decrypt)"
    default_group = 4
    survey_fpmul_per_item = 9.5e6               # SURVEY.md 8d, config 5 restructured work
    launches_per_submit = {}

    def __init__(self, r):
        from rabe_amd import engine as E
        from rabe_amd import hostprep as hp
        self.r, self.E, self.hp = r, E, hp
        args, eng = r.args, r.eng
        self.B = args.batch or 1024
        self.n_attr = args.attrs or 200
        self.n_auth = 10
        R, le = hp.R_ORDER, hp.fr_le
        krnd = random.Random(args.seed * 1000003 + 5)

        def kfr():
            return krnd.randrange(1, R)
        g1 = eng.g1_mul([G1_GEN], [le(kfr())])[0]
        g2 = eng.g2_mul([G2_GEN], [le(kfr())])[0]
        egg = eng.pairing([g1], [g2])[0]
        per = self.n_attr // self.n_auth
        self.attrs = ["AUTH%dX%d" % (i // per, i % per) for i in range(self.n_attr)]          # upper-case, no '_' (aw11/mod.rs:137)
        alpha = [kfr() for _ in self.attrs]
        y = [kfr() for _ in self.attrs]
        te, t2, t1 = eng.gt_table(egg), eng.g2_table(g2), eng.g1_table(g1)
        egg_alpha = te.mul([le(a) for a in alpha])
        g2_y = t2.mul([le(v) for v in y])
        eng.sync()
        t0 = time.perf_counter()
        self.pk = E.Aw11Pk(eng, g1, g2, egg_alpha, g2_y)
        eng.sync()
        self.table_build_ms = 1e3 * (time.perf_counter() - t0)
        # the user's key (aw11::keygen, :165-231): K_x = g1*alpha_x + (g1*h(gid))*y_x = g1 * (alpha_x + h(gid) y_x)
        hg = hp.h_fr("alice")
        self.d_sk_k = eng.upload(b"".join(t1.mul([le(a + hg * v) for a, v in zip(alpha, y)])))
        self.d_sk_hash = eng.upload(t1.mul([le(hg)])[0])
        self.d_sk_attr_off = eng.upload_u32([0, self.n_attr])
        t1.destroy()
        t2.destroy()
        self.e_tab = te
        prnd = random.Random(args.seed)
        self.trees = []
        for _ in range(args.policies):
            names = list(self.attrs)
            prnd.shuffle(names)
            if args.ragged:                      # ciphertexts of mixed shapes: policies over 10 .. n_attr of the attributes
                k = prnd.randrange(10, self.n_attr + 1)
                names = names[:k - (k % 2)]
            self.trees.append(make_tree("nested" if args.tree == "flat" else args.tree, names, binary_and_only=True))
        t0 = time.perf_counter()
        self.tt = hp.TreeTables(self.trees)
        self.sel = []
        for t in self.trees:
            ok, idx = hp.pruned_leaf_indices(self.attrs, t)
            assert ok
            z = hp.leaf_coefficients(t)
            names = hp.flatten_tree(t)["names"]
            self.sel.append((idx, [self.attrs.index(names[y_]) for y_ in idx], [z[y_] for y_ in idx]))
        self.host_prep_ms_per_policy = 1e3 * (time.perf_counter() - t0) / len(self.trees)
        self.dtt = E.DevTreeTables(eng, self.tt)
        self.d_leaf_attr = eng.upload_u32([self.attrs.index(nm) for f in self.tt.flat for nm in f["names"]])

    def prepare(self, G, lanes):
        r, eng, E, hp, tt = self.r, self.r.eng, self.E, self.hp, self.tt
        R, le = hp.R_ORDER, hp.fr_le
        B, P = self.B, len(self.trees)
        GB = G * B
        self.G, self.lanes = G, lanes
        pol = [((i % B) * P // B) if r.args.ragged else i % P for i in range(GB)]
        sel_start_p, so, sel_ct, sel_sk, sel_z = [], 0, [], [], []
        for idx, ska, z in self.sel:
            sel_start_p.append(so)
            sel_ct += idx
            sel_sk += ska
            sel_z += z
            so += len(idx)
        self.n_sel = so
        row_off, pair_off, coef_off = [0], [0], [0]
        for p in pol:
            row_off.append(row_off[-1] + tt.n_leaves(p))
            pair_off.append(pair_off[-1] + len(self.sel[p][0]) + 1)
            coef_off.append(coef_off[-1] + 2 * tt.n_coef(p))
        assert all(row_off[(j + 1) * B] == (j + 1) * row_off[B] for j in range(G)), "batch must be a multiple of the policy count"
        self.rows_per_batch, self.pairs_per_batch = row_off[B], pair_off[B]
        self.max_pairs = max(len(s[0]) + 1 for s in self.sel)
        self.d_row_off = eng.upload_u32(row_off)
        self.d_pair_off = eng.upload_u32(pair_off)
        self.d_item_tree_leaf = eng.upload_u32([tt.first_leaf[p] for p in pol])
        self.d_item_tree_gate = eng.upload_u32([tt.first_gate[p] for p in pol])
        self.d_item_n_coef = eng.upload_u32([tt.n_coef(p) for p in pol])
        self.d_item_coef_off = eng.upload_u32(coef_off[:-1])
        self.d_sel_start = eng.upload_u32([sel_start_p[p] for p in pol])
        self.d_sel_ct, self.d_sel_sk = eng.upload_u32(sel_ct), eng.upload_u32(sel_sk)
        self.d_sel_z = eng.upload(b"".join(le(z) for z in sel_z))
        self.d_sk_idx = eng.upload_u32([0] * GB)
        irnd = random.Random(r.args.seed * 7919 + 23 + r.rank)
        self.d_s = eng.upload(b"".join(le(irnd.randrange(1, R)) for _ in range(GB)))
        self.rho_all, rho_grp = global_rho(r, B, G, R)
        rho = eng.upload(b"".join(le(x) for x in rho_grp))
        self.d_msg = eng.alloc(GB * 384)
        eng._check(eng.lib.rhip_gt_table_pow(eng.ctx, self.e_tab.h, E._sz(GB), rho.ptr, self.d_msg.ptr))
        self.d_coef = eng.upload(rand_fr_bytes(irnd, coef_off[-1]))
        self.d_rand = eng.upload(rand_fr_bytes(irnd, row_off[-1]))
        total = row_off[-1]
        self.bufs = [(e_.alloc(GB * 384), e_.alloc(total * 384), e_.alloc(total * 128), e_.alloc(total * 128), ExtBuf(r.torch, GB * 384, r.dev))
                     for e_ in lanes]
        eng.sync()

    def submit(self, lane, g):
        E, e_ = self.E, self.lanes[lane]
        c0, c1, c2, c3, out = self.bufs[lane]
        n = g * self.B
        E.aw11_encrypt_dev(e_, self.pk, n, g * self.rows_per_batch, self.d_row_off, self.d_item_tree_leaf, self.d_item_tree_gate, self.d_item_n_coef,
                           self.dtt, self.d_leaf_attr, self.d_s, self.d_coef, self.d_item_coef_off, self.d_rand, self.d_msg, c0, c1, c2, c3)
        if self.r.args.only_encrypt:
            return
        E.aw11_decrypt_dev(e_, n, self.max_pairs, g * self.pairs_per_batch, self.n_sel, self.d_pair_off, self.d_sel_start, self.d_sel_ct, self.d_sel_sk,
                           self.d_sel_z, c0, c1, c2, c3, self.d_row_off, self.d_sk_hash, self.d_sk_k, self.d_sk_attr_off, self.d_sk_idx, out)

    def check(self, lane, g):
        n = g * self.B * 384
        return self.bufs[lane][4].t[:n].cpu().numpy().tobytes() == self.r.eng.download(self.d_msg)[:n]

    def out_tensor(self, lane):
        return self.bufs[lane][4].t

    def expected_first_slot_all_ranks(self):
        eng, E = self.r.eng, self.E
        n = len(self.rho_all)
        d = eng.alloc(n * 384)
        eng._check(eng.lib.rhip_gt_table_pow(eng.ctx, self.e_tab.h, E._sz(n), eng.upload(b"".join(self.hp.fr_le(x) for x in self.rho_all)).ptr, d.ptr))
        return eng.download(d)

    def describe(self):
        m = sum(len(s[0]) for s in self.sel) / len(self.sel)
        tree = "nested" if self.r.args.tree == "flat" else self.r.args.tree
        return {"workload": "AW11 multi-authority CP-ABE, %d authorities x %d attributes, %s binary-AND policy over all %d (%d distinct), batch %d "
                            "encrypt+decrypt per GPU" % (self.n_auth, self.n_attr // self.n_auth, tree, self.n_attr, len(self.trees), self.B),
                "attrs": self.n_attr, "authorities": self.n_auth, "policies": len(self.trees), "tree": tree, "ragged": bool(self.r.args.ragged),
                "pruned_leaves_avg": round(m, 2),
                "pairings_per_item": round(m + 1, 1), "reference_pairings_per_item": round(2 * m + m + 1, 1),
                "table_build_ms_per_public_key_set": round(self.table_build_ms, 1), "host_prep_ms_per_policy": round(self.host_prep_ms_per_policy, 3),
                "attribute_tables": "16-bit windows for the %d per-attribute bases (%.0f GB, opted in with RABE_AW11_ATTR_W16=1) "
                                    "-- 16 instead of 32 table entries per power" % (2 * self.n_attr, self.n_attr * 16 * 65535 * 512 / 1e9)
                if os.environ.get("RABE_AW11_ATTR_W16", "0") == "1" else self.signed_tables_note()}

    def signed_tables_note(self):
        w = int(os.environ.get("RABE_AW11_ATTR_BITS", "10"))
        if not 9 <= w <= 14:
            return "8-bit windows for the per-attribute bases (0.8 GB)"
        n = (255 + w - 1) // w
        return ("signed %d-bit windows for the %d per-attribute bases (%d instead of 32 table entries per power; %.1f GB beside the 0.8 GB of 8-bit "
                "tables they are built from; the default is 10 bits)" % (w, 2 * self.n_attr, n, self.n_attr * n * (1 << (w - 1)) * 512 / 1e9))

    def algorithmic_fpmul_per_item(self):
        # enc 201 fixed-base Gt pow (0.34 MM) + 200 var-base Gt pow (1.6 MM) + 400 fixed-base + 200 var-base G2 (2.1 MM);
        # dec 200 var-base G2 (1.7 MM) + 200 var-base G1 (0.56 MM) + 201 Miller (1.6 MM) + 200 var-base Gt pow (1.6 MM) + 1 final exponentiation
        m = sum(len(s[0]) for s in self.sel) / len(self.sel)
        n = self.n_attr
        return {"k_miller_multi": (m + 1) * 8000.0, "k_aw11_dec_pairs": m * 2800.0, "k_msm_partial_g2": m * 8400.0, "k_gt_multiexp_partial": m * 8000.0,
                "k_final_exp": 9000.0, "k_aw11_enc_c1": n * (1700.0 + 8000.0), "k_aw11_enc_c3": n * (1056.0 + 8400.0), "k_table_mul_g2": n * 1056.0,
                "k_table_pow_gt_mul": 1700.0}

    def impl_fpmul_per_item(self):
        m = sum(len(s[0]) for s in self.sel) / len(self.sel)
        return {"k_miller_multi": (m + 1) * 5976.0}

    def algorithmic_bytes_per_step(self):
        return self.rows_per_batch * (384 + 256 + 32) * 2 + self.B * (32 + 384 * 3) + self.pairs_per_batch * 4

    def cpu_baseline(self):
        from oracle import cport
        if not cport.available():
            return {"error": "oracle/c not built"}
        n = self.r.args.cpu_sample or 4
        tree = "nested" if self.r.args.tree == "flat" else self.r.args.tree
        dt = cport.aw11_encdec(self.n_attr, n, tree=tree, seed=self.r.args.seed)
        return {"value": round(n / dt, 4), "unit": "ops/s", "cores": 1, "cpu_model": cpu_model(), "kind": "port",
                "sample": "%d AW11 encrypt+decrypt at %d attributes (%s tree) in %.1f s; the reference's operation order (aw11/mod.rs:241-289,298-366: "
                          "a pairing e(g1,g2) and two Gt::pow per row in encrypt, two full pairings and a Gt::pow per row in decrypt, binary "
                          "double-and-add everywhere) over the C primitives of oracle/c/rabe_ref.c, single thread like the reference"
                          % (n, self.n_attr, tree, dt)}

    def close(self):
        self.pk.destroy()
