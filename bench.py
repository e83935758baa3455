#!/usr/bin/env python3
"""bench.py -- ABE ops/s on MI355X (BASELINE.json metric).  Default: config 2, AC17 CP-ABE encrypt+decrypt.

One step = one pass of the hot path over one batch: `--batch` (4096) independent
ac17::cp_encrypt + ac17::cp_decrypt group-arithmetic calls at `--attrs` (50) attributes, 16 distinct
random binary AND/OR policies x 256 items, one public key, one secret key holding all attributes
(SURVEY.md 8d config 2).  Inputs (randomness s0,s1 and the Gt message per item, policy tables, key
material, selection lists) are resident in HBM before the timed region; ciphertexts go
encrypt -> HBM -> decrypt without touching the host.

Steps are submitted in GROUPS: up to `--group` (16) steps' batches lie contiguously in HBM and go through the
engine as ONE launch set (16 x 4096 items: 3072 Miller waves = 3 full rounds of the chip's 1024 SIMDs, 1024
final-exponentiation waves = one round), so a single launch fills the chip and the groups follow one another on one
stream (`--inflight 1`, the default of configs 2-4; config 5, whose launches are smaller, keeps 2 groups in flight).  No
hardware-queue tuning is involved.

N > 1 (`--gpus N`): one process per GPU.  Launched by torch.distributed.run the ranks come from the environment;
started plainly (`python bench.py --gpus N`) the script spawns the N ranks itself.  The global batch (N x batch
items per step, one global randomness stream) is cut with rabe_amd.shard.shard_range; ranks are independent on the
timed path (weak scaling, no data-path collective); after the timed region the fixed-size result records are
gathered with ONE all_gather_into_tensor on device (RCCL) and rank 0 checks them against the unsharded order.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (integer-VALU bound: the path is modular
big-integer arithmetic, neither HBM- nor MFMA-bound -- DESIGN.md section 5), measured on a launch that fills the
chip; `cpu_baseline` times the oracle's reference-order restatement on the host CPU (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import random
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# No GPU_MAX_HW_QUEUES override by default (HIP's 4 hardware queues are enough for 2 groups in flight);
# --hw-queues N sets it for experiments and must act before the HIP runtime initialises.
for _i, _a in enumerate(sys.argv):
    if _a == "--hw-queues" and _i + 1 < len(sys.argv):
        os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[_i + 1]
    elif _a.startswith("--hw-queues="):
        os.environ["GPU_MAX_HW_QUEUES"] = _a.split("=", 1)[1]

from benchkit.lib import G1_GEN, G2_GEN, MAC_PER_FPMUL, ExtBuf, cpu_model, regions_summary, same_on_all_ranks, split_steps, timed_regions  # noqa: E402

# Algorithmic work, in Fp multiplications (1 Fp mul = 136 32x32 multiply-adds: 8-limb CIOS), per lane:
#   SURVEY.md 8d constants (the "algorithmic minimum" the roofline is priced against) and the
#   instrumented counts of this engine's own code (tests/count_muls.py).
SURVEY_MILLER_FPMUL = 8000          # "Miller loop (optimal ate, 65-bit loop) ~ 8 kM"
SURVEY_MIXED_ADD_FPMUL = 11
IMPL_MILLER_FPMUL = 8723            # tests/count_muls.py: miller_loop (NAF chain) with Jacobian P
IMPL_MILLER2_FPMUL = 11910          # tests/count_muls.py: miller_loop_pair_parked (A replays prepared lines, B Jacobian, merged lines), two pairings
IMPL_FINAL_EXP_FPMUL = 7553         # tests/count_muls.py: final_exponentiation_ws (width-3 NAF exponent chain)
IMPL_MILLER_MULTI6_FPMUL = 29931    # tests/count_muls.py: miller_loop_multi, an AC17 item's six pairs (3 prepared + 3 walking) on one accumulator
IMPL_MILLER_MULTI6_WALK_FPMUL = 37194   # the same with nothing prepared
# the reduced-radix kernels (engine_rr.hip, 9 x 29-bit limbs): multiply-add INSTRUCTIONS per lane, tests/count_muls.py (81 per schoolbook
# product, 81 per reduction; the line products are taken as dot products: more products, far fewer reductions and no carry instructions)
IMPL_RR_MILLER_MULTI6_MADS = 5050755     # an AC17 item's six pairs (3 prepared + 3 walking) on one accumulator; prepared lines carry a unit y-coefficient
IMPL_RR_MILLER_MULTI6_WALK_MADS = 6682014
IMPL_RR_FINAL_EXP_MADS = 1199772 + 600 * 136   # + the one inversion, which runs on the 8 x 32-bit core (~600 Fp multiplications)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configs[N-1]")
    ap.add_argument("--batch", type=int, default=0, help="items per step per GPU (0 = the config's: 4096 / 4096 / 2048 / 1024)")
    ap.add_argument("--attrs", type=int, default=0, help="attributes (0 = the config's: 50 / 100 / 200 / 200)")
    ap.add_argument("--policies", type=int, default=16)
    ap.add_argument("--tree", default="flat", choices=["flat", "nested", "mixed"],
                    help="configs 3-5: shape of the access tree (flat n-ary AND / balanced binary ANDs / AND over two-leaf ORs)")
    ap.add_argument("--ragged", action="store_true",
                    help="a batch of mixed shapes: every policy draws its leaf count from 10 .. --attrs (config 2: rows per ciphertext; configs 3, 4, 5: pairs per item)")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--group", type=int, default=16, help="steps submitted as ONE launch set (their batches are contiguous in HBM)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="groups in flight on separate HIP streams; 0 = the config's default (configs 2-4: 1 -- every launch fills the chip and "
                         "two groups' kernels only disturb each other, config 2: 1.13 M vs 1.02 M ops/s; config 5: 2, its launches are smaller)")
    ap.add_argument("--min-time", type=float, default=1.0,
                    help="the K-step timed region is repeated until this many seconds have been timed; every region times exactly --steps steps")
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for experiments (0 = leave the HIP default)")
    ap.add_argument("--pairing-mode", type=int, default=0, choices=[0, 1, 3, 6, 29, 58, 99],
                    help="0 auto, 1 one lane per pairing, 3 three cooperating lanes per pairing, 6 six lanes per accumulator, 29 reduced radix "
                         "(identical results); 99 cross-check of all families on every launch (not a measurement)")
    ap.add_argument("--g-window", type=int, default=20,
                    help="window width (bits) of the fixed-base table of g: 16 (67 MB, unsigned digits), or 17..27 signed digits "
                         "(20: 0.4 GB -- the default, a third of what the key's other tables take; 24: 5.4 GB; 26: 19 GB, reported as value_wide_tables)")
    ap.add_argument("--wide-window", type=int, default=26, help="g-window of the informational value_wide_tables leg (0 = skip the leg)")
    ap.add_argument("--no-tail-overlap", action="store_true",
                    help="run a remainder group (--steps not a multiple of --group) after the full groups on the same stream instead of first, "
                         "with the next group's encrypt kernels beside its final exponentiation")
    ap.add_argument("--tail-mode", choices=["pairing", "final-exp"], default="pairing",
                    help="what of a remainder group runs beside the next full group's encrypt kernels: 'pairing' = its Miller loops and its final "
                         "exponentiation (rhip_ctx_release_when_miller_resident); 'final-exp' = only its final exponentiation "
                         "(rhip_ctx_release_before_final_exp, round 3's form)")
    ap.add_argument("--no-single-batch", action="store_true", help="skip the informational single-batch legs (--group 1 submissions)")
    ap.add_argument("--no-configs-leg", action="store_true", help="skip the bounded runs of BASELINE configs 3-5 (N = 1 only)")
    ap.add_argument("--configs-min-time", type=float, default=0.3, help="timed seconds per config of the configs leg")
    ap.add_argument("--no-host-io-leg", action="store_true",
                    help="skip the informational second timed region in which every step also moves its inputs and outputs over PCIe")
    ap.add_argument("--only-encrypt", action="store_true", help="diagnostic: skip the decrypt half of every step (value is then not the metric)")
    ap.add_argument("--no-prepared-sk", action="store_true",
                    help="decrypt without the per-key prepared lines (6 independent Miller loops per item)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-object-api", action="store_true", help="skip the informational object-level (rabe_* C++ host layer) leg")
    ap.add_argument("--cpu-sample", type=int, default=0, help="items for the CPU baseline (0 = auto)")
    args = ap.parse_args()
    if args.inflight <= 0:
        args.inflight = 2 if args.config == 5 else 1
    return args


# ---------------------------------------------------------------------------------------------------------------- rank spawning
def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) and relay rank 0's line.
    On a box with fewer than N GPUs the ranks share devices (LOCAL_RANK modulo the device count) and rendezvous over
    gloo -- RCCL refuses two ranks on one device; that mode exists for functional checks of the N > 1 path."""
    import torch
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    assert n_dev > 0, "bench.py needs a GPU (the engine has no CPU fallback)"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r % n_dev), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        if n_dev < args.gpus:
            env["RABE_DIST_BACKEND"] = "gloo"
            env["RABE_SHARED_DEVICE"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    sys.exit(rc)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU fallback)"
    n_dev = torch.cuda.device_count()
    # RABE_FORCE_DIST=1: initialise the process group and run the collective path with ONE rank too (RCCL works with a single rank) --
    # the way a one-GPU box executes the nccl branch the 8-GPU run will take (tests/test_gpu_multirank.py)
    dist_on = world > 1 or bool(os.environ.get("RABE_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # nccl (= RCCL) with one rank per GPU.  With fewer GPUs than local ranks (a launcher started more ranks than the box has
        # devices: a functional check) the ranks share devices and rendezvous over gloo -- RCCL refuses two ranks on one device.
        shared = int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) > n_dev
        backend = os.environ.get("RABE_DIST_BACKEND", "gloo" if shared else "nccl")
        torch.cuda.set_device(local_rank % n_dev)
        if backend == "nccl":          # bind the group to this rank's GPU (no "guessing device ID" in barrier / collectives)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank % n_dev))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    if args.config != 2:
        from benchkit import schemes as bench_schemes
        return bench_schemes.run(args, world, rank, local_rank, dist_on)

    from rabe_amd import Engine
    from rabe_amd import engine as E
    from rabe_amd import hostprep as hp
    from rabe_amd import shard

    args.batch = args.batch or 4096
    args.attrs = args.attrs or 50
    eng = Engine(local_rank)
    n_cu, dev_name = eng.device_info()
    stream = torch.cuda.Stream(device=local_rank)
    eng.set_stream(stream.cuda_stream)
    eng.set_pairing_mode(args.pairing_mode)
    dev = torch.device("cuda", local_rank)

    krnd = random.Random(args.seed * 1000003)          # key material: the same on every rank (one public key, one secret key)
    R = hp.R_ORDER
    le = hp.fr_le

    def kfr():
        return krnd.randrange(1, R)

    # ---------------------------------------------------------------- key material (ac17::setup, :141-182) via Level E ops
    a = [kfr(), kfr()]
    b = [kfr(), kfr()]
    k = [kfr(), kfr(), kfr()]
    g = eng.g1_mul([G1_GEN], [le(kfr())])[0]
    h = eng.g2_mul([G2_GEN], [le(kfr())])[0]
    h_a = eng.g2_mul([h, h], [le(a[0]), le(a[1])]) + [h]
    g_k = eng.g1_mul([g] * 3, [le(x) for x in k])
    e_gh = eng.pairing([g], [h])[0]
    e_gh_ka = eng.gt_pow([e_gh] * 2, [le(k[i] * a[i] + k[2]) for i in range(2)])
    eng.sync()
    t_tab = time.perf_counter()
    pk = E.Ac17Pk(eng, g, h_a, e_gh_ka)
    if args.g_window > 16:
        pk.set_g_window(args.g_window)
    eng.sync()
    table_build_ms = 1e3 * (time.perf_counter() - t_tab)
    gw = args.g_window
    g_table_bytes = 64 * (16 * 65535 if gw <= 16 else sum(E.wide_count(gw, i) for i in range(E.wide_windows(gw))))
    table_bytes = g_table_bytes + 64 * 32 * 255 + (64 * 16 * 65535 if gw > 16 else 0) + 3 * 128 * (32 * 255 + 16 * 65535) + 2 * 384 * (32 * 255 + 16 * 65535)

    # ---------------------------------------------------------------- secret key with all attributes (ac17::cp_keygen, :191-264)
    attrs = ["a%d" % (i + 1) for i in range(args.attrs)]
    g_tab, h_tab = eng.g1_table(g), eng.g2_table(h)
    H, H01 = hp.ac17_keygen_tables(attrs)
    dk0, dk, dkp = eng.alloc(3 * 128), eng.alloc(len(attrs) * 3 * 64), eng.alloc(3 * 64)
    E.ac17_keygen_dev(eng, g_tab, h_tab, eng.upload(b"".join(g_k)), eng.upload(b"".join(le(pow(x, R - 2, R)) for x in a)),
                      eng.upload(b"".join(le(x) for x in b)), 1, len(attrs), eng.upload(H), eng.upload(H01),
                      eng.upload(le(kfr()) + le(kfr())), eng.upload(b"".join(le(kfr()) for _ in attrs)), eng.upload(le(kfr())),
                      dk0, dk, dkp)

    # ---------------------------------------------------------------- policies (host: parse/MSP/prune are string work)
    prnd = random.Random(args.seed)          # the same policies on every rank
    if args.ragged:                          # a batch of mixed shapes: every policy over 10 .. --attrs of the attributes
        trees = []
        for _ in range(args.policies):
            names = list(attrs)
            prnd.shuffle(names)
            trees.append(hp.random_binary_tree(names[:prnd.randrange(10, len(attrs) + 1)], prnd))
    else:
        trees = [hp.random_binary_tree(attrs, prnd) for _ in range(args.policies)]
    t_prep = time.perf_counter()
    tables, sels, pol_rows, nnz = [], [], [], []
    for t in trees:
        pi, A, z = hp.ac17_policy_table(t)
        ok, ct_sel, sk_sel = hp.ac17_decrypt_selection(attrs, pi, t)
        assert ok
        tables.append(A)
        sels.append((ct_sel, sk_sel))
        pol_rows.append(len(pi))
        nnz.append(z)
    host_prep_ms_per_policy = 1e3 * (time.perf_counter() - t_prep) / len(trees)
    A_off = [0]
    for r_ in pol_rows:
        A_off.append(A_off[-1] + r_)
    dA = eng.upload(b"".join(tables))

    # ---------------------------------------------------------------- the groups: up to G steps' batches contiguous in HBM
    B = args.batch
    sizes = split_steps(args.steps, max(1, min(args.group, args.steps)))
    if os.environ.get("RABE_BENCH_SPLIT"):           # experiments: explicit group sizes, e.g. 16,4
        sizes = [int(x) for x in os.environ["RABE_BENCH_SPLIT"].split(",")]
        assert sum(sizes) == args.steps
    G = max(sizes)
    GB = G * B
    # ragged: the items of one policy are contiguous inside every step's batch (a caller that batches mixed shapes groups them)
    item_pol = [((i % B) * args.policies // B) if args.ragged else i % args.policies for i in range(GB)]
    ct_row_off = [0]
    ct_sel_all, sk_sel_all, ct_sel_off, sk_sel_off = [], [], [0], [0]
    for i in range(GB):
        p_ = item_pol[i]
        ct_row_off.append(ct_row_off[-1] + pol_rows[p_])
        ct_sel_all += sels[p_][0]
        sk_sel_all += sels[p_][1]
        ct_sel_off.append(len(ct_sel_all))
        sk_sel_off.append(len(sk_sel_all))
    rows_per_batch = ct_row_off[B]
    assert all(ct_row_off[(j + 1) * B] == (j + 1) * rows_per_batch for j in range(G)), "batch must be a multiple of the policy count"
    sel_per_batch = ct_sel_off[B]
    d_item_A_off = eng.upload_u32([A_off[p_] for p_ in item_pol])
    d_ct_row_off = eng.upload_u32(ct_row_off)
    d_ct_sel, d_ct_sel_off = eng.upload_u32(ct_sel_all), eng.upload_u32(ct_sel_off)
    d_sk_sel, d_sk_sel_off = eng.upload_u32(sk_sel_all), eng.upload_u32(sk_sel_off)
    d_sk_idx = eng.upload_u32([0] * GB)
    d_sk_row_off = eng.upload_u32([0, len(attrs)])

    # per-item randomness of the GLOBAL batch (world x B items per step, one stream): this rank owns [lo, hi); step slot j
    # of a group uses the same per-step inputs shifted by j, so that the slots differ
    lo, hi = shard.shard_range(world * B, rank, world)
    assert hi - lo == B
    irnd = random.Random(args.seed * 7919 + 13)
    s_all = [irnd.randrange(1, R) for _ in range(world * B * 2)]
    rho_all = [irnd.randrange(1, R) for _ in range(world * B)]
    s_loc, rho_loc = s_all[2 * lo:2 * hi], rho_all[lo:hi]
    s_grp = [((s_loc[2 * (i % B) + j] + (i // B)) % R) or 1 for i in range(GB) for j in range(2)]
    rho_grp = [((rho_loc[i % B] + (i // B)) % R) or 1 for i in range(GB)]
    ds = eng.upload(b"".join(le(x) for x in s_grp))
    e_tab = eng.gt_table(e_gh)
    dmsg = eng.alloc(GB * 384)
    drho = eng.upload(b"".join(le(x) for x in rho_grp))
    eng._check(eng.lib.rhip_gt_table_pow(eng.ctx, e_tab.h, E._sz(GB), drho.ptr, dmsg.ptr))

    # `inflight` groups in flight: each lane = its own engine context, stream, scratch and output buffers; inputs and key
    # tables are shared, read-only
    S = max(1, min(args.inflight, len(sizes)))
    lanes_ctx = [eng]
    streams = [stream]
    for _ in range(S - 1):
        e2 = Engine(local_rank)
        st2 = torch.cuda.Stream(device=local_rank)
        e2.set_stream(st2.cuda_stream)
        e2.set_pairing_mode(args.pairing_mode)
        lanes_ctx.append(e2)
        streams.append(st2)
    # spare contexts (streams) for the remainder group and the single-batch legs, created here with the others: four in all
    pool = [(e_, st_) for e_, st_ in zip(lanes_ctx, streams)]
    while world == 1 and len(pool) < 4:
        e2 = Engine(local_rank)
        st2 = torch.cuda.Stream(device=local_rank)
        e2.set_stream(st2.cuda_stream)
        e2.set_pairing_mode(args.pairing_mode)
        pool.append((e2, st2))
    total_rows = ct_row_off[GB]
    bufs = [(e_.alloc(GB * 3 * 128), e_.alloc(total_rows * 3 * 64), e_.alloc(GB * 384), ExtBuf(torch, GB * 384, dev)) for e_ in lanes_ctx]
    launch_no = [0]

    # the key is loaded once: Miller-loop lines of k_0 (rhip_ac17_sk_prepare), reused by every decryption with it
    sk_lines = None if args.no_prepared_sk else E.Ac17SkLines(eng, 1, dk0)
    eng.sync()

    def submit(g_steps, lane=None, on=None, part=3):
        """one launch set over g_steps contiguous batches (on = (engine context, its buffers) of an extra lane); part: 1 = the
        encrypt half, 2 = the decrypt half, 3 = both"""
        i = launch_no[0] % S if lane is None else lane
        if part & 1:
            launch_no[0] += 1
        e_, (c0_, c_, cp_, out_) = on if on is not None else (lanes_ctx[i], bufs[i])
        n = g_steps * B
        if part & 1:
            E.ac17_encrypt_dev(e_, pk, n, dA, d_item_A_off, d_ct_row_off, g_steps * rows_per_batch, ds, dmsg, c0_, c_, cp_)
        if args.only_encrypt or not (part & 2):
            return
        if sk_lines is None:
            E.ac17_decrypt_dev(e_, n, c0_, c_, d_ct_row_off, cp_, dk0, dk, d_sk_row_off, dkp, d_sk_idx,
                               d_ct_sel, d_ct_sel_off, d_sk_sel, d_sk_sel_off, out_)
        else:
            E.ac17_decrypt_prepared_dev(e_, n, c0_, c_, d_ct_row_off, cp_, sk_lines, dk, d_sk_row_off, dkp, d_sk_idx,
                                        d_ct_sel, d_ct_sel_off, d_sk_sel, d_sk_sel_off, out_)

    # A remainder group (the driver's --steps 20 = 16 + 4) is a launch set whose final exponentiation -- one wave per item, 256
    # waves for 4 steps -- leaves three quarters of the SIMDs idle for as long as a full group's.  It goes FIRST, on its own
    # context, and the first full group's encrypt kernels run beside that final exponentiation (rhip_ctx_release_before_final_exp)
    # instead of behind it; every Miller kernel still has the chip to itself.
    tail = None
    if S == 1 and len(sizes) >= 2 and sizes[-1] < sizes[0] and not args.no_tail_overlap and not args.only_encrypt:
        if len(pool) > 1:
            e2, st2 = pool[1]
        else:
            e2 = Engine(local_rank)
            st2 = torch.cuda.Stream(device=local_rank)      # a plain stream: the release waits until the remainder's final-exponentiation
            e2.set_stream(st2.cuda_stream)                  # blocks are resident, so no stream priority is needed
            e2.set_pairing_mode(args.pairing_mode)
            pool.append((e2, st2))
        tb = sizes[-1] * B
        tail = (e2, (e2.alloc(tb * 3 * 128), e2.alloc(sizes[-1] * rows_per_batch * 3 * 64), e2.alloc(tb * 384), ExtBuf(torch, tb * 384, dev)), st2)

    def run_steps():
        launch_no[0] = 0
        if tail is not None:
            e2, b2, _ = tail
            e2.wait_for(eng)                                   # the previous region's work on the main stream is done
            submit(sizes[-1], on=(e2, b2), part=1)
            if args.tail_mode == "pairing":
                e2.release_when_miller_resident(eng)           # the main stream resumes when the tail's Miller blocks own their CUs
            else:
                e2.release_before_final_exp(eng)               # the main stream resumes when the tail's Miller loops are done
            submit(sizes[-1], on=(e2, b2), part=2)
            submit(sizes[0], lane=0, part=1)                   # beside the tail's pairings / final exponentiation
            eng.wait_for(e2)
            submit(sizes[0], lane=0, part=2)
            for g_ in sizes[1:-1]:
                submit(g_, lane=0)
            return
        for g_ in sizes:
            submit(g_)

    def sync_all():
        for e_ in lanes_ctx:
            e_.sync()
        if tail is not None:
            tail[0].sync()

    def barrier():
        if dist_on:
            dist.barrier()

    # ---------------------------------------------------------------- timed region(s): EXACTLY --steps steps each
    if args.warmup:
        for _ in range((args.warmup + args.steps - 1) // args.steps):
            run_steps()
    sync_all()
    torch.cuda.synchronize()
    prof_on = bool(os.environ.get("RABE_MILLER_PROF")) and hasattr(eng.lib, "rhip_debug_miller_prof")          # diagnostic build (tools/prof_miller.sh)
    if prof_on:
        pbuf = (ctypes.c_ulonglong * 8)()
        eng.lib.rhip_debug_miller_prof(eng.ctx, pbuf)
    regions = timed_regions(run_steps, sync_all, args.min_time)
    elapsed = sum(regions) / len(regions)
    if prof_on:
        eng.lib.rhip_debug_miller_prof(eng.ctx, pbuf)
        w = max(1, pbuf[6])
        names = ["squaring", "prepared: loads + scaling", "prepared: line products", "walking: loads + G2 step + store", "walking: line products", "whole loop"]
        print("k_miller_multi_rr regions, shader cycles per wave (mean over %d waves):" % w, file=sys.stderr)
        for k in range(6):
            print("   %-34s %12.0f  %5.1f %%" % (names[k], pbuf[k] / w, 100.0 * pbuf[k] / max(1, pbuf[5])), file=sys.stderr)

    # ---------------------------------------------------------------- size-independent correctness property on the FULL batch:
    # decrypt(encrypt(msg)) == msg, bit for bit, for every item of every group slot (oracle parity at small sizes is in tests/)
    want = eng.download(dmsg)
    used = {}
    for j, g_ in enumerate(sizes):
        used[j % S] = g_                # the LAST group a lane ran is what its buffer holds
    if tail is not None:
        used = {0: sizes[-2] if len(sizes) > 1 else sizes[0]}
    ok = all(bufs[i][3].t[:g_ * B * 384].cpu().numpy().tobytes() == want[:g_ * B * 384] for i, g_ in used.items())
    if tail is not None:
        ok = ok and tail[1][3].t[:sizes[-1] * B * 384].cpu().numpy().tobytes() == want[:sizes[-1] * B * 384]
    gather = None
    if dist_on:
        f = torch.tensor([1 if ok else 0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        ok = bool(f.item())
        # the trivial gather: every rank's first-slot result records (B x 384 B) in ONE all_gather_into_tensor on device;
        # rank 0 recomputes the unsharded batch's messages from the global randomness stream and compares in global order
        mine = bufs[0][3].t[:B * 384]
        src = mine if dist.get_backend() == "nccl" else mine.cpu()
        allrec = torch.empty(world * B * 384, dtype=torch.uint8, device=src.device)
        torch.cuda.synchronize()
        tg0 = time.perf_counter()
        dist.all_gather_into_tensor(allrec, src)
        torch.cuda.synchronize()
        tg = time.perf_counter() - tg0
        if rank == 0:
            dall = eng.alloc(world * B * 384)
            dr_all = eng.upload(b"".join(le(x) for x in rho_all))
            eng._check(eng.lib.rhip_gt_table_pow(eng.ctx, e_tab.h, E._sz(world * B), dr_all.ptr, dall.ptr))
            gather = {"collective": "all_gather_into_tensor", "backend": dist.get_backend(), "bytes_per_rank": B * 384,
                      "ms": round(1e3 * tg, 3), "matches_unsharded_order": allrec.cpu().numpy().tobytes() == eng.download(dall)}

    value = world * B * args.steps / elapsed
    result = {
        "metric": "ABE ops/sec (AC17 CP-ABE encrypt+decrypt)", "value": round(value, 2), "unit": "ops/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "dtype_note": "exact BN254 integers in 32-bit limbs (Montgomery: 8 x 32 bits; 9 signed x 29 bits in the pairing kernels of full launches)",
        "data": "synthetic", "roundtrip_bit_exact": ok,
        "timed_regions": regions_summary(regions, args.steps),
        "config": {"workload": "AC17 CP-ABE, %d-attribute random binary AND/OR MSP policies (%d distinct), batch %d encrypt+decrypt per GPU"
                               % (args.attrs, args.policies, B),
                   "batch_per_gpu": B, "attrs": args.attrs, "policies": args.policies, "rows": rows_per_batch // B,
                   "pruned_leaves_avg": round(sel_per_batch / B, 2), "msp_nnz_avg": round(sum(nnz) / len(nnz), 1),
                   "steps_per_launch_set": sizes, "launch_sets_in_flight": S, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default"),
                   "remainder_group": (None if tail is None else
                                       "first, on its own stream; the next group's encrypt kernels run beside its final exponentiation" if args.tail_mode != "pairing" else
                                       "first, on its own stream; the next group's encrypt kernels run beside its Miller loops (on the CUs those leave) and "
                                       "its final exponentiation"),
                   "pairing_mode": args.pairing_mode,
                   "value_definition": "n_gpus x batch_per_gpu x steps / (max over ranks of the barrier-to-barrier time of the K steps): every rank runs the same "
                                       "per-GPU batch whatever n_gpus is (weak scaling), so value / n_gpus is directly comparable across rank counts",
                   "parallelism": "batch-sharded x%d (no data-path collective)" % world, "device": dev_name},
        "tables": {"bytes_per_public_key": table_bytes, "g_window_bits": gw, "g_table_bytes": g_table_bytes,
                   "build_ms_per_public_key": round(table_build_ms, 1),
                   "break_even_ops": round(table_build_ms * 1e-3 * value / max(world, 1)),
                   "note": "built once per public key, untimed; break_even_ops = encrypt+decrypt cycles that take as long as the build at the measured rate; "
                           "every rank builds its own replica (about as long as a broadcast of the same bytes over xGMI would take)"},
        "host_prep_ms_per_policy": round(host_prep_ms_per_policy, 3),
    }
    if gather is not None:
        result["gather"] = gather

    # ---------------------------------------------------------------- informational: ONE batch of 4096 per launch set (BASELINE config 2
    # words the workload as "batch of 4096"): its latency alone, and the rate of a stream of single-batch submissions with up to
    # four of them in flight on separate HIP streams (what a server that receives batches one at a time sees)
    if world == 1 and not args.no_single_batch and not args.only_encrypt:
        # four lanes = the four contexts / HIP streams of the pool (the main context's and the remainder group's included): HIP's default
        # of four hardware queues, one each
        extra = [(e2, (e2.alloc(B * 3 * 128), e2.alloc(rows_per_batch * 3 * 64), e2.alloc(B * 384), ExtBuf(torch, B * 384, dev)), st2)
                 for e2, st2 in pool[:4]]
        for e2, b2, _ in extra:
            submit(1, on=(e2, b2))
            e2.sync()
        torch.cuda.synchronize()
        reps = 6
        t0 = time.perf_counter()
        for _ in range(reps):
            submit(1, on=(extra[0][0], extra[0][1]))
            extra[0][0].sync()
        lat = (time.perf_counter() - t0) / reps
        single = {"batch": B, "latency_ms": round(1e3 * lat, 3), "ops_per_s_alone": round(B / lat, 1)}
        for k_in in (2, 4):
            # the same machinery as the headline: exactly-16-submission regions repeated for >= 0.3 s (what `bench.py --group 1
            # --inflight 4 --steps 16` measures)
            def run_k():
                for j in range(16):
                    e2, b2, _ = extra[j % k_in]
                    submit(1, on=(e2, b2))

            def sync_k():
                for e2, _, _ in extra[:k_in]:
                    e2.sync()
            run_k()
            sync_k()
            reg_k = timed_regions(run_k, sync_k, 0.3)
            single["ops_per_s_%d_in_flight" % k_in] = round(16 * B / (sum(reg_k) / len(reg_k)), 1)
        single["roundtrip_bit_exact"] = all(b2[3].t[:B * 384].cpu().numpy().tobytes() == want[:B * 384] for _, b2, _ in extra)
        single["fraction_of_grouped_rate"] = round(single["ops_per_s_4_in_flight"] / value, 3)
        single["note"] = ("one step per launch set (--group 1): a lone batch leaves most SIMDs idle, so the engine takes the six-lane kernels for it by itself "
                          "(k_miller_c6, k_final_exp_c6, k_gt_table_pow_c6: one Fq12 accumulator over six lanes, the same bytes; docs/coop6.md) and runs the three "
                          "independent encrypt kernels side by side; round 4 (one lane per item): 16.7 ms / 245 k ops/s alone. Batches in flight on separate "
                          "streams overlap each other's chains")
        result["single_batch"] = single

    # ---------------------------------------------------------------- informational: the same steps with host buffers
    # (inputs s, msg uploaded; ciphertext c_0, c, c_p and the decrypted Gt downloaded; pinned memory, copies on a copy
    # stream per group in flight, ordered by events).  Never `value`.
    if not args.no_host_io_leg and not args.only_encrypt and not os.environ.get("RABE_SHARED_DEVICE"):
        n_s, n_msg = 2 * GB * 32, GB * 384
        sizes_out = (GB * 3 * 128, total_rows * 3 * 64, GB * 384, GB * 384)
        io = []
        for e_ in lanes_ctx:
            h_in = (e_.host_alloc(n_s), e_.host_alloc(n_msg))
            h_out = tuple(e_.host_alloc(z) for z in sizes_out)
            io.append((e_.alloc(n_s), e_.alloc(n_msg), h_in, h_out))
        hs, hm = eng.download(ds), eng.download(dmsg)
        for (_, _, h_in, _) in io:
            ctypes.memmove(h_in[0], hs, n_s)
            ctypes.memmove(h_in[1], hm, n_msg)
        copy_ctx = []
        for _ in lanes_ctx:
            c2 = Engine(local_rank)
            st3 = torch.cuda.Stream(device=local_rank)
            c2.set_stream(st3.cuda_stream)
            copy_ctx.append((c2, st3))

        def submit_io(g_steps):
            i = launch_no[0] % S
            launch_no[0] += 1
            e_ = lanes_ctx[i]
            cpy = copy_ctx[i][0]
            c0_, c_, cp_, out_ = bufs[i]
            ds_, dmsg_, h_in, h_out = io[i]
            n = g_steps * B
            e_.wait_for(cpy)                               # the previous round's downloads of these buffers are done
            e_.upload_async(ds_, h_in[0], 2 * n * 32)
            e_.upload_async(dmsg_, h_in[1], n * 384)
            E.ac17_encrypt_dev(e_, pk, n, dA, d_item_A_off, d_ct_row_off, g_steps * rows_per_batch, ds_, dmsg_, c0_, c_, cp_)
            cpy.wait_for(e_)
            cpy.download_async(h_out[0], c0_, n * 3 * 128)
            cpy.download_async(h_out[1], c_, g_steps * rows_per_batch * 3 * 64)
            cpy.download_async(h_out[2], cp_, n * 384)
            if sk_lines is None:
                E.ac17_decrypt_dev(e_, n, c0_, c_, d_ct_row_off, cp_, dk0, dk, d_sk_row_off, dkp, d_sk_idx,
                                   d_ct_sel, d_ct_sel_off, d_sk_sel, d_sk_sel_off, out_)
            else:
                E.ac17_decrypt_prepared_dev(e_, n, c0_, c_, d_ct_row_off, cp_, sk_lines, dk, d_sk_row_off, dkp, d_sk_idx,
                                            d_ct_sel, d_ct_sel_off, d_sk_sel, d_sk_sel_off, out_)
            cpy.wait_for(e_)
            cpy.download_async(h_out[3], out_, n * 384)

        def run_io():
            launch_no[0] = 0
            for g_ in sizes:
                submit_io(g_)
            sync_all()
            for c2, _ in copy_ctx:
                c2.sync()

        run_io()
        barrier()
        t0 = time.perf_counter()
        reps_io = 0
        while same_on_all_ranks(reps_io < 1 or (time.perf_counter() - t0 < 0.5 * args.min_time and reps_io < 50)):
            run_io()
            reps_io += 1
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        el_io = shard.max_over_ranks(t1 - t0) / reps_io
        ok_io = all(ctypes.string_at(io[i][3][3], g_ * B * 384) == want[:g_ * B * 384] for i, g_ in used.items())
        per_step = (2 * 32 + 384 + 3 * 128 + 384 + 384) * B + rows_per_batch * 3 * 64
        result["host_io_leg"] = {"ops_per_s": round(world * B * args.steps / el_io, 2), "ms_per_step": round(1e3 * el_io / args.steps, 3),
                                 "pcie_bytes_per_step": per_step, "pcie_GBps": round(per_step * args.steps / el_io / 1e9, 2),
                                 "roundtrip_bit_exact": ok_io,
                                 "note": "inputs uploaded and all outputs downloaded every step (pinned host memory; downloads on a copy stream per group in flight, ordered by events)"}
        for i, e_ in enumerate(lanes_ctx):
            for hp_ in io[i][2] + io[i][3]:
                e_.host_free(hp_)
        for c2, _ in copy_ctx:
            c2.close()

    if rank == 0:
        # ------------------------------------------------------------ roofline of the dominant kernel: HIP events on the launch
        # stream around every kernel of ONE full group (G x B items, nothing else running) -- the launch fills the chip
        eng.timing(True)
        eng.timing_read()
        reps = 3
        for _ in range(reps):
            submit(G, lane=0)
            eng.sync()
        tim = eng.timing_read()
        eng.timing(False)
        per_kernel = {kname: ms / cnt for kname, (ms, cnt) in tim.items()}
        dom = max(per_kernel, key=lambda kk: per_kernel[kk])
        dom_ms = per_kernel[dom]
        # peak: dependent-free v_mad_u64_u32 issue rate measured live on this chip
        ms_c, ops_c = min((eng.calibrate(0, 20000) for _ in range(2)), key=lambda r_: r_[0])          # best of two: the first may see clocks ramping
        peak_tmac = ops_c / (ms_c * 1e-3) / 1e12
        m_avg = sel_per_batch / B
        NB = G * B
        rows_g = G * rows_per_batch
        # lanes of a launch and the algorithmic Fp-muls per lane (SURVEY 8d constants); the shared-accumulator decrypt path
        # (k_ac17_dec_pairs + k_miller_multi) carries all six Miller loops of an item in one k_miller_multi unit
        lanes = {"k_ac17_dec_miller": NB * 6, "k_ac17_dec_miller2": NB * 3, "k_miller_multi": NB, "k_ac17_dec_pairs": NB * 6, "k_final_exp": NB,
                 "k_ac17_dec_miller_c3": NB * 6, "k_final_exp_c3": NB, "k_ac17_enc_rows": rows_g, "k_ac17_enc_c0": NB * 3, "k_ac17_enc_cp": NB}
        alg = {"k_ac17_dec_miller": SURVEY_MILLER_FPMUL + m_avg * SURVEY_MIXED_ADD_FPMUL,
               "k_ac17_dec_miller2": 2 * (SURVEY_MILLER_FPMUL + m_avg * SURVEY_MIXED_ADD_FPMUL),
               "k_miller_multi": 6 * SURVEY_MILLER_FPMUL, "k_ac17_dec_pairs": (m_avg + 0.5) * SURVEY_MIXED_ADD_FPMUL,
               "k_ac17_dec_miller_c3": SURVEY_MILLER_FPMUL + m_avg * SURVEY_MIXED_ADD_FPMUL, "k_final_exp_c3": 9000 + 6 * 54,
               "k_final_exp": 9000 + 6 * 54, "k_ac17_enc_rows": 3 * 352, "k_ac17_enc_c0": 1056, "k_ac17_enc_cp": 2 * 1700 + 54}
        impl = {"k_ac17_dec_miller": IMPL_MILLER_FPMUL + m_avg * 11, "k_ac17_dec_miller2": IMPL_MILLER2_FPMUL + 2 * m_avg * 11, "k_final_exp": IMPL_FINAL_EXP_FPMUL + 6 * 54,
                "k_miller_multi": IMPL_MILLER_MULTI6_FPMUL if sk_lines is not None else IMPL_MILLER_MULTI6_WALK_FPMUL,
                "k_ac17_dec_miller_c3": IMPL_MILLER_FPMUL + m_avg * 11, "k_final_exp_c3": IMPL_FINAL_EXP_FPMUL + 6 * 54}
        # the reduced-radix kernels compute the same units of work (engine_rr.hip); what they EXECUTE is counted in multiply-add instructions
        lanes["k_miller_multi_rr"], lanes["k_final_exp_rr"] = lanes["k_miller_multi"], lanes["k_final_exp"]
        alg["k_miller_multi_rr"], alg["k_final_exp_rr"] = alg["k_miller_multi"], alg["k_final_exp"]
        impl["k_miller_multi_rr"] = (IMPL_RR_MILLER_MULTI6_MADS if sk_lines is not None else IMPL_RR_MILLER_MULTI6_WALK_MADS) / MAC_PER_FPMUL
        impl["k_final_exp_rr"] = (IMPL_RR_FINAL_EXP_MADS + 6 * 54 * 162) / MAC_PER_FPMUL
        equiv_8x32 = {"k_miller_multi_rr": impl["k_miller_multi"], "k_final_exp_rr": impl["k_final_exp"]}.get(dom)
        macs = lanes.get(dom, 0) * alg.get(dom, 0) * MAC_PER_FPMUL
        achieved = macs / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        macs_impl = lanes.get(dom, 0) * impl.get(dom, alg.get(dom, 0)) * MAC_PER_FPMUL
        achieved_impl = macs_impl / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        # whole step against the same peak: SURVEY 8d's 0.12 M Fp-mul per op (16.4 M MAC32)
        step_macs = 0.12e6 * MAC_PER_FPMUL * B
        # HBM side (reported, not binding): algorithmic bytes of the whole step
        alg_bytes = B * (2 * 32 + 384) + rows_per_batch * 192 * 2 + B * (384 + 384) * 2 + sel_per_batch * 4 * 2 + B * 384
        traffic, traffic_src = pmc_traffic(dom)
        # ALL VALU instructions the kernel issues per second (SQ_INSTS_VALU of the committed counter run x 64 lanes / this run's kernel time)
        # against the same ceiling: the chip issues ~0.54 G wave-instructions per second and SIMD whatever the mix (round 6: at a higher
        # VALU density the clock drops -- docs/tried.md "two waves per SIMD"), so this is how close the kernel is to the machine; `frac`
        # is the share of that which is multiply-adds
        vi = pmc_valu_issue(dom)
        frac_issue = None
        if vi and dom_ms > 0 and peak_tmac:
            # (the committed counters are per launch of tools/pmc_sq.sh's full group: 16 steps x 4096 items; this run's launch has G x B)
            frac_issue = round(vi["valu_insts_per_launch"] * (G * B / 65536.0) * 64.0 / (dom_ms * 1e-3) / (peak_tmac * 1e12), 4)
        result["roofline"] = {
            "bound": "valu_int (v_mad_u64_u32 issue rate; not hbm, not mfma)", "kernel": dom,
            "kernel_ms": round(dom_ms, 4), "items_per_launch": NB, "steps_per_launch": G,
            "kernel_ms_per_step": round(dom_ms / G, 4),
            "achieved": round(achieved_impl, 4), "peak": round(peak_tmac, 3), "unit": "TMAC32/s",
            "frac": round(achieved_impl / peak_tmac, 4) if peak_tmac else None,
            "work": ("multiply-add instructions this kernel issues per lane (instrumented: tests/count_muls.py; 9 x 29-bit limbs: 81 per product, 81 per "
                     "reduction) x lanes = %.3e per launch" if dom.endswith("_rr") else
                     "Fp multiplications this engine executes per lane (instrumented: tests/count_muls.py) x 136 MAC32 x lanes = %.3e MAC32 per launch")
                    % macs_impl,
            "frac_8x32_work": (round(lanes.get(dom, 0) * equiv_8x32 * MAC_PER_FPMUL / (dom_ms * 1e-3) / 1e12 / peak_tmac, 4) if (equiv_8x32 and peak_tmac and dom_ms > 0) else None),
            "frac_8x32_work_note": "the 8 x 32-bit kernel's instrumented work (Fp multiplications x 136) over THIS kernel's time: comparable with earlier rounds' `frac`",
            "achieved_survey": round(achieved, 4),
            "frac_survey": round(achieved / peak_tmac, 4) if peak_tmac else None,
            "work_survey": "SURVEY.md 8d's algorithmic constants (8 k Fp-mul per Miller loop) x 136 MAC32 x lanes = %.3e MAC32 per launch; the engine "
                           "executes fewer multiplications than that (shared squarings, merged and prepared lines), so frac_survey overstates the "
                           "utilisation of the multiplier -- `frac` is the honest figure" % macs,
            "whole_step_frac": round(step_macs / (elapsed / args.steps) / 1e12 / peak_tmac, 4) if peak_tmac else None,
            "traffic": traffic, "traffic_unit": "bytes of HBM fetch + write per launch of the dominant kernel (PMC FETCH_SIZE + WRITE_SIZE, separate passes)",
            "traffic_source": traffic_src,
            "valu_issue": vi, "frac_valu_issue": frac_issue,
            "hbm": {"algorithmic_bytes_per_step": alg_bytes, "GBps_at_measured_step": round(alg_bytes / (elapsed / args.steps) / 1e9, 3),
                    "peak_GBps": 8000},
            "kernels_ms": {kk: round(v, 4) for kk, v in sorted(per_kernel.items(), key=lambda x: -x[1])},
            "kernels_ms_sum_per_step": round(sum(per_kernel.values()) / G, 4),
            "dominant_by": "duration of one launch over a full group (every launch occupies all SIMDs); HIP events on the launch stream, nothing else running",
        }
        # ------------------------------------------------------------ the same config through the object-level host layer (rabe_*)
        if world == 1 and not args.no_object_api:
            try:
                result["object_api"] = object_api_leg(args, trees)
            except Exception as ex:
                result["object_api"] = {"error": repr(ex)}
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
        # ------------------------------------------------------------ informational: the same steps with the WIDE table of g (memory for work)
        if world == 1 and args.wide_window > 16 and args.wide_window != args.g_window and not args.only_encrypt:
            try:
                eng.sync()
                t_w = time.perf_counter()
                pk.set_g_window(args.wide_window)
                eng.sync()
                wide_build_ms = 1e3 * (time.perf_counter() - t_w)
                run_steps()
                sync_all()
                reg_w = timed_regions(run_steps, sync_all, min(0.4, args.min_time))
                el_w = sum(reg_w) / len(reg_w)
                v_w = B * args.steps / el_w
                wb = 64 * sum(E.wide_count(args.wide_window, i) for i in range(E.wide_windows(args.wide_window)))
                result["value_wide_tables"] = {
                    "value": round(v_w, 2), "unit": "ops/s", "ms_per_step": round(1e3 * el_w / args.steps, 4), "g_window_bits": args.wide_window,
                    "g_table_bytes": wb, "build_ms_per_public_key": round(wide_build_ms, 1), "break_even_ops": round(wide_build_ms * 1e-3 * v_w),
                    "speedup_over_value": round(v_w / value, 4),
                    "note": "same timed region with a %d-bit signed-window table of g (%.1f GB per public key instead of %.2f GB): %d instead of %d mixed "
                            "additions per fixed-base multiplication; not the headline because it makes the number single-tenant"
                            % (args.wide_window, wb / 1e9, g_table_bytes / 1e9, E.wide_windows(args.wide_window),
                               16 if gw <= 16 else E.wide_windows(gw))}
            except Exception as ex:
                result["value_wide_tables"] = {"error": repr(ex)}
        # ------------------------------------------------------------ BASELINE configs 3-5, bounded (their own processes: fresh tables, fresh memory)
        if world == 1 and not args.no_configs_leg and not args.only_encrypt:
            pk.destroy()
            pk = None
            result["configs"] = configs_leg(args)
        # the headline's honest neighbours, where a reader of the (truncated) record sees them: at the top level AND inside the objects the
        # driver's record keeps whole (`config`, `cpu_baseline`)
        sb = result.get("single_batch") if isinstance(result.get("single_batch"), dict) else {}
        oa = result.get("object_api") if isinstance(result.get("object_api"), dict) else {}
        pk_leg = oa.get("packed") if isinstance(oa.get("packed"), dict) else {}
        thr = oa.get("threads") if isinstance(oa.get("threads"), dict) else {}
        result["value_lone_batch"] = sb.get("ops_per_s_alone")
        result["value_end_to_end"] = pk_leg.get("ops_per_s")
        if2 = oa.get("packed_inflight2") if isinstance(oa.get("packed_inflight2"), dict) else {}
        if if2.get("ops_per_s") and if2.get("plaintexts_match"):
            result["value_end_to_end_inflight2"] = if2["ops_per_s"]
        result["config"]["neighbours"] = {
            "value_lone_batch": sb.get("ops_per_s_alone"), "lone_batch_latency_ms": sb.get("latency_ms"),
            "value_end_to_end": pk_leg.get("ops_per_s"), "end_to_end_batch": pk_leg.get("batch"),
            "value_blocking_64_threads": (thr.get("blocking") or {}).get("ops_per_s") if isinstance(thr.get("blocking"), dict) else None,
            "note": "`value` fuses %d steps into one launch set with inputs resident in HBM; value_lone_batch = ONE %d-item batch at a time (--group 1), "
                    "value_end_to_end = rabe_ac17_cp_{encrypt,decrypt}_packed with policy parse, MSP, hashing, pruning, membership checks, KDF + AES-GCM and the "
                    "PCIe copies inside the timed region, value_blocking_64_threads = one blocking call per ciphertext from 64 host threads" % (max(sizes), B)}
        if isinstance(result.get("cpu_baseline"), dict) and isinstance(result.get("configs"), dict):
            result["cpu_baseline"]["configs"] = {k: (v.get("cpu_baseline") if isinstance(v, dict) else None) for k, v in result["configs"].items()
                                                 if k in ("3", "4", "5")}
        from benchkit.lib import emit_line
        emit_line(result)

    if pk is not None:
        pk.destroy()
    for e2, _ in pool[len(lanes_ctx):]:
        e2.close()
    for e_ in lanes_ctx[1:]:
        e_.close()
    eng.close()
    if dist_on:
        dist.destroy_process_group()


def configs_leg(args):
    """`python bench.py --config K` for K = 3, 4, 5 as bounded sub-runs (a few steps, --configs-min-time seconds timed each, default --
    i.e. small -- AW11 attribute tables), so that the driver's one default run carries every BASELINE configuration.  Each
    entry is that run's own JSON line reduced to the fields a reviewer checks."""
    out = {}
    env = dict(os.environ)
    for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k_, None)
    env["RABE_BENCH_FULL_LINE"] = "1"          # the sub-runs hand their whole object over the pipe (benchkit.lib.emit_line)
    # (key, config, steps, extra flags, items of the CPU sample): "3_mixed" is SURVEY 8d's 50 %-OR variant of config 3, the "_ragged" legs draw
    # every policy's leaf count from 10 .. the config's (a batch of mixed shapes)
    legs = (("2_ragged", 2, 16, ["--ragged", "--no-single-batch", "--no-configs-leg", "--no-host-io-leg", "--wide-window", "0"], 0),
            ("3", 3, 8, [], 5), ("3_mixed", 3, 8, ["--tree", "mixed"], 0), ("3_ragged", 3, 8, ["--ragged"], 0), ("4", 4, 8, [], 2), ("4_ragged", 4, 8, ["--ragged", "--no-object-api"], 0), ("5", 5, 8, [], 3), ("5_ragged", 5, 8, ["--ragged", "--no-object-api"], 0))
    for key, cfg, steps, extra, cpu_n in legs:
        cmd = [sys.executable, os.path.abspath(__file__), "--config", str(cfg), "--gpus", "1", "--steps", str(steps), "--warmup", str(steps),
               "--min-time", str(args.configs_min_time), "--seed", str(args.seed), "--no-object-api"] + extra
        cmd += ["--cpu-sample", str(cpu_n)] if cpu_n and not args.no_cpu_baseline else ["--no-cpu-baseline"]
        t0 = time.perf_counter()
        try:
            pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            line = pr.stdout.decode().strip().splitlines()[-1] if pr.stdout.strip() else ""
            d = json.loads(line)
            rf = d.get("roofline", {})
            out[key] = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                             "pairings_per_item": d["config"].get("pairings_per_item"), "cpu_baseline": d.get("cpu_baseline"),
                             "batch_per_gpu": d["config"]["batch_per_gpu"], "workload": d["config"]["workload"],
                             "steps_per_launch_set": d["config"]["steps_per_launch_set"], "roundtrip_bit_exact": d["roundtrip_bit_exact"],
                             "timed_regions": {"count": d["timed_regions"]["count"], "ms_mean": d["timed_regions"]["ms_mean"]},
                             "roofline": {"kernel": rf.get("kernel"), "kernel_ms": rf.get("kernel_ms"), "frac": rf.get("frac"),
                                          "frac_survey": rf.get("frac_survey"), "peak": rf.get("peak"), "kernels_ms": rf.get("kernels_ms")},
                             "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as ex:
            out[key] = {"error": repr(ex)[:300]}
    return out


def object_api_leg(args, trees):
    """AC17 config 2 through the reference-shaped API of the C++ host layer: policy strings and plaintext bytes in, canonical ciphertext
    records out (AES-GCM sealing included) and back to plaintext bytes -- what a caller of the scheme functions sees; parse, MSP, pruning
    on the host, KDF + AES-GCM and record assembly on the device, the PCIe copies: all inside the timed region.
    `packed` / `packed_full_group`: rabe_ac17_cp_{encrypt,decrypt}_packed (one blob + offsets per call) at 5 and 16 steps' worth of items;
    `threads`: the one-call API from many threads through the submission queue; `objects`: one handle per ciphertext, one batch call."""
    import numpy as np
    from rabe_amd import hostlib as hl
    from rabe_amd import hostprep as hp
    from rabe_amd.schemes import ac17
    host = hl.Host(0)
    try:
        attrs = ["a%d" % (i + 1) for i in range(args.attrs)]
        pk, msk = ac17.setup(host)
        sk = ac17.cp_keygen(host, msk, attrs)
        pols = [hp.to_json(t) for t in trees]

        def packed_leg(n, reps, host=host):
            pts = [b"dance like no one's watching, encrypt like everyone is!" + i.to_bytes(4, "little") for i in range(n)]
            item_pol = np.arange(n, dtype=np.uint32) % len(pols)
            pt_blob = b"".join(pts)
            pt_off = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.uint64)
            pt_np = np.frombuffer(pt_blob, dtype=np.uint8)
            # caller-allocated, re-used buffers (the first round also builds the tables and the pinned staging buffers: untimed)
            ct_buf, _ = ac17.cp_encrypt_packed(host, pk, pols, item_pol, pt_np, pt_off)
            ct_buf = np.empty(ct_buf.size, dtype=np.uint8)
            ct_buf[:] = 0
            pt_buf = np.zeros(ct_buf.size, dtype=np.uint8)
            best = None
            best_tr = None
            ok_packed = True
            for rep in range(reps):
                t0 = time.perf_counter()
                ct_blob, ct_off = ac17.cp_encrypt_packed(host, pk, pols, item_pol, pt_np, pt_off, out=ct_buf)
                t1 = time.perf_counter()
                out_blob, out_off, status = ac17.cp_decrypt_packed(host, sk, ct_blob, ct_off, out=pt_buf)          # checked: membership pass on
                t2 = time.perf_counter()
                ok_packed = ok_packed and out_blob.tobytes() == pt_blob and not status.any() and (out_off == pt_off).all()
                out_blob, out_off, status = ac17.cp_decrypt_packed(host, sk, ct_blob, ct_off, out=pt_buf, trusted=True)
                t3 = time.perf_counter()
                ok_packed = ok_packed and out_blob.tobytes() == pt_blob and not status.any()
                if rep and (best is None or t2 - t0 < best[0]):
                    best = (t2 - t0, t1 - t0, t2 - t1)
                if rep and (best_tr is None or t3 - t2 < best_tr):
                    best_tr = t3 - t2
            return {"ops_per_s": round(n / best[0], 1), "encrypt_s": round(best[1], 4), "decrypt_s": round(best[2], 4), "batch": n,
                    "decrypt_trusted_s": round(best_tr, 4), "ops_per_s_trusted": round(n / (best[1] + best_tr), 1),
                    "ciphertext_bytes": int(ct_blob.size), "plaintexts_match": bool(ok_packed)}
        # one packed call carries five steps' worth of items (a single 4096-item launch under-fills the chip) ...
        packed = packed_leg(5 * args.batch, 4)
        packed["note"] = ("decrypt_s includes the batched group-membership pass over every decoded element (c_0 in G2's r-torsion, rows on the G1 "
                          "curve, c_p in Gt's order-r subgroup, coordinates < p): the default for external ciphertexts; *_trusted skips it "
                          "(RABE_PACKED_TRUSTED). KDF + AES-256-GCM and record assembly run on the device (round 4)")
        # ... and sixteen: the size of the device-level launch groups, where the decrypt's final exponentiation fills the chip
        packed_full = packed_leg(16 * args.batch, 3)
        # ... and the same sixteen steps' worth through a device GROUP (include/rabe_host.h: rabe_host_open_group): every visible GPU, or --
        # on a one-GPU box -- two engines on GPU 0, i.e. the call cut into two blocks that run side by side (one block's copies and host
        # stages beside the other's kernels)
        packed_group = None
        try:
            import torch
            n_dev = torch.cuda.device_count()
            devs = list(range(n_dev)) if n_dev > 1 else [0, 0]
            ghost = hl.Host(devices=devs)
            try:
                packed_group = packed_leg(16 * args.batch, 3, host=ghost)
                packed_group["devices"] = devs
            finally:
                ghost.close()
        except Exception as ex:          # the leg is informational
            packed_group = {"error": repr(ex)[:300]}
        # ... and TWO packed calls in flight: two host handles (own engine, stream, staging buffers and table replicas each), one caller thread
        # per handle, every thread encrypts and decrypts (checked) its own 16 steps' worth of items over and over -- one call's parsing, copies
        # and AES beside the other's group arithmetic (tools/bench_packed_inflight.py: 2 x 32 768 items 672 k, 2 x 65 536 856 k ops/s).
        try:
            runs = [packed_inflight_leg(hl, ac17, pk, sk, pols, 16 * args.batch, 2, 4) for _ in range(2)]          # two caller threads: the rate
            packed_inflight2 = max(runs, key=lambda r_: r_["ops_per_s"])          # depends on how their calls interleave (0.68-0.86 M run to run): best of 2
            packed_inflight2["ops_per_s_runs"] = [r_["ops_per_s"] for r_ in runs]
            packed_inflight2["plaintexts_match"] = all(r_["plaintexts_match"] for r_ in runs)
        except Exception as ex:          # informational
            packed_inflight2 = {"error": repr(ex)[:300]}
        n = args.batch
        pts = [b"dance like no one's watching, encrypt like everyone is!" + i.to_bytes(4, "little") for i in range(n)]
        # ---- one object handle per ciphertext (round 1's path), one step's worth
        items = [pols[i % len(pols)] for i in range(n)]
        ac17.cp_decrypt_batch(host, [sk] * 64, ac17.cp_encrypt_batch(host, pk, items[:64], pts[:64], hl.JSON_POLICY))
        t0 = time.perf_counter()
        cts = ac17.cp_encrypt_batch(host, pk, items, pts, hl.JSON_POLICY)
        t1 = time.perf_counter()
        out = ac17.cp_decrypt_batch(host, [sk] * n, cts)
        t2 = time.perf_counter()
        objects = {"ops_per_s": round(n / (t2 - t0), 1), "encrypt_s": round(t1 - t0, 4), "decrypt_s": round(t2 - t1, 4), "plaintexts_match": out == pts}
        del cts
        threads = threads_leg(host, pk, sk, pols)
        return {"ops_per_s": packed["ops_per_s"], "packed": packed, "packed_full_group": packed_full, "packed_group": packed_group,
                "packed_inflight2": packed_inflight2, "threads": threads,
                "objects": objects,
                "note": "policy text + plaintext bytes -> canonical ciphertext records -> plaintext bytes through the C ABI of the host layer; "
                        "parse/MSP/pruning/KDF/AES-GCM, record assembly and the PCIe copies are inside the timed region (best of 3 for `packed`)"}
    finally:
        host.close()


def packed_inflight_leg(hl, ac17, pk, sk, pols, n, handles, rounds):
    """`handles` host handles on GPU 0, one Python thread each (ctypes releases the GIL inside the C ABI), every thread runs `rounds` timed
    rounds of rabe_ac17_cp_encrypt_packed + rabe_ac17_cp_decrypt_packed (membership pass on) over its own n items after one untimed round;
    rate = all items of all timed rounds / (last end - first start)."""
    import threading
    import numpy as np
    hosts = [hl.Host(0) for _ in range(handles)]
    try:
        pts = [b"dance like no one's watching, encrypt like everyone is!" + i.to_bytes(4, "little") for i in range(n)]
        pt_blob = b"".join(pts)
        pt_off = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.uint64)
        pt_np = np.frombuffer(pt_blob, dtype=np.uint8)
        item_pol = np.arange(n, dtype=np.uint32) % len(pols)
        gate = threading.Barrier(handles)
        res = [None] * handles

        def worker(w):
            try:
                h = hosts[w]
                ct_buf, _ = ac17.cp_encrypt_packed(h, pk, pols, item_pol, pt_np, pt_off)
                ct_buf = np.empty(ct_buf.size, dtype=np.uint8)
                pt_buf = np.zeros(ct_buf.size, dtype=np.uint8)
                ct_blob, ct_off = ac17.cp_encrypt_packed(h, pk, pols, item_pol, pt_np, pt_off, out=ct_buf)          # untimed: tables, staging buffers
                ac17.cp_decrypt_packed(h, sk, ct_blob, ct_off, out=pt_buf)
                gate.wait()
                if w:
                    time.sleep(0.5 * res_hint[0])          # start half a call apart: one handle's copies beside the other's kernels
                t0 = time.perf_counter()
                ok = True
                for _ in range(rounds):
                    ct_blob, ct_off = ac17.cp_encrypt_packed(h, pk, pols, item_pol, pt_np, pt_off, out=ct_buf)
                    out_blob, out_off, status = ac17.cp_decrypt_packed(h, sk, ct_blob, ct_off, out=pt_buf)
                    ok = ok and not status.any() and int(out_off[n]) == len(pt_blob)
                t1 = time.perf_counter()
                ok = ok and out_blob.tobytes() == pt_blob
                res[w] = (t0, t1, ok)
            except Exception as ex:
                res[w] = ex
                try:
                    gate.abort()
                except Exception:
                    pass
        # how long one call takes alone (for the half-call offset)
        h0 = hosts[0]
        ct0, off0 = ac17.cp_encrypt_packed(h0, pk, pols, item_pol, pt_np, pt_off)
        ac17.cp_decrypt_packed(h0, sk, ct0, off0)
        ta = time.perf_counter()
        ct0, off0 = ac17.cp_encrypt_packed(h0, pk, pols, item_pol, pt_np, pt_off)
        ac17.cp_decrypt_packed(h0, sk, ct0, off0)
        res_hint = [time.perf_counter() - ta]
        del ct0
        th = [threading.Thread(target=worker, args=(w,)) for w in range(handles)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for r in res:
            if isinstance(r, Exception):
                raise r
        wall = max(r[1] for r in res) - min(r[0] for r in res)
        return {"ops_per_s": round(handles * rounds * n / wall, 1), "handles": handles, "items_per_call": n, "rounds_per_handle": rounds,
                "wall_s": round(wall, 4), "one_call_alone_s": round(res_hint[0], 4), "ops_per_s_one_call_alone": round(n / res_hint[0], 1),
                "plaintexts_match": all(r[2] for r in res),
                "note": "two callers, one host handle each, every call = encrypt + checked decrypt of its items; copies and host stages of one call run "
                        "beside the other call's kernels"}
    finally:
        for h in hosts:
            h.close()


def threads_leg(host, pk, sk, pols):
    """The reference's API shape under load: host threads that issue rabe_ac17_cp_encrypt / rabe_ac17_cp_decrypt ONE CALL AT A TIME on one
    host handle, through the submission queue (include/rabe_host.h: concurrent calls are collected into packed batches).  `blocking`:
    every thread waits for each call (throughput = threads / latency of a launch set: what the API shape allows); `in_flight`: every
    thread keeps `depth` submitted calls in flight (rabe_*_submit / rabe_ticket_wait)."""
    import ctypes
    from rabe_amd import hostlib as hl
    host.set_coalescing(True, window_us=200)          # a batch stays open for 0.2 ms: the threads that were released together join it
    out = {"window_us": 200}
    try:
        arr, npol = hl._strs(pols)
        prev = [0] * 6
        for name, T, depth, seconds in (("blocking", 64, 1, 1.5), ("blocking_1024", 1024, 1, 1.5), ("in_flight", 64, 64, 2.0), ("in_flight_256", 64, 256, 2.5)):
            ops, bad = ctypes.c_uint64(), ctypes.c_uint64()
            t0 = time.perf_counter()
            hl._check(host.lib.rabe_bench_ac17_threads(host.h, pk.ptr, sk.ptr, arr, npol, hl.JSON_POLICY, ctypes.c_uint32(T), ctypes.c_uint32(depth),
                                                       ctypes.c_double(seconds), ctypes.byref(ops), ctypes.byref(bad)), host.h)
            dt = time.perf_counter() - t0
            st = (ctypes.c_uint64 * 6)()
            host.lib.rabe_host_queue_stats(host.h, st)
            d_ = [int(st[k]) - prev[k] for k in range(5)]
            prev = [int(st[k]) for k in range(6)]
            out[name] = {"threads": T, "calls_in_flight_per_thread": depth, "ops_per_s": round(ops.value / dt, 1), "ops": ops.value,
                         "seconds": round(dt, 2), "plaintexts_match": bad.value == 0,
                         "queue": {"batches": d_[0], "requests": d_[1], "packed_calls": d_[2], "ran_singly": d_[3],
                                   "ms_inside_batches": round(d_[4] / 1e3, 1), "requests_per_batch": round(d_[1] / max(d_[0], 1), 1)}}
    finally:
        host.set_coalescing(False)
    out["note"] = ("one call per ciphertext from native host threads on ONE host handle (rabe_bench_ac17_threads calls the public entry points); calls that "
                   "arrive while a batch runs are collected into the next packed batch (group commit). A blocking caller has one call in flight, so "
                   "`blocking*` is bounded by threads / (latency of an encrypt batch + a decrypt batch: ~6 ms each way round 5, ~20 ms round 4); `in_flight` is the same API "
                   "with 64 submitted calls per thread (rabe_*_submit / rabe_ticket_wait)")
    return out


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (tools/pmc_traffic.sh writes them; rocprofv3
    cannot run inside the bench).  None when no summary names the kernel."""
    import glob
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.txt")) if "_cfg" not in os.path.basename(f))   # config 2's sets
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


def pmc_valu_issue(kernel):
    """VALU issue-slot occupancy of `kernel` from the newest committed SQ-counter summary (tools/pmc_sq.sh): VALU
    instructions per wave issue slot (SQ_WAVE_CYCLES counts 4-cycle slots)."""
    import glob
    import re
    for f in reversed(sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_sq.txt")) if "_cfg" not in os.path.basename(f))):
        vals, cur = {}, None
        for line in open(f):
            if not line.startswith(" "):
                cur = line.strip()
                continue
            m = re.match(r"\s+(SQ_\w+)\s+([0-9.e+]+) per launch", line)
            if m and cur == kernel:
                vals[m.group(1)] = float(m.group(2))
        if "SQ_INSTS_VALU" in vals and vals.get("SQ_WAVE_CYCLES"):
            return {"valu_per_slot": round(vals["SQ_INSTS_VALU"] / vals["SQ_WAVE_CYCLES"], 3),
                    "valu_insts_per_launch": vals["SQ_INSTS_VALU"], "source": "profiles/" + os.path.basename(f)}
    return None


def cpu_baseline_multicore(args, tree):
    """The same C restatement on several host cores at once (independent processes, a few items each): what a
    multi-threaded caller of the single-threaded reference would get.  Informational, beside cpu_baseline.
    Plain subprocesses with a hard deadline -- nothing here can hold up the GPU result."""
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
    return {"value": round(n / busy, 3), "unit": "ops/s", "cores": len(done), "cpu_model": cpu_model(), "kind": "port",
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
        n = args.cpu_sample or 96
        _outs, dt = cport.ac17_encdec(policy, args.attrs, n, seed=args.seed)
        return {"value": round(n / dt, 4), "unit": "ops/s", "cores": 1, "cpu_model": cpu_model(), "kind": "port",
                "sample": "%d AC17 encrypt+decrypt (policy parse + MSP + group loops) at %d attributes in %.1f s; C restatement "
                          "of the reference's operation order (oracle/c/rabe_ref.c: binary double-and-add for every G*Fr, "
                          "per-row hash-to-group, one final exponentiation per pairing), single thread like the reference"
                          % (n, args.attrs, dt),
                "note": "includes the per-item host work (parse, MSP, hashing) that the GPU `value` keeps outside its timed region; "
                        "object_api is the GPU path at the same boundary"}
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
    return {"value": round(1.0 / est, 6), "unit": "ops/s", "cores": 1, "cpu_model": cpu_model(), "kind": "port",
            "sample": "pure-Python big-int oracle (reference operation order): 1 encrypt+decrypt at %d attributes measured "
                      "(%.1f s + %.1f s), encrypt scaled linearly to %d attributes" % (n_attr, t_enc, t_dec, args.attrs)}


if __name__ == "__main__":
    main()
