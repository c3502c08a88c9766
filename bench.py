#!/usr/bin/env python3
"""bench.py -- BLS verifies/s (= BN254 pairings/s) of the MI355X-native engine, BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the whole verify hot path (hash-to-G1, G2 subgroup check, 2-pair Miller loop,
final exponentiation, bitmap) over one synthetic batch per GPU, inputs already resident in HBM
(PCIe-inclusive rate: DESIGN.md).  Workload at N = 1 is BASELINE.json configs[1]: 262144 batched
single-signature verifies on one MI355X; at N > 1 it is configs[3]: every rank gets its own 1048576
(8 M over 8 GPUs, weak scaling) and the ranks exchange only the validity bitmap (RCCL all-reduce).
Rank 0 prints ONE JSON line.

Launching: with WORLD_SIZE set (torch.distributed.run) this process is one rank and WORLD_SIZE must equal
--gpus.  With WORLD_SIZE unset and --gpus N > 1 this process starts the N ranks itself as child processes
(python -m torch.distributed.run ... bench.py ...) BEFORE anything touches the GPU, waits, and relays their
output and exit code -- it never re-executes a process that has initialised the GPU.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PER_GPU = 262144         # BASELINE configs[1]: one GPU
N_PER_GPU_MULTI = 1048576  # BASELINE configs[3]: 8 M tuples over 8 GPUs
UNIQ = 64                 # cpu_baseline sample only: distinct tuples signed by the CPU oracle, tiled
INVALID_EVERY = 64        # 1/64 of the tuples are corrupted
FP_MUL_MADS = 136         # 32x32->64 MADs of one 8x32-bit-limb Montgomery multiplication (2n^2 + n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tuples-per-gpu", dest="n", type=int, default=0,
                    help="tuples per GPU (default: 262144 at --gpus 1 = configs[1], 1048576 at --gpus N > 1 = configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-paths", action="store_true",
                    help="skip the untimed exact_path / rlc_path legs (profiling runs: keeps rocprofv3's per-kernel averages to the timed path)")
    ap.add_argument("--key-pool", dest="pool", type=int, default=1024,
                    help="distinct public keys in the batch (SURVEY.md 8d: 1024); 0 = every tuple has its own key (the all-distinct case)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--all-on-device0", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench: --gpus must be >= 1")
    if args.n <= 0:
        args.n = N_PER_GPU if args.gpus == 1 else N_PER_GPU_MULTI

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)
    run_rank(args)


def spawn_ranks(args):
    """Parent of an N-rank run: no torch.cuda / HIP call happens in this process."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    # The host driver of this pool only supports dmabuf IPC: with the legacy IPC mode RCCL's intra-node transport setup
    # (and any CUDA-tensor sharing across processes) fails with `hipIpcGetMemHandle: invalid argument`.  The image exports
    # this already; it is pinned here so that the ranks get it whatever shell started the bench.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)      # children inherit stdout / stderr: rank 0's JSON line goes straight through
    sys.exit(rc)


def run_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench: WORLD_SIZE=%d but --gpus %d (launch with --nproc-per-node equal to --gpus)" % (world, args.gpus))
    local_rank = 0 if args.all_on_device0 else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import blsbn254_loader
    M = blsbn254_loader.load()
    sharded = __import__("bls_bn254_amd.sharded", fromlist=["x"])
    from oracle import oracle as O          # data generation + cpu_baseline leg only (never the measured path)
    from tests import synth

    eng = M.Engine(local_rank)
    dst = M.DEFAULT_DST
    n = args.n
    n_total = n * world
    lo = rank * n

    # synthetic batch: n UNIQUE tuples per rank (global indices lo..lo+n), key pool of 1024, signed by the
    # engine's own GPU signing kernels and spot-checked at 1000 random indices against the CPU oracle; 1/64 corrupted
    pool = args.pool if args.pool > 0 else n
    t_gen = time.perf_counter()
    pks, msgs, sigs, exp = synth.make_batch_gpu(eng, O, n, dst, pool=pool, invalid_every=INVALID_EVERY, spot=1000, base=lo)
    data, off = M.engine.pack_messages(msgs)
    t_pk = torch.frombuffer(bytearray(pks), dtype=torch.uint8).to(dev)
    t_sg = torch.frombuffer(bytearray(sigs), dtype=torch.uint8).to(dev)
    t_ms = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    t_bm = torch.zeros((n + 7) // 8, dtype=torch.uint8, device=dev)
    del pks, sigs, data, msgs
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    if rank == 0:
        print("bench: rank 0 generated and staged %d tuples in %.1f s (GPU signing + oracle spot checks; outside the timed region)" % (n, t_gen),
              file=sys.stderr, flush=True)
    ar_ms = []                                # per step: the bitmap exchange alone (N > 1)

    def step():
        eng.verify_batch_dev(t_pk.data_ptr(), t_ms.data_ptr(), t_off.data_ptr(), t_sg.data_ptr(), n, t_bm.data_ptr(), dst)
        eng.synchronize()                     # the engine runs on its own stream
        if world > 1:
            ta = time.perf_counter()
            w = sharded.allreduce_bitmap(t_bm, lo, n, n_total, dist, torch)
            torch.cuda.synchronize()
            ar_ms.append((time.perf_counter() - ta) * 1e3)
            return w
        return None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # correctness of what is being timed: bitmap equals the closed-form expectation
    step()
    torch.cuda.synchronize()
    ok = bytes(t_bm.cpu().numpy()) == synth.bitmap_of(exp)
    if not ok:
        raise SystemExit("bench: GPU bitmap differs from the expected pattern")

    eng.profile_enable(True)
    eng.profile_reset()
    barrier()
    t0 = time.perf_counter()
    words = None
    del ar_ms[:]
    for _ in range(args.steps):
        words = step()
    dt_own = time.perf_counter() - t0         # this rank's own K steps, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # per-rank spread (a straggler would show here) and the cost of the exchange itself
        ar_mean = sum(ar_ms) / max(len(ar_ms), 1)
        lo_t = torch.tensor([dt_own, ar_mean, t_gen], dtype=torch.float64, device=dev)
        hi_t = lo_t.clone()
        dist.all_reduce(lo_t, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_t, op=dist.ReduceOp.MAX)
        rank_spread = {"ms_per_step_min": round(float(lo_t[0]) / args.steps * 1e3, 3), "ms_per_step_max": round(float(hi_t[0]) / args.steps * 1e3, 3),
                       "allreduce_ms_min": round(float(lo_t[1]), 3), "allreduce_ms_max": round(float(hi_t[1]), 3),
                       "datagen_s_max": round(float(hi_t[2]), 1),
                       "note": "per rank: its own K steps before the closing barrier (verify + bitmap exchange), and the exchange alone "
                               "(all_reduce of %d int32 words + synchronize); datagen is outside the timed region" % ((n_total + 31) // 32)}
        # every rank holds the same full bitmap: check it against the closed-form expectation over all global indices
        full = sharded.words_to_bitmap_bytes(words.cpu().numpy(), n_total)
        if full != synth.bitmap_of(synth.expected_bits(n_total, INVALID_EVERY)):
            raise SystemExit("bench: all-reduced bitmap differs from the expected pattern")

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_total * args.steps / dt
        core = O.verify_core_counts()         # exact Fp-mul counts of the algorithmic unit (instrumented oracle)
        # dominant kernel: the 2-pair Miller loop.  Which one ran depends on the batch: few distinct public keys (this
        # workload: a 1024-key pool) -> every key is prepared once and the table-only loop k_miller_prepared runs
        # (algorithmic work: the shared squarings of f + two table pairs); mostly distinct keys -> k_miller_verify
        # (one variable pair with its point arithmetic + one table pair).
        prepared = "miller_prepared" in prof
        dom = "miller_prepared" if prepared else "miller_verify"
        alg_fp_mul = (core[4] + 2 * core[1]) if prepared else (core[0] + core[1])
        count_desc = ("shared squarings of f %d + two table pairs 2 x %d" % (core[4], core[1]) if prepared
                      else "variable pair %d + table pair %d" % (core[0], core[1]))
        mil = prof.get(dom, {"launches": 1, "total_ms": float("nan")})
        mil_ms = mil["total_ms"] / max(mil["launches"], 1)
        mil_mads = alg_fp_mul * FP_MUL_MADS * n
        achieved = mil_mads / (mil_ms * 1e-3) / 1e12
        probe = eng.valu_probe()
        peak = probe["mad_per_s"] / 1e12
        ceiling = probe["issue_ceiling_per_s"] / 1e12
        kern = {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in prof.items()}
        # HBM-side bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes of this build
        traffic, traffic_note = None, "no PMC profile committed for this batch size"
        for tname in ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath) and n == N_PER_GPU:
                tj = json.load(open(tpath))
                if "k_" + dom in tj.get("kernels", {}):
                    traffic = tj["kernels"]["k_" + dom]["hbm_bytes_per_launch"]
                    traffic_note = ("PMC FETCH_SIZE/WRITE_SIZE of the same kernel at the same batch size (profiles/%s): "
                                    "%.1f KB per tuple against ~0.6 KB algorithmic (sig 64 B, H 108 B, f out 432 B; line tables from cache)"
                                    % (tname, traffic / n / 1024.0))
                    break
        exec_mads, exec_ratio = None, None
        epath = os.path.join(ROOT, "profiles", "r03_executed_mads.json")
        if os.path.exists(epath):
            for pname, pe in json.load(open(epath)).get("phases", {}).items():
                if "(k_" + dom + ")" in pname:
                    exec_mads, exec_ratio = pe["executed_mads"], pe.get("executed_over_algorithmic")
        if world == 1:
            wl = "BASELINE configs[1]"
        else:
            wl = "BASELINE configs[3] shape: %d tuples sharded over %d ranks, bitmap all-reduce (%s)" % (
                n_total, world, "RCCL over xGMI" if args.backend == "nccl" else args.backend)
        out = {
            "metric": "BN254 pairings/sec (= BLS verifies/sec)", "value": round(value, 1), "unit": "verifies/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": "%d batched single-sig BLS verifies per GPU (%s), unique 32-byte messages, "
                                   "%s, 1/64 invalid tuples; key de-duplication + hash-to-G1 + per-key G2 checks and line tables "
                                   "+ 2-pair Miller loop + final exp" % (n, wl, ("%d-key pool" % pool) if args.pool > 0 else "every tuple its own public key"),
                       "tuples_per_gpu": n, "tuples_total": n_total, "key_pool": pool, "datagen_s": round(t_gen, 1),
                       "path": "prepared-keys (distinct public keys validated and turned into line tables once per batch, inside the timed step)"
                               if prepared else "exact per-tuple"},
            "roofline": {"bound": "valu", "kernel": "k_" + dom, "achieved": round(achieved, 4), "peak": round(peak, 3),
                         "unit": "T int-MAD/s", "frac": round(achieved / peak, 4),
                         "peak_issue_ceiling": round(ceiling, 3), "frac_issue_ceiling": round(achieved / ceiling, 4),
                         "peak_vop2_measured": round(probe["vop2_per_s"] / 1e12, 3),
                         "clock_ghz_under_probe": round(probe["clock_hz_mad"] / 1e9, 3),
                         "traffic": traffic, "traffic_note": traffic_note,
                         "executed_mads_per_tuple": exec_mads, "executed_over_algorithmic": exec_ratio,
                         "achieved_executed": None if exec_mads is None else round(exec_mads * n / (mil_ms * 1e-3) / 1e12, 4),
                         "note": "achieved = %d Fp-mul x %d MAD x %d tuples / %.3f ms (HIP events on the engine stream).  Counting "
                                 "convention: Fp-mul = the CPU oracle's textbook count for this kernel's 2-pair loop (squarings count 1; "
                                 "%s); MADs per Fp-mul = 136, the 8x32-bit-limb Montgomery product of SURVEY.md 8d.  executed_mads_per_tuple "
                                 "= what this kernel's loop really issues per tuple with 9x29-bit lazy limbs (162 per product, 243 per double "
                                 "product, 9 per linear-combination term), counted by running the device headers on the host with operation "
                                 "counters (scripts/executed_mads.py -> profiles/r03_executed_mads.json; data independent); achieved_executed "
                                 "is the same launch priced at that count.  peak = v_mad_u64_u32 rate measured in this run at 4 waves/SIMD "
                                 "(blsbn254_valu_probe); peak_issue_ceiling = CUs x 4 SIMDs x 16 lanes x the clock held under the probe (one "
                                 "VALU instruction per 4 cycles per SIMD, MI355X_MICROARCH.md).  The path is bound by VALU integer issue, not "
                                 "HBM (algorithmic traffic ~1.4 KB/verify) and not MFMA"
                                 % (alg_fp_mul, FP_MUL_MADS, n, mil_ms, count_desc)},
            "kernel_ms": kern,
            "algorithmic_fp_mul_per_verify": {"miller_variable_pair": core[0], "miller_fixed_pair_lines": core[1], "final_exp": core[2],
                                              "miller_shared_squarings": core[4], "dominant_kernel": alg_fp_mul},
        }
        if world > 1:
            out["ranks"] = rank_spread
        if world == 1 and prepared and not args.no_side_paths:
            # the same batch through the exact per-tuple path (what a batch of all-distinct keys takes), reported beside the
            # headline so that the number does not hinge on the workload's key pool; outside the timed region above
            eng.set_auto_prepare(False)
            step(); torch.cuda.synchronize()
            assert bytes(t_bm.cpu().numpy()) == synth.bitmap_of(exp), "exact path: bitmap differs"
            eng.profile_enable(True); eng.profile_reset()
            te = time.perf_counter()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            de = (time.perf_counter() - te) / 3
            pe = eng.profile_read(); eng.profile_enable(False)
            eng.set_auto_prepare(True)
            mv = pe.get("miller_verify", {"launches": 1, "total_ms": float("nan")})
            mv_ms = mv["total_ms"] / max(mv["launches"], 1)
            out["exact_path"] = {"value": round(n / de, 1), "unit": "verifies/s", "ms_per_step": round(de * 1e3, 3),
                                 "kernel_ms": {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in pe.items()},
                                 "roofline_frac_k_miller_verify": round((core[0] + core[1]) * FP_MUL_MADS * n / (mv_ms * 1e-3) / 1e12 / peak, 4),
                                 "note": "same batch with key de-duplication / preparation switched off: per-tuple G2 check + variable-Q Miller loop"}
        if world == 1 and prepared and not args.no_side_paths:
            # the same batch through random-linear-combination batch verification: per-key chunks of 16
            # tuples checked as one virtual tuple each, failed chunks re-verified exactly; same bitmap (2^-64 per chunk).
            # Reported beside the headline, never as `value`: a verify here is no longer one pairing.
            def rlc_step():
                eng.verify_batch_rlc_dev(t_pk.data_ptr(), t_ms.data_ptr(), t_off.data_ptr(), t_sg.data_ptr(), n, t_bm.data_ptr(), dst)
                eng.synchronize()
            t_bm.zero_()
            eng.set_rlc_key_round(True)             # fresh back-off state
            rlc_step(); torch.cuda.synchronize()
            assert bytes(t_bm.cpu().numpy()) == synth.bitmap_of(exp), "RLC path: bitmap differs"
            s0 = eng.rlc_stats()
            eng.profile_enable(True); eng.profile_reset()
            tr = time.perf_counter()
            for _ in range(3):
                rlc_step()
            torch.cuda.synchronize()
            dr = (time.perf_counter() - tr) / 3
            pr = eng.profile_read(); eng.profile_enable(False)
            s1 = eng.rlc_stats()
            out["rlc_path"] = {"value": round(n / dr, 1), "unit": "verifies/s", "ms_per_step": round(dr * 1e3, 3),
                               "chunks_per_step": (s1["chunks"] - s0["chunks"]) // 3,
                               "fallback_tuples_per_step": (s1["fallback_tuples"] - s0["fallback_tuples"]) // 3,
                               "key_rounds_that_decided_the_batch": "%d of %d" % (s1["key_rounds_passed"] - s0["key_rounds_passed"], s1["key_rounds"] - s0["key_rounds"]),
                               "kernel_ms_per_step": {k: round(v["total_ms"] / 3, 4) for k, v in pr.items()},
                               "note": "blsbn254_verify_batch_rlc_dev on the same batch, same bitmap: weighted points r_i sig_i, r_i H(msg_i), first ONE "
                                       "check per key over all its tuples (decides a batch without invalid signatures; not this one: 1/64 are invalid), then one "
                                       "table-only Miller loop + final exponentiation per key-sorted chunk of 16, exact re-verification of the "
                                       "eligible tuples of failed chunks.  This workload's invalid tuples (every 64th, key = index mod 1024) all "
                                       "fall on 16 of the 1024 keys, which keeps the fallback small; profiles/r02_rlc.json has the spread-out cases"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(O, synth, dst)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(O, synth, dst):
    """The CPU oracle (kind "port") timed on this box's host cores on a bounded sample of the same workload shape."""
    threads = min(os.cpu_count() or 1, 16)
    sample = 512 * threads          # ~20 CPU-seconds of oracle work (~1.2 s wall on 16 threads)
    spk, smsg, ssig, sexp = synth.make_batch(O, sample, dst, invalid_every=INVALID_EVERY, uniq=UNIQ)
    t1 = time.perf_counter()
    bm = O.verify_batch(spk, smsg, ssig, dst, nthreads=threads)
    cdt = time.perf_counter() - t1
    assert bm == synth.bitmap_of(sexp)
    res = {"value": round(sample / cdt, 1), "unit": "verifies/s", "cores": threads, "kind": "port",
           "sample": "%d tuples of the same workload shape (32-byte messages, 1/64 invalid), built from %d distinct "
                     "oracle-signed tuples over an 8-key pool and tiled (CPU signing is the slow part); C oracle, Montgomery "
                     "4x64 (oracle/bn254_oracle.c), %d host threads, %.1f s wall" % (sample, UNIQ, threads, cdt)}
    # second figure of SURVEY.md 8d: whole verifies TIMED in the reference's own arithmetic (canonical operands, wide
    # product, bit-serial const_rem_wide, schoolbook Fp2: fp.rs:404-407, fp2.rs:377-390) on a small sample
    if hasattr(O, "verify_batch_refstyle"):
        rs_n = 4 * threads
        t2 = time.perf_counter()
        bm2 = O.verify_batch_refstyle(spk[:128 * rs_n], smsg[:rs_n], ssig[:64 * rs_n], dst, nthreads=threads)
        rdt = time.perf_counter() - t2
        assert bm2 == synth.bitmap_of(sexp[:rs_n])
        res["reference_style"] = {"value": round(rs_n / rdt, 3), "unit": "verifies/s", "cores": threads, "kind": "port",
                                  "sample": "%d whole verifies (%d per thread) timed with every Fp multiplication done the reference's way: "
                                            "canonical 4x64 operands, 512-bit product, 259-round bit-serial reduction (crypto-bigint 0.5.5 "
                                            "const_rem_wide, fp.rs:404-407); same bitmap as the Montgomery run, %.1f s wall" % (rs_n, 4, rdt)}
    return res


if __name__ == "__main__":
    main()
