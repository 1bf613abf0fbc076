#!/usr/bin/env python3
"""Headline benchmark: FALKON fit + infer samples/sec (N=1e6, D=1024, M=1e4, 30 classes).

One "step" = one pass of the hot path over the whole synthetic job: for each of the C classes
fit a FALKON classifier on the N rows (Nystroem centres by the reference rule, f64
preconditioner, f32 K_nM build, 20 CG iterations on the stored K_nM) and score all N rows with
it.  The N rows are sharded contiguously over the ranks (one process per GPU); classes go in batches of
world-size, each rank owning (preconditioner + CG state of) one class of the batch; the batch advances in
lock step and each CG iteration exchanges one all-gather and one reduce-scatter of a (world, M) f64
matrix (RCCL).  N is the job size at every GPU count => "scaling": "strong".

    python bench.py [--gpus N --steps K --warmup W] [--rows 1000000 --dim 1024 --centres 10000 --classes 30]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: the parent touches no
GPU, runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process, relays rank 0's JSON line
and exits with the child's status (no retry).

Rank 0 prints ONE JSON line (contract in the round prompt).  Inputs are resident in HBM before
the timed region.  `roofline` is measured live with HIP events around the launches of the kernel family
with the most device time (`roofline_hbm` / `roofline_mfma`: both families; `roofline_step`: the whole step
against SURVEY 8(d)'s roofline time); `cpu_baseline` times the numpy oracle on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
F16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: Peak BF16/FP16 MFMA, dense
HBM_PEAK_GBS = 8000.0          # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", dest="n", type=int, default=1_000_000)
    ap.add_argument("--dim", dest="D", type=int, default=1024)
    ap.add_argument("--centres", dest="M", type=int, default=10_000)
    ap.add_argument("--classes", type=int, default=30)
    ap.add_argument("--sigma", type=float, default=15.0)
    ap.add_argument("--lam", type=float, default=1e-5)
    ap.add_argument("--maxiter", type=int, default=20)
    ap.add_argument("--warmup-classes", type=int, default=2, help="classes run per warm-up step")
    ap.add_argument("--precond-depth", type=int, default=0, help="batches of preconditioners in flight ahead of the fit (default 2)")
    ap.add_argument("--precond-batch", type=int, default=0, help="classes whose preconditioners one batched launch chain builds "
                    "(odx_falkon_precond_batched_f64), one group ahead of the fits; 1 = one chain per class (--precond-depth applies); "
                    "default: min(6, classes this rank owns)")
    ap.add_argument("--lockstep-batch", type=int, default=0, help="classes per lock-step batch (a divisor of the rank count; default: "
                    "planned from the HBM budget, odx/plan.py — the rank count when it fits); smaller batches rotate their owners "
                    "through the ranks")
    ap.add_argument("--cg-exchange", choices=("lockstep", "allreduce"), default="lockstep",
                    help="lockstep (default, what the headline times): batches of classes, one owner rank per class, one all-gather + "
                    "one reduce-scatter per CG iteration; allreduce: the north star's literal form — every rank holds every "
                    "class's preconditioner and M-sized state, ONE all-reduce of the (M,) partial per CG iteration")
    ap.add_argument("--precond-cus", type=int, default=0, help="confine the preconditioner chains (their stream and the library's "
                    "helper streams) to this many compute units, spread over the XCDs (0: the whole device; an experiment knob — "
                    "measured slower at 48..128 CUs: the confined chains starve behind the main stream's grids, docs/HISTORY.md 7)")
    ap.add_argument("--precond-lookahead", type=int, default=1, help="chain groups in flight ahead of the group being fitted (one more "
                    "factor block each)")
    ap.add_argument("--precond-cus-full-only", action="store_true", help="with --precond-cus k: only the full-size chain groups are "
                    "confined to k compute units (their f64 work then runs beside the HBM-bound passes); the ramp groups the first fits "
                    "wait for keep the whole chip")
    ap.add_argument("--main-priority", action="store_true", help="run the job's main stream (builds, passes, scoring, CG) on a HIGH-priority "
                    "stream: the chains' kernels are then dispatched where the main stream has no workgroup waiting — beside the "
                    "persistent passes, not beside the builds (an experiment; combine with --precond-lookahead 2)")
    ap.add_argument("--gauss-on-complement", action="store_true", help="with --precond-cus k: launch the K_nM builds and the scoring on a "
                    "stream confined to the other (all - k) compute units — chain and Gaussian workgroups then never share a CU (an "
                    "experiment: DESIGN.md section 9 has the sweep)")
    ap.add_argument("--reserve-cus", type=int, default=0, help="CUs the persistent pass kernel leaves to the side streams")
    ap.add_argument("--precond-behind-cg", dest="precond_after_fit", action="store_true",
                    help="issue the look-ahead preconditioner behind the batch's CG instead of before its fit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the rls / forward / detect / minibootstrap extra keys (N = 1 only)")
    ap.add_argument("--extras", choices=("child", "inprocess"), default="child", help="where the full set of extras runs: a fresh child "
                    "process (default; the latency-bound ones are ALSO measured in this process, key extras_in_headline_process) or this one")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="0 = pick by host core count")
    ap.add_argument("--check", action="store_true", help="verify one class against the oracle on a row sample")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) in production; gloo only for the "
                    "single-device debugging mode below")
    ap.add_argument("--emulate-world", type=int, default=0, help="run ONE rank's share of a job of this many ranks alone on this GPU "
                    "(its 1 / W of the rows for all classes, the chains of the classes it owns, collectives replaced by local copies "
                    "of the right size): the rank's compute-only step time — a prediction, not a measurement of W GPUs")
    ap.add_argument("--emulate-rank", type=int, default=0, help="which rank of --emulate-world to run")
    ap.add_argument("--single-device", action="store_true",
                    help="debugging: every rank uses cuda:0 (exercises the sharded path on a 1-GPU box)")
    return ap.parse_args()


def synth_rows(n_lo, n_hi, D, C, seed, device):
    """Rows [n_lo, n_hi) of the synthetic job: class blobs mu_c + 0.7 eps, normalised with the
    reference rule (x - mean) * 20 / mean_norm (OnlineRegionClassifier.py:224-227); row i has
    class id i % C (one-vs-rest labels, N/C positives per class).  Generated in blocks from
    per-block seeds so any shard can be produced independently and identically."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    mu = torch.randn(C, D, generator=g).to(device)
    X = torch.empty((n_hi - n_lo, D), dtype=torch.float32, device=device)
    blk = 65536
    for b0 in range((n_lo // blk) * blk, n_hi, blk):
        gb = torch.Generator(device=device).manual_seed(seed * 1000003 + b0 // blk)
        e = torch.randn((blk, D), generator=gb, device=device, dtype=torch.float32)
        ids = (torch.arange(b0, b0 + blk, device=device) % C)
        rows = mu[ids] + 0.7 * e
        lo, hi = max(b0, n_lo), min(b0 + blk, n_hi)
        X[lo - n_lo:hi - n_lo] = rows[lo - b0:hi - b0]
    # population statistics of this generator (mean of the blob centres; E|x| by sampling)
    mean = mu.mean(0)
    X -= mean
    # the blob spread and the noise make |x|^2 ~ |mu_c - mean|^2 + 0.49 D: use the exact expectation
    mean_norm = torch.sqrt(((mu - mean) ** 2).sum(1) + 0.49 * D).mean()
    X *= 20.0 / mean_norm
    return X


def centre_indices(N, C, M, seed):
    """Reference rule (FALKONWrapper.compute_indices_selection): <= M/2 positives first (sampled
    with replacement when there are more), negatives fill up to M.  Host-side, seeded, identical
    on every rank; positives of class c are the rows i with i % C == c."""
    out = []
    rng = np.random.default_rng(seed)
    for c in range(C):
        npos = (N - c + C - 1) // C
        if npos > M // 2:
            p = rng.integers(0, npos, M // 2) * C + c
        else:
            p = np.arange(npos) * C + c
        nneg = N - npos
        k = M - p.shape[0]
        if nneg > k:
            q = rng.integers(0, nneg, k)
        else:
            q = np.arange(nneg)
        # q-th row (in order) with r % C != c: block b = q // (C-1), p = q % (C-1), r = b C + p + (p >= c)
        neg = q + (q // (C - 1)) + ((q % (C - 1)) >= c) if C > 1 else q
        out.append(np.concatenate([p, neg]).astype(np.int64))
    return out


class Phase:
    """Accumulates device time of one kernel family with HIP events on the launch stream."""

    def __init__(self):
        self.pairs = []

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()
        return self

    def __exit__(self, *exc):
        self.b.record()
        self.pairs.append((self.a, self.b))

    def total_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs)

    def count(self):
        return len(self.pairs)

    def reset(self):
        self.pairs = []


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as ONE child process tree
    (python -m torch.distributed.run, one process per GPU) and relay rank 0's JSON line.  This parent never
    initialises the GPU (no HIP call before or after the child), never execs, never retries: the child's exit status
    is this process's exit status."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out                       # rank 0's result: printed last, alone on stdout
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc != 0:
        print("bench.py: the %d-rank child exited with status %d" % (args.gpus, rc), file=sys.stderr)
        sys.exit(rc if 0 < rc < 256 else 1)
    if line is None:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        sys.exit(1)
    print(line, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:       # under a launcher the launcher decides; the line reports what ran
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; running %d" % (args.gpus, world, world),
              file=sys.stderr)
    if world > 1:
        torch.cuda.set_device(0 if args.single_device else local_rank)
        kw = {}
        if args.dist_backend == "nccl":           # bind the RCCL communicator to this rank's GPU at creation (barriers included)
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group(backend=args.dist_backend, init_method="env://", **kw)
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", torch.cuda.current_device())
    # the rank count the result line reports is the one a real collective saw, not an argument echoed back
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, dtype=torch.float64, device=device)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(one.item())))
        if ranks_seen != dist.get_world_size():
            print("bench.py: all-reduce saw %d ranks, world size is %d" % (ranks_seen, dist.get_world_size()), file=sys.stderr)
            sys.exit(3)

    import odx
    from odx.backend import Features
    from odx.dist import RowShard
    from odx.job import LockstepClassJob
    from odx.solver import SolverOptions

    be = odx.get_backend()
    shard = RowShard()
    emulated = args.emulate_world > 1
    if emulated:
        if world > 1:
            print("bench.py: --emulate-world runs in one process", file=sys.stderr)
            sys.exit(2)
        from odx.dist import EmulatedShard
        shard = EmulatedShard(args.emulate_world, args.emulate_rank)
    N, D, M, C = args.n, args.D, args.M, args.classes
    lo, hi = shard.bounds(N)
    n_loc = hi - lo
    seed = 1234 + 3
    X = synth_rows(lo, hi, D, C, seed, device)
    F = None                                       # kernel operands of X (row norms, packed f16 split): derived inside every step
    row_ids = torch.arange(lo, hi, device=device)
    cidx = centre_indices(N, C, M, seed)
    if emulated:
        # the centres' rows live on all ranks; this process holds one rank's: every centre index is folded onto a row of THIS
        # shard with the same class id (row i is of class i % C), so that a class's centres keep their positives / negatives
        # make-up and the Gaussian blocks their statistics
        def fold(idx):
            cls = idx % C
            first = lo + ((cls - lo) % C)                          # first row of that class in [lo, hi)
            cnt = np.maximum((hi - first + C - 1) // C, 1)
            return first + C * ((idx // C) % cnt)
        cidx = [fold(i) for i in cidx]
        assert all(int(i.min()) >= lo and int(i.max()) < hi for i in cidx)
    cidx_dev = [torch.from_numpy(i).to(device) for i in cidx]     # inputs of the job: resident before the timed region
    opt = SolverOptions(check_pivots=False)       # no host sync inside the timed region: every status is read after it
    infos = []                                    # Cholesky status words of every preconditioner built in the timed region
    ph = {k: Phase() for k in ("knm", "ktk", "ktk2", "precond", "mmv")}

    # The schedule (odx/job.py): classes in lock-step batches of `world`, every rank owning one class of a batch; the
    # preconditioners of the classes a rank owns are built ahead on a side stream, by default class-batched (the
    # factorisation chain of ONE preconditioner is ~1500 dependent small launches that leave most of the chip idle;
    # odx_falkon_precond_batched_f64 advances G classes with the same chain: groups of 1, 2, 3, then G = 6 batches, group
    # g + 1 built while group g is fitted).  --precond-batch 1 = one chain per class, `depth` batches ahead, each on its own
    # side stream with its own output slot; measured on one GPU at the headline size, three runs each (s per step):
    #   issued before the batch's fit, 2 batches ahead                               8.68-8.71
    #   issued before the batch's fit, 3 batches ahead                               8.88
    #   issued behind the batch's CG, 2 / 3 batches ahead (--precond-behind-cg)      8.96 / 9.09
    #   issued before the fit, 1 batch ahead                                         8.97
    #   (helper streams shared by all chains, behind the CG, 3 ahead: 8.84-8.87; --reserve-cus 16 / 32 change < 1 %)
    be.reserve_cus_during_passes(args.reserve_cus)
    job = LockstepClassJob(be, X, N, M, lambda c: torch.where((row_ids % C) == c, 1.0, -1.0).to(torch.float64), cidx_dev,
                           args.sigma, args.lam, args.maxiter, opt, shard=shard, precond_batch=args.precond_batch,
                           precond_depth=args.precond_depth, precond_after_fit=args.precond_after_fit, precond_cus=args.precond_cus,
                           batch=args.lockstep_batch, exchange=args.cg_exchange, gauss_on_complement=args.gauss_on_complement,
                           precond_lookahead=args.precond_lookahead, precond_cus_full_only=args.precond_cus_full_only)
    G, ldk, scores = job.G, job.ldk, job.scores
    job_b, plan_gb = job.b, round(job.plan.total_bytes / 1e9, 1)
    kfmt = be.knm_format(n_loc, M)                 # storage of the K_nM shards ("u24" at the headline size, "f32" for small ones)

    main_stream = torch.cuda.Stream(priority=-1) if args.main_priority else None

    def run_classes(classes, timed):
        if main_stream is None:
            return job.run(F, classes, phases=ph if timed else None, infos=infos if timed else None)
        main_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(main_stream):
            r = job.run(F, classes, phases=ph if timed else None, infos=infos if timed else None)
        torch.cuda.current_stream().wait_stream(main_stream)
        return r

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (untimed): a few classes are enough to touch every kernel and allocation
    for _ in range(args.warmup):
        F = be.features(X)
        run_classes(list(range(min(max(args.warmup_classes, world), C))), False)
    barrier()

    # The same pass kernel with nothing else on the GPU (3 launches on the K_nM left by the warm-up): in the timed region
    # it deliberately shares the chip with the preconditioner stream, so its rate there is not the kernel's own.
    alone_gbps = None
    if args.warmup > 0 and n_loc > 0:
        Kw = be._knm_block(n_loc, M, kfmt, job.kbufs[0])      # the block the warm-up's last class left there
        vv, oo = torch.ones(M, dtype=torch.float64, device=device), torch.empty(M, dtype=torch.float64, device=device)
        be.ktk(Kw, v=vv, out=oo)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            be.ktk(Kw, v=vv, out=oo)
        e1.record()
        torch.cuda.synchronize()
        alone_gbps = 3 * float(be.knm_bytes(n_loc, M)) / (e0.elapsed_time(e1) * 1e-3) / 1e9
    barrier()

    # ---- timed region: exactly K steps
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        F = be.features(X)       # row norms now, the packed f16 split at the first Gaussian launch: derived data, inside the step
        last = run_classes(list(range(C)), True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    ms_per_step = dt * 1e3 / args.steps
    value = N * args.steps / dt

    # ---- health of what was timed (outside the timing): no failed Cholesky, every score finite, on every rank
    bad_pivots = int(sum(int(i.item() != 0) for i in infos))
    finite = bool(torch.isfinite(scores).all().item())
    hl = torch.tensor([bad_pivots, 0 if finite else 1], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(hl, op=dist.ReduceOp.SUM)
    health = {"failed_choleskys": int(hl[0].item()), "ranks_with_nonfinite_scores": int(hl[1].item()),
              "preconditioners_checked_rank0": len(infos)}
    healthy = health["failed_choleskys"] == 0 and health["ranks_with_nonfinite_scores"] == 0

    if rank == 0:
        # Two kernel families, each with its own roofline object under a fixed key, compared family against family:
        #   roofline_hbm   the CG passes over the stored K_nM: knm_pass_kernel (one vector) and knm_pass2_kernel (two vectors
        #                  from one read; one launch per class), each also reported on its own
        #   roofline_mfma  the Gaussian contraction: K_nM build + fused scoring (one tile core, two kernels)
        # `roofline` is the FAMILY with more device time in THIS run's timed region (`dominant_kernel` names the single kernel
        # with the most); `roofline_step` is the whole step against SURVEY 8(d)'s roofline time.
        gauss_ms = ph["knm"].total_ms() + ph["mmv"].total_ms()
        gauss_launches = ph["knm"].count() + ph["mmv"].count()
        flops_per_launch = 2.0 * n_loc * M * D
        p1_ms, p1_n = ph["ktk"].total_ms(), ph["ktk"].count()
        p2_ms, p2_n = ph["ktk2"].total_ms(), ph["ktk2"].count()
        pass_ms, pass_launches = p1_ms + p2_ms, p1_n + p2_n
        bytes_per_pass = float(be.knm_bytes(n_loc, M))      # the stored shard, read exactly once per launch (n M s_K)
        # the pass kernels' names as rocprofv3 lists them: asked of the library (its own dispatch rule), not restated here
        from odx import hip as _hip
        kcode = {"f32": _hip.KNM_F32, "u24": _hip.KNM_U24, "bf16": _hip.KNM_BF16}[kfmt]
        pk = tuple((be.lib.odx_knm_pass_kernel_name(M, kcode, nv) or b"").decode() or "knm_pass(no configuration, nv=%d)" % nv for nv in (1, 2))
        gach = flops_per_launch * gauss_launches / max(gauss_ms * 1e-3, 1e-12) / 1e12

        def per_kernel(ms, cnt, unit_work, scale):
            return {"launches": cnt, "avg_launch_ms": round(ms / max(cnt, 1), 3),
                    "achieved": round(unit_work * cnt / max(ms * 1e-3, 1e-12) / scale, 2)} if cnt else None
        if be.gauss == "h2":
            # algorithmic flops (2 n M D) against the dense f16 MFMA peak; the two-term split issues 3 f16 MFMAs per
            # algorithmic product, so this formulation's own ceiling is peak / 3 (frac_of_split_ceiling)
            core = "h2w256" if be.lib.odx_gauss_h2_tile(n_loc, M) == 256 else "h2s16"
            gk = ("gauss_knm_%s_kernel" % core, "gauss_mmv_%s_kernel" % core)
            gpeak = F16_MFMA_PEAK_TFLOPS
        else:
            gk = ("gauss_knm_f32_kernel", "gauss_mmv_f32_kernel")
            gpeak = F32_MFMA_PEAK_TFLOPS
        roof_g = {"bound": "mfma", "kernel": "+".join(gk), "achieved": round(gach, 2), "peak": gpeak, "unit": "TFLOP/s",
                  "frac": round(gach / gpeak, 4), "traffic": None, "avg_launch_ms": round(gauss_ms / max(gauss_launches, 1), 3),
                  "family_ms_per_step": round(gauss_ms / args.steps, 2),
                  "per_kernel": {gk[0]: per_kernel(ph["knm"].total_ms(), ph["knm"].count(), flops_per_launch, 1e12),
                                 gk[1]: per_kernel(ph["mmv"].total_ms(), ph["mmv"].count(), flops_per_launch, 1e12)}}
        if be.gauss == "h2":
            roof_g["frac_of_split_ceiling"] = round(3 * gach / gpeak, 4)
        pach = bytes_per_pass * pass_launches / max(pass_ms * 1e-3, 1e-12) / 1e9
        roof_p = {"bound": "hbm", "kernel": pk[0] + ("+" + pk[1] if p2_n else ""), "achieved": round(pach, 1),
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(pach / HBM_PEAK_GBS, 4), "traffic": None,
                  "avg_launch_ms": round(pass_ms / max(pass_launches, 1), 3), "family_ms_per_step": round(pass_ms / args.steps, 2),
                  "bytes_per_launch": bytes_per_pass, "storage": kfmt,
                  "per_kernel": {pk[0]: per_kernel(p1_ms, p1_n, bytes_per_pass, 1e9),
                                 pk[1]: per_kernel(p2_ms, p2_n, bytes_per_pass, 1e9)}}
        if alone_gbps is not None:
            roof_p["achieved_alone"] = round(alone_gbps, 1)
            roof_p["frac_alone"] = round(alone_gbps / HBM_PEAK_GBS, 4)
            roof_p["note"] = ("achieved: K_nM bytes of all pass launches / their device time over the timed region (a two-vector "
                              "launch counts its ONE read of K_nM), where %d CUs are left to the preconditioner stream; "
                              "achieved_alone: the one-vector kernel, same buffer, idle GPU, before the timed region" % args.reserve_cus)
        if (n_loc, M, D) == (1_000_000, 10_000, 1024):
            # HBM traffic per launch, PER KERNEL (round-4 review: a median over both members of a family describes neither): each
            # kernel's own figure sits beside its launches; the family's `traffic` is that of its member with more device time
            for r in (roof_g, roof_p):
                best = None
                for kname, pkd in r["per_kernel"].items():
                    if pkd is None:
                        continue
                    pkd["traffic_GB_per_launch"], unit = profiled_traffic_gb(kname)
                    if pkd["traffic_GB_per_launch"] is not None and (best is None or pkd["launches"] * pkd["avg_launch_ms"] > best[0]):
                        best = (pkd["launches"] * pkd["avg_launch_ms"], kname, pkd["traffic_GB_per_launch"], unit)
                if best is not None:
                    r["traffic"], r["traffic_unit"] = best[2], "%s of %s" % (best[3], best[1])
                elif unit:
                    r["traffic_unit"] = unit
        kernel_ms = {pk[0]: p1_ms, pk[1]: p2_ms, gk[0]: ph["knm"].total_ms(), gk[1]: ph["mmv"].total_ms()}
        dominant = max(kernel_ms, key=kernel_ms.get)
        # `roofline` = the kernel FAMILY with more measured device time in this run's timed region (round-3 review: the single
        # kernel with the most time is the one-vector pass, but build + scoring together outweigh the passes — the family
        # furthest from its roofline must not hide behind the one closest to it); the single kernel with the most time is
        # still named, and both families are always there under their fixed keys
        roof = dict(roof_g if gauss_ms >= pass_ms else roof_p)
        roof["family_rule"] = "family with more device time in the timed region (gauss %.0f ms, passes %.0f ms per step)" % (
            gauss_ms / args.steps, pass_ms / args.steps)
        roof["dominant_kernel"] = dominant
        roof["dominant_kernel_ms_per_step"] = round(kernel_ms[dominant] / args.steps, 2)
        # The whole step against SURVEY 8(d)'s roofline time, per class: F_K = 2 n M D flop for the fit's K_nM and once more for
        # predict-all (dense peak of the contraction's dtype), B_CG = (t + 1) n M s_K bytes (s_K = the bytes an entry is stored
        # in; 8 TB/s), F_pc = 2 M^2 D + M^3 flop (the f64 MFMA peak: the factorisations are f64 here), over the GPUs sharing
        # the rows; achieved = roofline_time / measured_time.
        s_K = bytes_per_pass / max(float(n_loc) * M, 1.0)
        pk_f64 = 78.6
        F_K = 2.0 * (2.0 * N * M * D) * C
        B_CG = (args.maxiter + 1.0) * float(N) * M * s_K * C
        F_pc = (2.0 * M * M * D + float(M) ** 3) * C
        wshare = args.emulate_world if emulated else world      # GPUs the job's work is spread over
        t_K, t_CG, t_pc = F_K / (gpeak * 1e12) / wshare, B_CG / (HBM_PEAK_GBS * 1e9) / wshare, F_pc / (pk_f64 * 1e12) / wshare
        roof_step = {"F_K_flop": F_K, "B_CG_bytes": B_CG, "F_pc_flop": F_pc, "s_K_bytes_per_entry": round(s_K, 3),
                     "peaks": {"mfma_TFLOPs": gpeak, "hbm_GBps": HBM_PEAK_GBS, "mfma_f64_TFLOPs": pk_f64},
                     "t_K_s": round(t_K, 4), "t_CG_s": round(t_CG, 4), "t_pc_s": round(t_pc, 4),
                     "roofline_time_s": round(t_K + t_CG + t_pc, 4), "measured_time_s": round(ms_per_step / 1e3, 4),
                     "achieved": round((t_K + t_CG + t_pc) / (ms_per_step / 1e3), 4),
                     "formula": "SURVEY 8(d): F_K / P_mfma + B_CG / BW_hbm + F_pc / P_mfma(f64), per GPU share; F_K counts the "
                                "fit's build and predict-all"}
        phases = {k: round(v.total_ms() / args.steps, 2) for k, v in ph.items()}
        # the preconditioners run on side streams beside everything else: this is first-to-last-kernel time, not GPU time
        phases["precond_side_stream_span"] = phases.pop("precond")
        phases["pass_GBps"] = round(pach, 1)
        phases["gauss_TFLOPs"] = round(gach, 2)
        out = {
            "metric": "FALKON fit+infer samples/sec (N=1e6 D=1024 M=1e4)",
            "value": round(value, 1), "unit": "samples/s", "n_gpus": ranks_seen, "ranks": ranks_seen, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": (("f32-accurate K_nM (X Z' as a two-term f16 split on the f16 MFMA, f32 accumulate) stored as %s + f64 solver%s"
                                                 % ({"f32": "f32", "u24": "24-bit fixed point (step 2^-24)", "bf16": "bf16 (throughput only)"}[kfmt],
                                                    "" if odx.options.current().precond == "f64" or (M < 4096 and odx.options.current().precond != "split") else
                                                    " (preconditioner: K_MM, T and T^-1 in f64; the products of the A factor, which only preconditions, at f32 accuracy on the f16 MFMA)"))
                                                if be.gauss == "h2" else "f32 K_nM (f32-input MFMA) + f64 solver"),
            "data": "synthetic",
            "config": {"workload": "%d-class one-vs-rest FALKON fit + score-all, N=%d D=%d M=%d, %d CG iterations, "
                                   "rows sharded over %d GPU(s)" % (C, N, D, M, args.maxiter, args.emulate_world if emulated else world),
                       "N": N, "D": D, "M": M, "classes": C, "sigma": args.sigma, "lambda": args.lam,
                       "rows_per_gpu": n_loc, "preconditioners_per_batched_chain": G, "preconditioner_cus": args.precond_cus or "all",
                       "gaussians_on_the_complement": bool(args.gauss_on_complement and args.precond_cus),
                       "main_stream_high_priority": bool(args.main_priority), "preconditioner_lookahead_groups": args.precond_lookahead, "preconditioner_cus_full_groups_only": bool(args.precond_cus_full_only),
                       "lockstep_batch": job_b, "planned_GB_per_rank": plan_gb, "cg_exchange": args.cg_exchange,
                       # the one options table (odx/options.py): what this run's kernels and schedules were selected by
                       "options": odx.options.as_dict()},
            "roofline": roof,
            "roofline_hbm": roof_p,
            "roofline_mfma": roof_g,
            "roofline_step": roof_step,
            "phases_ms_per_step_rank0": phases,
            "health": health,
        }
        if emulated:
            out["metric"] += " — ONE RANK'S COMPUTE of a %d-rank job, emulated on one GPU (a prediction)" % args.emulate_world
            out["emulated"] = {"world": args.emulate_world, "rank": args.emulate_rank, "rows_of_this_rank": n_loc,
                               "collectives_replaced_by_local_copies_warmup_included": {k: {"calls": v[0], "bytes": v[1]} for k, v in shard.calls.items()},
                               "note": "value = N / this rank's compute-only step time: what the %d-rank job would reach if every rank took "
                                       "this long and the collectives were free; unmeasured on hardware" % args.emulate_world}
        if args.check:
            out["check"] = check_against_oracle(be, F, last, X, row_ids, args, C - 1)
        want_cpu = not args.no_cpu_baseline and world == 1 and not emulated      # reported at N = 1 only
        if want_cpu and (args.no_extras or world != 1 or emulated):
            out["cpu_baseline"] = cpu_baseline(args)
        if not args.no_extras and world == 1 and not emulated:
            # the other halves of BASELINE configs 2 and 3 (RLS regressors, feature forward) and the reference-regime
            # minibootstrap: measured after and outside the timed headline region, with its buffers released first
            job.release()
            last = F = X = scores = job = None
            be.release_workspaces()
            torch.cuda.empty_cache()
            # The extras run in a CHILD process (this one stays alive, idle, its buffers released): a fresh HIP runtime.  In this
            # process, behind the headline job, the latency-bound ones (one-image forwards, the harvest loop, the Minibootstrap)
            # read 10-60 % worse than the same calls in a fresh process, and by how much depends on what the job's streams left
            # behind — helper streams of the chains (found and released this round: DESIGN section 7), which hardware queues
            # later streams land on (the Minibootstrap: 0.46 or 0.51 s).  What is wanted here is what those paths cost, not what
            # the headline job's leftovers add to them; --extras inprocess keeps the old arrangement.
            # ... and, first, the latency-bound ones HERE, behind the job (round-5 review, item 6: both figures are reported — a
            # user's process will have run a fit before it runs a forward)
            try:
                from tools import bench_extras as _be_extras
                out["extras_in_headline_process"] = _be_extras.after_headline()
            except Exception as e:      # noqa: BLE001
                out["extras_in_headline_process"] = {"error": "%s: %s" % (type(e).__name__, e)}
            be.release_workspaces()
            torch.cuda.empty_cache()
            # (the CPU baseline BEHIND the in-process extras: its numpy run leaves a pool of host threads — one per core —
            # spinning for a while, and that, not anything the headline job left on the GPU, is what made the latency-bound
            # extras read 10-20 % slower in this process than in a fresh one: tools/after_headline_probe.py, round 6)
            if want_cpu:
                out["cpu_baseline"] = cpu_baseline(args)
            if args.extras == "inprocess":
                from tools import bench_extras
                out.update(bench_extras.collect(args))
            else:
                import subprocess
                cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_extras.py")] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT)
                    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
                    extras = json.loads(lines[-1]) if r.returncode == 0 and lines else {"extras_error": (r.stderr or r.stdout)[-400:]}
                except Exception as e:      # noqa: BLE001 — the extras must not take the headline line down with them
                    extras = {"extras_error": "%s: %s" % (type(e).__name__, e)}
                out.update(extras)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not healthy:
        print("bench.py: unhealthy run: %s" % json.dumps(health), file=sys.stderr)
        sys.exit(4)


def _same_kernel(name, profiled):
    """Whether the profiler's kernel name (namespace, `void`, spaces after commas, argument list) is the kernel `name`
    (`knm_passq_stag_kernel<10,2,1>`; a name without template arguments matches every instantiation)."""
    import re
    m = re.search(r"(?:\w+::)*(\w+)(<[^(]*>)?", profiled.replace("void ", "").strip())
    if not m:
        return False
    base, targs = m.group(1), (m.group(2) or "").replace(" ", "")
    want = name.replace(" ", "")
    return want == base + targs or ("<" not in want and want == base)


def csrc_sha16():
    """Fingerprint of the kernel sources this tree builds libodx.so from (sha256 over the sorted csrc/*.hip, *.h, *.cpp and
    the Makefile, first 16 hex digits).  tools/profile_round.sh stores it beside the PMC passes it takes
    (profiles/rNN_pmc_meta.json); `roofline.traffic` is only quoted from passes whose fingerprint is this tree's — a kernel
    changed after the last profile yields `null`, not stale bytes (round-5 review, weak 6).  Works without .git (GPU box)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "online-detection_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.cpp"))
                    + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def profiled_traffic_gb(kernel_names):
    """HBM bytes per launch of ONE kernel from the newest committed rocprofv3 PMC passes
    (profiles/rNN_pmc_*: the counters cannot be read from inside this process): FETCH_SIZE (KiB, doubled per the gfx950
    correction of the guide) + WRITE_SIZE (KiB), median over the profiled launches at this same shard shape.  The file
    stem the number came from is named in the unit string.  (None, reason) when there are no passes, or when the newest
    ones were taken on OTHER kernel sources than this tree's (profiles/rNN_pmc_meta.json: csrc_sha16)."""
    import csv
    import glob
    import re
    import statistics
    stems = sorted({re.match(r"(r\d+)_pmc_", os.path.basename(f)).group(1)
                    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*_counter_collection.csv"))
                    if re.match(r"r\d+_pmc_", os.path.basename(f))})
    for stem in reversed(stems):
        meta_path = os.path.join(ROOT, "profiles", "%s_pmc_meta.json" % stem)
        try:
            prof_sha = json.load(open(meta_path)).get("csrc_sha16")
        except Exception:
            prof_sha = None
        if prof_sha != csrc_sha16():
            return None, ("null: the newest PMC passes (profiles/%s_pmc_*) were taken on kernel sources %s, this tree's are %s "
                          "— re-run tools/profile_round.sh" % (stem, prof_sha or "without a recorded fingerprint", csrc_sha16()))
        tot, found = 0.0, 0
        for counter, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            path = os.path.join(ROOT, "profiles", "%s_pmc_%s_counter_collection.csv" % (stem, counter))
            if not os.path.exists(path):
                break
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                    if r["Counter_Name"] == counter and _same_kernel(kernel_names, r["Kernel_Name"])]
            if vals:
                found += 1
                tot += mult * 1024.0 * statistics.median(vals)
        if found == 2:
            return round(tot / 1e9, 2), ("GB per launch, rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, from the committed "
                                         "profiles/%s_pmc_* (not measured in this run; taken on this tree's kernel sources, "
                                         "csrc_sha16 %s)" % (stem, prof_sha))
    return None, None


def cpu_baseline(args):
    """The CPU path of the same algorithm (numpy f32 restatement of falkon's in-core fit with a
    stored K_nM, then predict) for ONE class on a bounded row sample, timed on this host."""
    from oracle import falkon_ref as fr
    try:
        from threadpoolctl import threadpool_info
        thr = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        thr = os.cpu_count() or 1
    big_host = thr >= 64
    ns = args.cpu_sample_rows if args.cpu_sample_rows > 0 else (100000 if big_host else 20000)
    D, M = args.D, (args.M if big_host else min(args.M, 4000))
    rng = np.random.default_rng(7)
    X = rng.standard_normal((ns, D)).astype(np.float32)
    X *= 20.0 / np.sqrt(D)
    y = np.where(np.arange(ns) % args.classes == 0, 1.0, -1.0).astype(np.float32)
    idx = fr.compute_indices_selection(y, M, lambda h, s: rng.integers(0, h, s))
    t0 = time.perf_counter()
    alpha, Z = fr.falkon_fit(X, y, idx, args.sigma, args.lam, maxiter=args.maxiter, dtype=np.float32)
    fr.falkon_predict(X, Z, alpha, args.sigma, np.float32)
    dt = time.perf_counter() - t0
    return {"value": round(ns / (dt * args.classes), 2), "unit": "samples/s", "cores": int(thr), "kind": "port",
            "sample": "oracle/falkon_ref.py (numpy f32, stored K_nM): 1 class fit+predict on %d rows, D=%d, M=%d "
                      "in %.1f s; value = rows / (seconds x %d classes)%s" % (
                          ns, D, M, dt, args.classes,
                          "" if M == args.M else "; M reduced from %d to bound the O(M^3) host Cholesky" % args.M)}


def check_against_oracle(be, F, last, X, row_ids, args, c):
    """Scores of the last fitted class on 2000 local rows vs the oracle's predict with the same alpha; and, when the job is
    small enough for the f64 oracle to fit it on the host (N x M <= 1e9), the class's alpha itself against the oracle's
    fit on ALL rows (regenerated here from the job's seeds) — under several ranks that is the sharded, lock-step fit
    against the single-process algorithm."""
    from oracle import falkon_ref as fr
    alpha, Zf = last
    rows = X[:2000].cpu().numpy().astype(np.float64)
    ref = fr.falkon_predict(rows, Zf.X.cpu().numpy().astype(np.float64), alpha.cpu().numpy()[:, None], args.sigma)
    Fs = be.features(X[:2000])
    got = be.mmv(Fs, Zf, args.sigma, alpha).cpu().numpy()
    out = {"max_abs_score_diff_vs_oracle_predict": float(np.abs(got - ref).max())}
    if args.n * args.M <= 1e9:                    # f64 K_nM on the host: 8 GB at N = 1e5, M = 1e4 (the headline's own width)
        Xall = synth_rows(0, args.n, args.D, args.classes, 1234 + 3, X.device).cpu().numpy().astype(np.float64)
        y = np.where(np.arange(args.n) % args.classes == c, 1.0, -1.0)
        idx = centre_indices(args.n, args.classes, args.M, 1234 + 3)[c]
        a_ref, _ = fr.falkon_fit(Xall, y, idx, args.sigma, args.lam, maxiter=args.maxiter, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-7)
        out["alpha_rel_err_vs_oracle_fit"] = float(np.linalg.norm(alpha.cpu().numpy() - a_ref[:, 0]) / np.linalg.norm(a_ref[:, 0]))
    return out


if __name__ == "__main__":
    main()
