#!/usr/bin/env python3
"""alpha of the headline's OWN kernels against the f64 oracle, at the headline's own width (M = 1e4) and up to its full height.

The north star's bar is "alphas within 1e-4 relative" (FALKONWrapper_with_centers_selection_incore.py:56-68 is the fit it
speaks of).  The parity tests assert it up to M = 2000; this tool runs ONE class of bench.py's synthetic job (same generator,
same centre rule, sigma = 15, lambda = 1e-5, 20 CG steps) through exactly what bench.py times —

    storage "auto" = 24-bit fixed point, gauss_knm_h2w256_kernel, knm_passq_stag_kernel (+ the folded two-vector pass),
    the M = 1e4 preconditioner chain, gauss_mmv_h2w256_kernel

— and through the same path on f32-stored K_nM (ODX_KNM=f32), and compares both with oracle/falkon_ref.falkon_fit in f64 on
the host (stored f64 K_nM when the host's memory holds it: 8 N M bytes; otherwise the row-blocked form that recomputes K per
product).  Prints one JSON line; `--out` also writes it to a file (the numbers quoted in DESIGN.md live in profiles/).

    python tools/alpha_at_scale.py --rows 1000000                 # one class at the full N = 1e6 (80 GB of host memory)
    python tools/alpha_at_scale.py --rows 100000 --classes-run 0 7
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def host_free_bytes():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def gpu_fit(be, F, y, Zf, sigma, lam, maxiter, storage):
    """One class through solver.falkon_fit_lockstep at world 1 (bench.py's path) on `storage`; returns alpha, scores, fmt."""
    import odx
    from odx.solver import SolverOptions, falkon_fit_lockstep
    prev = be.knm_storage
    be.knm_storage = storage
    try:
        fmt = be.knm_format(F.n, Zf.n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        P = be.precond_batched([Zf], sigma, lam, SolverOptions().pc_epsilon)[0]        # the class-batched chain bench.py runs
        alpha = falkon_fit_lockstep(be, F, [y], [Zf], sigma, lam, maxiter, SolverOptions(), precond=P)[0]
        del P
        scores = be.mmv(F, Zf, sigma, alpha)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        be.knm_storage = prev
    be.release_workspaces()
    torch.cuda.empty_cache()
    return alpha, scores, fmt, dt


def compare(N, D=1024, M=10_000, C=30, classes_run=(0,), storages=("auto", "f32"), sigma=15.0, lam=1e-5, maxiter=20, oracle="auto",
            log=None):
    """The comparison as a function (tests/test_gpu_configs.py runs it at the full N): returns the result dict."""
    import bench
    import odx
    from oracle import falkon_ref as fr
    be = odx.get_backend()
    dev = be.device
    seed = 1234 + 3
    X = bench.synth_rows(0, N, D, C, seed, dev)
    cidx = bench.centre_indices(N, C, M, seed)
    F = be.features(X)
    row_ids = torch.arange(N, device=dev)
    Xh = X.cpu().numpy().astype(np.float64)
    need = 8.0 * N * M + 2.0 * 8 * M * M * 3
    free = host_free_bytes()
    stored = oracle == "stored" or (oracle == "auto" and free > 1.15 * need + 8e9)
    res = {"workload": "one-vs-rest FALKON fit + score-all of bench.py's synthetic job, N=%d D=%d M=%d sigma=%g lambda=%g, %d CG steps"
                       % (N, D, M, sigma, lam, maxiter),
           "oracle": "oracle/falkon_ref.falkon_fit, f64, %s" % ("stored K_nM (%.0f GB)" % (8.0 * N * M / 1e9) if stored else "row-blocked (K recomputed per product)"),
           "host": {"cores": os.cpu_count(), "mem_available_GB": round(free / 1e9, 1)}, "classes": {}}
    srows = np.unique(np.concatenate([np.arange(0, N, max(1, N // 4000)), np.arange(0, min(N, 300)), np.arange(max(0, N - 300), N)]))
    for c in classes_run:
        y = torch.where((row_ids % C) == c, 1.0, -1.0).to(torch.float64)
        Zf = be.features(X.index_select(0, torch.from_numpy(cidx[c]).to(dev)))
        t0 = time.perf_counter()
        a_ref, Zh = fr.falkon_fit(Xh, y.cpu().numpy(), cidx[c], sigma, lam, maxiter=maxiter, dtype=np.float64,
                                  pc_eps=1e-5, cg_epsilon=1e-7, store_knm=stored, row_block=16384)
        t_or = time.perf_counter() - t0
        p_ref = fr.falkon_predict(Xh[srows], Zh, a_ref, sigma)[:, 0]
        entry = {"oracle_s": round(t_or, 1), "alpha_norm": float(np.linalg.norm(a_ref)), "score_scale": float(np.abs(p_ref).max())}
        got = {}
        for st in storages:
            alpha, scores, fmt, dt = gpu_fit(be, F, y, Zf, sigma, lam, maxiter, st)
            a = got[st] = alpha.cpu().numpy()
            entry[st] = {"stored_as": fmt, "gpu_s": round(dt, 3),
                         "alpha_rel_err": float(np.linalg.norm(a - a_ref[:, 0]) / np.linalg.norm(a_ref[:, 0])),
                         "alpha_max_abs_err_over_max_abs": float(np.abs(a - a_ref[:, 0]).max() / np.abs(a_ref[:, 0]).max()),
                         "score_max_abs_err_sampled": float(np.abs(scores[torch.from_numpy(srows).to(dev), 0].cpu().numpy() - p_ref).max()),
                         "sampled_rows": int(srows.size)}
            del alpha, scores
        if "auto" in got and "f32" in got:
            entry["alpha_rel_diff_auto_vs_f32_storage"] = float(np.linalg.norm(got["auto"] - got["f32"]) / np.linalg.norm(got["f32"]))
        res["classes"][str(c)] = entry
        if log is not None:
            print("class %d: %s" % (c, json.dumps(entry)), file=log, flush=True)
        del a_ref, Zh
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", dest="n", type=int, default=1_000_000)
    ap.add_argument("--dim", dest="D", type=int, default=1024)
    ap.add_argument("--centres", dest="M", type=int, default=10_000)
    ap.add_argument("--classes", type=int, default=30, help="classes of the synthetic job (positives of class c: rows i %% classes == c)")
    ap.add_argument("--classes-run", type=int, nargs="*", default=[0], help="which classes to fit and compare")
    ap.add_argument("--sigma", type=float, default=15.0)
    ap.add_argument("--lam", type=float, default=1e-5)
    ap.add_argument("--maxiter", type=int, default=20)
    ap.add_argument("--storages", nargs="*", default=["auto", "f32"])
    ap.add_argument("--oracle", choices=("auto", "stored", "blocked"), default="auto")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    res = compare(args.n, args.D, args.M, args.classes, args.classes_run, args.storages, args.sigma, args.lam, args.maxiter,
                  args.oracle, log=sys.stderr)
    line = json.dumps(res)
    print(line, flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
