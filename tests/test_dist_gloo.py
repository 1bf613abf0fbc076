"""N > 1 on CPU: world_size-2 (and 3, ragged) gloo processes run the product's sharded FALKON
driver (odx.solver + odx.dist.RowShard) over contiguous row shards with the numpy-oracle backend,
and must reproduce the single-process fit; the sharded RLS Gram likewise."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import odx
        from odx.dist import RowShard
        from odx.rls import RegionRefinerTrainer
        from tests.oracle_backend import OracleBackend
        from tests.synth import blob_problem, centres
        odx.set_backend(OracleBackend(np.float64))
        be = odx.get_backend()
        X, y, rng = blob_problem(n, 32, seed=77)
        idx = centres(y, 96, rng)
        shard = RowShard()
        lo, hi = shard.bounds(n)
        assert shard.total(hi - lo) == n
        F = be.features(torch.from_numpy(X[lo:hi]))
        Zf = be.features(torch.from_numpy(X[idx]))          # centres replicated on every rank
        alpha = odx.falkon_fit(be, F, be.vec(y[lo:hi]), Zf, 6.0, 1e-4, 20, n_total=n, allreduce=shard.allreduce)
        # owner-computes mode: only the last rank ever builds / applies the preconditioner
        calls = []
        orig = be.precond
        be.precond = lambda *a, **k: (calls.append(rank), orig(*a, **k))[1]
        alpha_own = odx.falkon_fit(be, F, be.vec(y[lo:hi]), Zf, 6.0, 1e-4, 20, n_total=n, shard=shard, owner=world - 1)
        be.precond = orig
        assert calls == ([rank] if rank == world - 1 else []), calls
        assert torch.allclose(alpha_own, alpha, rtol=1e-9, atol=1e-12)
        # sharded RLS: every rank holds a slice of COXY
        g = torch.Generator().manual_seed(5)
        Xr = torch.randn(300, 12, generator=g)
        Cr = torch.randint(1, 3, (300, 1), generator=g).float()
        Yr = torch.randn(300, 4, generator=g) * 0.2
        l2, h2 = shard.bounds(300)
        cfg = {"CHOSEN_CLASSES": {0: "bg", 1: "a", 2: "b"}, "REGION_REFINER": {"opts": {"lambda": 5.0}}}
        import io
        from contextlib import redirect_stdout
        with redirect_stdout(io.StringIO()):
            models = RegionRefinerTrainer(cfg, 5.0, False, shard=shard)({"C": Cr[l2:h2], "O": None, "X": Xr[l2:h2], "Y": Yr[l2:h2]})
        W = np.stack([models[0]["Beta"][str(k)]["weights"].numpy() for k in range(4)])
        if rank == 0:
            ret["alpha"] = alpha.numpy()
            ret["W"] = W
            ret["T"] = models[0]["T"].numpy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1200), (3, 1001)])
def test_sharded_fit_equals_single_process(world, n):
    import odx
    from oracle import falkon_ref as fr
    from oracle import rls_ref
    from tests.synth import blob_problem, centres
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n, ret), nprocs=world, join=True)
    X, y, rng = blob_problem(n, 32, seed=77)
    idx = centres(y, 96, rng)
    ref, _ = fr.falkon_fit(X.astype(np.float64), y, idx, 6.0, 1e-4, maxiter=20, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-7)
    a = ret["alpha"]
    assert np.linalg.norm(a - ref[:, 0]) / np.linalg.norm(ref[:, 0]) < 1e-6
    g = torch.Generator().manual_seed(5)
    Xr = torch.randn(300, 12, generator=g)
    Cr = torch.randint(1, 3, (300, 1), generator=g).float()
    Yr = torch.randn(300, 4, generator=g) * 0.2
    m = rls_ref.train(Cr.numpy(), Xr.numpy(), Yr.numpy(), 3, 5.0)[0]
    assert np.abs(ret["W"] - m["W"]).max() < 2e-6 and np.abs(ret["T"] - m["T"]).max() < 2e-6


def test_shard_bounds_cover_rows_exactly():
    from odx.dist import shard_bounds
    for n in (0, 1, 7, 1000, 1000003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [h - l for l, h in b]
            assert max(sizes) - min(sizes) <= 1


def _lockstep_worker(rank, world, port, n, B, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import odx
        from odx.dist import RowShard
        from tests.oracle_backend import OracleBackend
        from tests.synth import blob_problem, centres
        odx.set_backend(OracleBackend(np.float64))
        be = odx.get_backend()
        X, y, rng = blob_problem(n, 24, seed=13)
        shard = RowShard()
        lo, hi = shard.bounds(n)
        F = be.features(torch.from_numpy(X[lo:hi]))
        M = 61                                             # odd: exercises the padded exchange rows
        ys, Zfs = [], []
        for b in range(B):
            yb = np.where(np.arange(n) % 5 == b, 1.0, -1.0)
            ys.append(be.vec(yb[lo:hi]))
            Zfs.append(be.features(torch.from_numpy(X[centres(yb.astype(np.float32), M, np.random.default_rng(b))])))
        built = []
        orig = be.precond
        be.precond = lambda *a, **k: (built.append(rank), orig(*a, **k))[1]
        alphas = odx.falkon_fit_lockstep(be, F, ys, Zfs, 5.0, 1e-4, 20, n_total=n, shard=shard)
        be.precond = orig
        assert built == ([rank] if rank < B else []), built            # one preconditioner per owner, none elsewhere
        # the same problems one at a time in owner mode
        for b in range(B):
            ref = odx.falkon_fit(be, F, ys[b], Zfs[b], 5.0, 1e-4, 20, n_total=n, shard=shard, owner=b)
            assert torch.allclose(alphas[b], ref, rtol=1e-9, atol=1e-12), b
        if rank == world - 1:
            ret["alphas"] = [a.numpy() for a in alphas]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 2), (3, 2)])
def test_lockstep_fit_equals_one_at_a_time(world, B):
    """falkon_fit_lockstep (one all-gather + one reduce-scatter per CG step for B problems) gives every problem the
    alpha of its own owner-mode fit, and of the single-process oracle."""
    from oracle import falkon_ref as fr
    from tests.synth import blob_problem, centres
    n = 1200      # 600 rows per rank at world 2: the folded full residual (two vectors per problem in the exchange); 400 at world 3: the plain sequence
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_lockstep_worker, args=(world, port, n, B, ret), nprocs=world, join=True)
    X, y, rng = blob_problem(n, 24, seed=13)
    for b in range(B):
        yb = np.where(np.arange(n) % 5 == b, 1.0, -1.0)
        idx = centres(yb.astype(np.float32), 61, np.random.default_rng(b))
        ref, _ = fr.falkon_fit(X.astype(np.float64), yb, idx, 5.0, 1e-4, maxiter=20, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-7)
        a = ret["alphas"][b]
        assert np.linalg.norm(a - ref[:, 0]) / np.linalg.norm(ref[:, 0]) < 1e-6


# ------------------------------------------------------------------ classes over ranks (Minibootstrap regime)
def _minibootstrap_problem(path):
    """4 classes (one without positives), 3 negative batches each, through the drop-in classes and the odx FALKONWrapper."""
    import yaml
    C, D = 4, 16
    cfg = {"NUM_CLASSES": C + 1, "CHOSEN_CLASSES": {i: ("bg" if i == 0 else "obj%d" % i) for i in range(C + 1)},
           "ONLINE_REGION_CLASSIFIER": {"MINIBOOTSTRAP": {"EASY_THRESH": -0.9, "HARD_THRESH": -0.7},
                                        "CLASSIFIER": {"lambda": 0.001, "sigma": 6.0, "M": 24, "kernel_type": "gauss"}}}
    with open(path, "w") as fid:
        yaml.safe_dump(cfg, fid)
    g = torch.Generator().manual_seed(11)
    mu = torch.randn((C, D), generator=g) * 2
    pos = [mu[c] + 0.6 * torch.randn((40, D), generator=g) if c != 2 else torch.empty((0, D)) for c in range(C)]
    neg = [[mu[(c + 1 + j) % C] * 0.6 + 0.9 * torch.randn((60, D), generator=g) for j in range(3)] for c in range(C)]
    return C, pos, neg


def _train_minibootstrap(path, pos, neg, opts):
    import io
    from contextlib import redirect_stdout
    from tests import dropin
    W, ORC = dropin.load("FALKONWrapper_with_centers_selection_incore"), dropin.load("OnlineRegionClassifier_incore")
    torch.manual_seed(3)
    with redirect_stdout(io.StringIO()):
        orc = ORC.OnlineRegionClassifier(W.FALKONWrapper(cfg_path=path), [p.clone() for p in pos], [[b.clone() for b in bs] for bs in neg],
                                         None, cfg_path=path)
        return orc.trainRegionClassifier(opts=dict(opts, normalized=True))


def _class_shard_worker(rank, world, port, path, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import odx
        from tests.oracle_backend import OracleBackend
        odx.set_backend(OracleBackend(np.float64))
        C, pos, neg = _minibootstrap_problem(path + ".%d" % rank)
        fits = []
        be = odx.get_backend()
        orig = be.precond
        be.precond = lambda *a, **k: (fits.append(1), orig(*a, **k))[1]
        models = _train_minibootstrap(path + ".%d" % rank, pos, neg, {"class_shard": True})
        be.precond = orig
        ret[rank] = ([None if m is None else (m.ny_points_.numpy(), m.alpha_.numpy(), m.M, m.kernel.sigma) for m in models], len(fits))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_class_sharded_minibootstrap_equals_single_process(world, tmp_path):
    """opts['class_shard']: every rank trains the classes i % world == rank and ends with ALL models, bit for bit those of
    one process that gives each class its own RNG stream (opts['class_rng']); no rank trains another rank's classes."""
    import odx
    from tests.oracle_backend import OracleBackend
    path = str(tmp_path / "cfg.yaml")
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_class_shard_worker, args=(world, port, path, ret), nprocs=world, join=True)
    odx.set_backend(OracleBackend(np.float64))
    try:
        C, pos, neg = _minibootstrap_problem(path)
        ref = _train_minibootstrap(path, pos, neg, {"class_rng": True})
        plain = _train_minibootstrap(path, pos, neg, {})
    finally:
        odx.set_backend(None)
    assert ref[2] is None and all(ref[c] is not None for c in (0, 1, 3))
    assert not all(torch.equal(ref[c].alpha_, plain[c].alpha_) for c in (0, 1, 3))      # the per-class streams are a different draw order
    for rank in range(world):
        models, nfits = ret[rank]
        assert nfits == 3 * sum(1 for c in (0, 1, 3) if c % world == rank)              # 3 batches per owned class, nothing else
        for c in range(C):
            if ref[c] is None:
                assert models[c] is None
                continue
            ny, alpha, M, sigma = models[c]
            assert np.array_equal(ny, ref[c].ny_points_.numpy()) and np.array_equal(alpha, ref[c].alpha_.numpy())
            assert M == ref[c].M and sigma == 6.0


def _job_worker(rank, world, port, N, D, M, C, ret, batch=0, chain=0, exchange="lockstep", lookahead=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import odx
        from odx.dist import RowShard
        from odx.job import LockstepClassJob
        from tests.oracle_backend import OracleBackend
        odx.set_backend(OracleBackend(np.float64))
        be = odx.get_backend()
        X, cidx = _job_problem(N, D, M, C)
        shard = RowShard()
        lo, hi = shard.bounds(N)
        row_ids = torch.arange(lo, hi)
        job = LockstepClassJob(be, torch.from_numpy(X[lo:hi]), N, M, lambda c: torch.where((row_ids % C) == c, 1.0, -1.0).double(),
                               [torch.from_numpy(i) for i in cidx], 6.0, 1e-4, 20, shard=shard, batch=batch, precond_batch=chain,
                               exchange=exchange, precond_lookahead=lookahead)
        F = be.features(job.X)
        alpha, _ = job.run(F)
        ret[rank] = {"scores": job.scores.numpy().copy(), "trace": list(job.trace), "G": job.G, "alpha_last": alpha.numpy().copy(),
                     "b": job.b, "kbufs": len(job.kbufs)}
    finally:
        dist.destroy_process_group()


def _job_problem(N, D, M, C):
    rng = np.random.default_rng(99)
    mu = rng.standard_normal((C, D))
    X = mu[np.arange(N) % C] + 0.7 * rng.standard_normal((N, D))
    X -= X.mean(0)
    X *= 6.0 / np.linalg.norm(X, axis=1).mean()
    X = np.ascontiguousarray(X.astype(np.float32))
    cidx = []
    for c in range(C):
        pos = np.flatnonzero(np.arange(N) % C == c)
        neg = np.flatnonzero(np.arange(N) % C != c)
        cidx.append(np.concatenate([pos[: M // 2], neg[rng.integers(0, len(neg), M - min(M // 2, len(pos)))]]).astype(np.int64))
    return X, cidx


def test_headline_control_flow_on_eight_ranks():
    """The headline job's own control flow (odx/job.py, what bench.py runs) as 8 gloo ranks on the CPU with the oracle
    backend: 30 classes on 8 ranks -> lock-step batches of 8, 8, 8, 6 — in the last one two ranks own nothing —, the
    owned classes' preconditioners in class-batched groups of 1 + 2 + 1 batches (n_batches = 4 -> G = 4), one all-gather + one
    reduce-scatter per CG iteration for a whole batch.  Every rank must run exactly that schedule, and the scores of all 30
    classes must equal a single process fitting the classes one after the other."""
    import odx
    from oracle import falkon_ref as fr
    N, D, M, C, world = 2400, 16, 40, 30, 8
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_job_worker, args=(world, port, N, D, M, C, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == list(range(world))
    want_batches = [tuple(range(0, 8)), tuple(range(8, 16)), tuple(range(16, 24)), tuple(range(24, 30))]
    for r in range(world):
        t = ret[r]["trace"]
        assert ret[r]["G"] == 4
        assert [p for k, p in t if k == "fit"] == want_batches
        owned = [p for k, p in t if k == "precond"]
        # groups of batches [0], [1, 2], [3] (sizes 1, 2, 3 while smaller than G = 4, cut by the 4 batches there are): rank r
        # owns class r of batch 0, then classes 8 + r and 16 + r, then — ranks 0..5 only — class 24 + r
        assert owned == [(r,), (8 + r, 16 + r)] + ([(24 + r,)] if r < 6 else []), (r, owned)
        # every group is issued one group ahead: the first two before the first fit, the third before the second fit
        kinds = [k for k, _ in t]
        assert kinds == (["precond", "precond", "fit", "precond", "fit", "fit", "fit"] if r < 6 else
                         ["precond", "precond", "fit", "fit", "fit", "fit"]), (r, kinds)
    scores = np.concatenate([ret[r]["scores"] for r in range(world)], axis=0)
    X, cidx = _job_problem(N, D, M, C)
    Xd = X.astype(np.float64)
    for c in (0, 7, 8, 23, 24, 29):
        y = np.where(np.arange(N) % C == c, 1.0, -1.0)
        a, Z = fr.falkon_fit(Xd, y, cidx[c], 6.0, 1e-4, maxiter=20, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-7)
        want = fr.falkon_predict(Xd, Z, a, 6.0)[:, 0]
        assert np.abs(scores[:, c] - want).max() < 1e-6 * max(1.0, np.abs(want).max()), c
    assert (scores.argmax(1) == np.arange(N) % C).mean() > 0.9


def test_chain_groups_several_ahead_of_the_fits():
    """precond_lookahead = 3: the class-batched preconditioner chains of three groups are issued before the first fit and every
    later group three groups ahead (four factor blocks in rotation) instead of one — what lets chains confined to a few compute
    units keep pace with the fits (bench.py --precond-lookahead / --precond-cus-full-only).  Same fits: scores equal the
    one-group-ahead schedule's bit for bit, and the recorded schedule shows the issue order."""
    N, D, M, C, world = 1300, 16, 40, 26, 2
    ret, ret1 = mp.Manager().dict(), mp.Manager().dict()
    mp.spawn(_job_worker, args=(world, _free_port(), N, D, M, C, ret, 0, 0, "lockstep", 3), nprocs=world, join=True)
    mp.spawn(_job_worker, args=(world, _free_port(), N, D, M, C, ret1), nprocs=world, join=True)
    for r in range(world):
        assert np.array_equal(ret[r]["scores"], ret1[r]["scores"])
        kinds = [k for k, _ in ret[r]["trace"]]
        # three groups before the loop and the fourth as the first group's fit is reached (one-group-ahead: precond, precond, fit)
        assert kinds[:5] == ["precond", "precond", "precond", "precond", "fit"], kinds[:6]
        assert [p for k, p in ret[r]["trace"] if k == "precond"] == [p for k, p in ret1[r]["trace"] if k == "precond"]   # same groups, same owners
        assert [p for k, p in ret[r]["trace"] if k == "fit"] == [p for k, p in ret1[r]["trace"] if k == "fit"]
        # every group is issued before the first fit of the group three behind it
        t = ret[r]["trace"]
        pos_pre = [i for i, (k, _) in enumerate(t) if k == "precond"]
        first_fit_of_class = {c: i for i, (k, p) in reversed(list(enumerate(t))) if k == "fit" for c in p}
        for gi, i in enumerate(pos_pre):
            for c in t[i][1]:
                assert i < first_fit_of_class[c]


@pytest.mark.parametrize("world", [2, 3])
def test_replicated_one_allreduce_form_of_the_job(world):
    """exchange = "allreduce" (the north star's literal form; bench.py --cg-exchange allreduce): classes one at a time, EVERY
    rank builds every class's preconditioner — chains over groups of 1, 2, 3 classes, one group ahead — and runs the M-sized
    algebra of every class, one all-reduce of the K_nM pass's partial per CG iteration; one K_nM shard buffer per rank.
    Scores of all classes equal the single-process oracle's and the lock-step job's (same arithmetic per class)."""
    from oracle import falkon_ref as fr
    N, D, M, C = 900, 16, 40, 7
    ret, ret2 = mp.Manager().dict(), mp.Manager().dict()
    mp.spawn(_job_worker, args=(world, _free_port(), N, D, M, C, ret, 0, 0, "allreduce"), nprocs=world, join=True)
    mp.spawn(_job_worker, args=(world, _free_port(), N, D, M, C, ret2), nprocs=world, join=True)
    for r in range(world):
        t = ret[r]["trace"]
        assert ret[r]["b"] == 1 and ret[r]["kbufs"] == 1
        assert [p for k, p in t if k == "fit"] == [(c,) for c in range(C)]
        assert [c for k, p in t if k == "precond" for c in p] == list(range(C))         # every rank builds every preconditioner
        g = ret[r]["G"]
        from odx import plan
        assert [len(p) for k, p in t if k == "precond"] == [len(grp) for grp in plan.precond_groups(C, max(g, 1))]
        kinds = [k for k, _ in t]
        assert kinds[:3] == ["precond", "precond", "fit"]                               # one group ahead of the fits
    scores = np.concatenate([ret[r]["scores"] for r in range(world)], axis=0)
    scores2 = np.concatenate([ret2[r]["scores"] for r in range(world)], axis=0)
    assert np.abs(scores - scores2).max() < 1e-9 * max(1.0, np.abs(scores2).max())     # the two exchanges: the same fits
    X, cidx = _job_problem(N, D, M, C)
    Xd = X.astype(np.float64)
    for c in range(C):
        y = np.where(np.arange(N) % C == c, 1.0, -1.0)
        a, Z = fr.falkon_fit(Xd, y, cidx[c], 6.0, 1e-4, maxiter=20, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-7)
        want = fr.falkon_predict(Xd, Z, a, 6.0)[:, 0]
        assert np.abs(scores[:, c] - want).max() < 1e-6 * max(1.0, np.abs(want).max()), c


def test_centres_are_assembled_from_the_rows_each_rank_owns():
    """LockstepClassJob.gather_centres under 3 gloo ranks: every rank contributes only the centre rows it owns (one
    all-gather of equal-sized blocks) and ends with X_global[idx] in idx order, duplicates and unbalanced ownership included."""
    N, D, world = 101, 8, 3
    ret = mp.Manager().dict()
    mp.spawn(_centre_worker, args=(world, _free_port(), N, D, ret), nprocs=world, join=True)
    rng = np.random.default_rng(5)
    X = rng.standard_normal((N, D)).astype(np.float32)
    for r in range(world):
        for idx, Z in ret[r]:
            assert np.array_equal(Z, X[idx]), r


def _centre_worker(rank, world, port, N, D, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import odx
        from odx.dist import RowShard
        from odx.job import LockstepClassJob
        from tests.oracle_backend import OracleBackend
        odx.set_backend(OracleBackend(np.float64))
        be = odx.get_backend()
        rng = np.random.default_rng(5)
        X = rng.standard_normal((N, D)).astype(np.float32)
        shard = RowShard()
        lo, hi = shard.bounds(N)
        idxs = [np.array([100, 0, 0, 50, 33, 34, 99, 1], dtype=np.int64),          # duplicates, every rank owns some
                np.arange(0, 30, dtype=np.int64),                                   # all rows on rank 0
                np.array([100], dtype=np.int64)]                                    # one row, last rank
        job = LockstepClassJob(be, torch.from_numpy(X[lo:hi]), N, 8, lambda c: None, [torch.from_numpy(i) for i in idxs], 6.0, 1e-4,
                               20, shard=shard)
        out = []
        for i in idxs:
            Zf = job.gather_centres(torch.from_numpy(i))
            out.append((i, Zf.X.numpy().copy()))
            Zf2 = job.gather_centres(job.cidx[0]) if i is idxs[0] else None       # the cached plan gives the same rows
            if Zf2 is not None:
                assert np.array_equal(Zf2.X.numpy(), X[idxs[0]])
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("b,chain", [(2, 2), (4, 0)])
def test_lockstep_batches_smaller_than_the_world(b, chain):
    """The memory plan's other shape (odx/plan.py): lock-step batches of b < world classes — b K_nM shards per rank instead
    of `world` — whose owners rotate through the ranks, so that every ROUND of world / b batches gives each rank one class;
    the preconditioner chains are grouped in rounds.  8 gloo ranks, 30 classes, b = 2 (15 batches, rounds of 4) and b = 4 (8
    batches, rounds of 2): every rank runs exactly that schedule, holds b shard buffers, builds only the preconditioners of
    the classes it owns, and the scores of all classes equal a single process fitting them one after the other."""
    from odx import plan
    from oracle import falkon_ref as fr
    N, D, M, C, world = 1600, 16, 40, 30, 8
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_job_worker, args=(world, port, N, D, M, C, ret, b, chain), nprocs=world, join=True)
    sched = plan.lockstep_batches(range(C), world, b)
    R = world // b
    n_rounds = (len(sched) + R - 1) // R
    owned_by = {r: [c for batch, owners in sched for c, o in zip(batch, owners) if o == r] for r in range(world)}
    assert sorted(c for cs in owned_by.values() for c in cs) == list(range(C))          # every class has exactly one owner
    assert max(len(v) for v in owned_by.values()) - min(len(v) for v in owned_by.values()) <= 1       # balanced
    for r in range(world):
        t = ret[r]["trace"]
        assert ret[r]["b"] == b and ret[r]["kbufs"] == b
        g = ret[r]["G"]
        assert g == (chain or min(6, n_rounds))
        assert [p for k, p in t if k == "fit"] == [tuple(batch) for batch, _ in sched]
        built = [c for k, p in t if k == "precond" for c in p]
        assert built == owned_by[r], (r, built, owned_by[r])
        assert all(len(p) <= g for k, p in t if k == "precond")                         # at most g classes per chain
    scores = np.concatenate([ret[r]["scores"] for r in range(world)], axis=0)
    X, cidx = _job_problem(N, D, M, C)
    Xd = X.astype(np.float64)
    for c in (0, 1, 2, 7, 8, 15, 28, 29):
        y = np.where(np.arange(N) % C == c, 1.0, -1.0)
        a, Z = fr.falkon_fit(Xd, y, cidx[c], 6.0, 1e-4, maxiter=20, dtype=np.float64, pc_eps=1e-5, cg_epsilon=1e-7)
        want = fr.falkon_predict(Xd, Z, a, 6.0)[:, 0]
        assert np.abs(scores[:, c] - want).max() < 1e-6 * max(1.0, np.abs(want).max()), c


@pytest.mark.parametrize("world,rank", [(8, 0), (8, 7), (4, 2)])
def test_emulated_rank_runs_the_schedule_of_the_real_rank(world, rank):
    """odx.dist.EmulatedShard (bench.py --emulate-world): ONE rank's share of a `world`-rank job in a single process — the
    schedule it executes (every lock-step batch fitted, the preconditioners of exactly the classes that rank owns, chain by
    chain), the rows it holds and the collectives it would have issued (one all-gather + one reduce-scatter of a (world, M)
    f64 matrix per exchange, one all-gather of the centres' rows per class) are those of the real rank of the gloo runs above; its
    numbers are finite (they are a rank's partial sums, not the job's)."""
    import odx
    from odx import plan
    from odx.dist import EmulatedShard, shard_bounds
    from odx.job import LockstepClassJob
    from tests.oracle_backend import OracleBackend
    N, D, M, C = 2400, 16, 40, 30
    odx.set_backend(OracleBackend(np.float64))
    try:
        be = odx.get_backend()
        X, cidx = _job_problem(N, D, M, C)
        shard = EmulatedShard(world, rank)
        lo, hi = shard.bounds(N)
        assert (lo, hi) == shard_bounds(N, world, rank)
        # centre indices folded onto this rank's rows, class ids kept (what bench.py does under --emulate-world)
        folded = []
        for i in cidx:
            cls = i % C
            first = lo + ((cls - lo) % C)
            cnt = np.maximum((hi - first + C - 1) // C, 1)
            folded.append(first + C * ((i // C) % cnt))
            assert folded[-1].min() >= lo and folded[-1].max() < hi and ((folded[-1] % C) == cls).all()
        row_ids = torch.arange(lo, hi)
        job = LockstepClassJob(be, torch.from_numpy(X[lo:hi]), N, M, lambda c: torch.where((row_ids % C) == c, 1.0, -1.0).double(),
                               [torch.from_numpy(i) for i in folded], 6.0, 1e-4, 20, shard=shard)
        job.run(be.features(job.X))
        sched = plan.lockstep_batches(range(C), world, job.b)
        assert [p for k, p in job.trace if k == "fit"] == [tuple(b) for b, _ in sched]
        owned = [c for b, owners in sched for c, o in zip(b, owners) if o == rank]
        assert [c for k, p in job.trace if k == "precond" for c in p] == owned
        assert torch.isfinite(job.scores).all() and tuple(job.scores.shape) == (hi - lo, C)
        calls = shard.calls
        assert calls["centre_gather"] == [C, C * M * D * 4] and calls["all_reduce"][0] == 0     # the centres of every class: one all-gather of f32 rows each
        assert calls["all_gather"][0] == calls["reduce_scatter"][0]     # paired: per exchange, + the right-hand side's scatter / alpha's gather ...
        Mp = (M + 1) // 2 * 2
        per = world * Mp * 8
        assert calls["all_gather"][1] % per == 0 and calls["reduce_scatter"][1] % per == 0      # ... of (world, M) f64 matrices (or two vectors wide)
        assert calls["all_gather"][0] >= len(sched) * 21 and calls["reduce_scatter"][0] >= len(sched) * 21
    finally:
        odx.set_backend(None)
