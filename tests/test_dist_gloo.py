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
