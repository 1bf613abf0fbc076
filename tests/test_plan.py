"""The lock-step job's memory plan and class schedule (odx/plan.py; SURVEY 8e, first row): planned bytes per rank of every
BASELINE config at 1 / 2 / 4 / 8 ranks against 0.9 x 288 GB, and the properties of the rotating-owner schedule every rank
derives for itself."""
import pytest

from odx import plan

GB = 1e9
# BASELINE.json's configs as (N rows, D, M, classes); config 4's rows are per class (5e5 mask pixels each): the job sees
# one class's rows at a time, so N is that; config 1 is the CPU plumbing case
CONFIGS = {1: (5_000, 256, 500, 1), 2: (100_000, 1024, 2_000, 30), 3: (1_000_000, 1024, 10_000, 30),
           4: (500_000, 256, 2_000, 21), 5: (5_000_000, 1024, 20_000, 100)}


@pytest.mark.parametrize("cfg", sorted(CONFIGS))
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_planned_bytes_fit_the_gpu(cfg, world):
    N, D, M, C = CONFIGS[cfg]
    p = plan.plan_lockstep(N, D, M, C, world)
    if cfg == 5 and world == 1:
        # 5e6 x 2e4 entries are 300 GB in the 24-bit format: one GPU cannot hold one class's K_nM; the plan says so
        assert not p.feasible and p.parts["knm_shards"] > 288 * GB
        return
    assert p.feasible, p.summary()
    assert p.total_bytes <= 0.9 * 288 * GB and abs(p.total_bytes - sum(p.parts.values())) < 1
    assert world % p.b == 0 and 1 <= p.g <= 6
    assert p.parts["knm_shards"] == p.b * plan.knm_bytes_rule(p.n_loc, M, p.knm_format)
    assert p.parts["factors_two_groups"] == 2 * p.g * plan.factor_bytes(M)
    if cfg in (1, 2, 3, 4):
        assert p.b == world                              # every rank owns a class of every batch
    if cfg == 5:
        # round 3's fixed shape (world shards + 2 x 6 sets of factors per rank) at 8 ranks: 300 + 154 GB — what the plan avoids
        old = world * plan.knm_bytes_rule(p.n_loc, M, p.knm_format) + 2 * 6 * plan.factor_bytes(M)
        assert old > 288 * GB or world < 4
        assert {2: (1, 1), 4: (2, 2), 8: (4, 2)}[world] == (p.b, p.g)


def test_pinned_batch_and_chain_are_respected_or_refused():
    N, D, M, C = CONFIGS[5]
    p = plan.plan_lockstep(N, D, M, C, 8, batch=8, chain=6)
    assert not p.feasible and (p.b, p.g) == (8, 6)
    p = plan.plan_lockstep(N, D, M, C, 8, batch=2)
    assert p.feasible and p.b == 2 and p.g >= 2
    with pytest.raises(ValueError):
        plan.plan_lockstep(N, D, M, C, 8, batch=3)
    small = plan.plan_lockstep(*CONFIGS[3], 8, hbm_bytes=40 * GB)
    assert small.feasible and small.b < 8               # a smaller GPU: fewer shards in flight, not a failure


def test_chain_workspace_formula_is_the_one_of_the_library():
    """The planner's bytes for the preconditioner chain's scratch are what libodx asks for (host arithmetic on both sides:
    the library loads without a GPU)."""
    import odx
    lib = odx.hip.load()
    for M, D in ((10_000, 1024), (2_000, 2048), (20_000, 1024), (700, 36), (129, 7)):
        assert lib.odx_falkon_precond_workspace_bytes(M, D) == plan.chain_workspace_bytes(M, D)
        for B in (1, 6, 32):
            assert lib.odx_falkon_precond_batched_workspace_bytes(M, D, B) == B * plan.chain_workspace_bytes(M, D)


def test_storage_rule_matches_the_documented_thresholds():
    assert plan.knm_format_rule(1_000_000, 10_000) == "u24" and plan.knm_format_rule(125_000, 10_000) == "u24"
    assert plan.knm_format_rule(100_000, 2_000) == "u24" and plan.knm_format_rule(20_000, 10_000) == "u24"      # configs 2 / small shards
    assert plan.knm_format_rule(22_000, 2_000) == "f32" and plan.knm_format_rule(10_000_000, 1_000) == "f32"    # Minibootstrap fits; narrow blocks
    assert plan.knm_format_rule(1_000_000, 4096) == "u24" and plan.knm_format_rule(1_000_000, 10_000, "f32") == "f32"
    assert plan.knm_bytes_rule(1_000_000, 10_000, "u24") == 30_000_000_000 and plan.knm_bytes_rule(10, 10, "f32") == 10 * 12 * 4


@pytest.mark.parametrize("world,b,C", [(8, 8, 30), (8, 4, 30), (8, 2, 30), (8, 1, 11), (4, 2, 7), (1, 1, 3), (6, 3, 20)])
def test_owner_rotation(world, b, C):
    sched = plan.lockstep_batches(range(C), world, b)
    assert [c for batch, _ in sched for c in batch] == list(range(C))
    R = world // b
    for k, (batch, owners) in enumerate(sched):
        assert len(batch) == len(owners) <= b and len(set(owners)) == len(owners) and all(0 <= o < world for o in owners)
        assert owners == [(k % R) * b + j for j in range(len(batch))]
    # a full round gives every rank exactly one class
    for r0 in range(0, len(sched) - R + 1, R):
        owners = [o for _, os_ in sched[r0:r0 + R] for o in os_]
        if len(owners) == world:
            assert sorted(owners) == list(range(world))
    counts = [sum(o == r for _, os_ in sched for o in os_) for r in range(world)]
    assert max(counts) - min(counts) <= 1


def test_chain_groups_ramp_up():
    assert plan.precond_groups(30, 6) == [[0], [1, 2], [3, 4, 5]] + [list(range(s, min(s + 6, 30))) for s in range(6, 30, 6)]
    assert plan.precond_groups(4, 4) == [[0], [1, 2], [3]]
    assert plan.precond_groups(3, 3) == [[0], [1, 2]]
    assert plan.precond_groups(2, 1) == [[0], [1]]
