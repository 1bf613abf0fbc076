"""Memory plan and class schedule of the lock-step job (odx/job.py) — SURVEY §8e, first row.

The job's HBM-sized state per rank is (a) one stored K_nM shard per class of the lock-step batch in flight and (b) the
f64 preconditioner factors of the classes whose chains are in flight (two groups: the one being fitted, the one being
built) plus the chain's workspace.  Round 3 sized (a) as `world` shards and (b) as 2 x 6 classes whatever the problem:
N M s_K bytes of K_nM per rank at EVERY world size, which BASELINE config 5 (N = 5e6, M = 2e4, 100 classes on 8 GPUs:
300 GB of K_nM + 154 GB of factors per rank against 288 GB of HBM) cannot hold.  Here both are chosen from a budget:

    b   classes per lock-step batch, a divisor of the world size (b = world when it fits).  With b < world the owners
        of a batch are b of the ranks; consecutive batches rotate through the ranks, so that over a ROUND of world / b
        batches every rank owns exactly one class (the preconditioner work stays balanced);
    g   classes per rank and preconditioner chain = rounds per chain group (<= 6: past that the batched chain gains
        nothing, docs/HISTORY.md 7).

Everything here is host arithmetic on quantities every rank agrees on (N, D, M, C, world, the budget), so all ranks derive
the same schedule and issue the same collectives.  The reference has no counterpart (single process, one class at a time:
OnlineRegionClassifier_incore.py:96-155).
"""
from dataclasses import dataclass, field

HBM_BYTES_MI355X = 288e9          # /opt/skills/guides/MI355X_MICROARCH.md: 288 GB HBM3E per GPU
BUDGET_FRACTION = 0.9
MAX_CHAIN_CLASSES = 6
RESERVE_BYTES = 3e9               # allocator slack, RCCL buffers, code objects, small vectors


def _round_up(x, m):
    return (int(x) + m - 1) // m * m


def knm_format_rule(n, M, storage="auto", wide_tile=True):
    """The storage format HipBackend.knm_format gives an (n, M) block (tests/test_gpu_modules.py compares the two): 24-bit
    fixed point where the passes are HBM-bound (>= 2^27 entries, at least 1024 centres, the wide tile core), f32 below."""
    if storage in ("f32", "u24", "bf16"):
        return storage
    if n * M >= (1 << 27) and M >= 1024 and wide_tile:
        return "u24"
    return "f32"


def knm_bytes_rule(n, M, fmt):
    """Bytes of a stored (n, M) block: f32 rows of roundup(M, 4) floats; the compact formats planes of roundup(M, 8)
    entries, 3 (u16 + u8 planes) or 2 (bf16) bytes each (include/odx.h, odx_knm_bytes)."""
    if fmt == "f32":
        return int(n) * _round_up(M, 4) * 4
    return int(n) * _round_up(M, 8) * (3 if fmt == "u24" else 2)


def chain_workspace_bytes(M, D):
    """odx_falkon_precond_batched_workspace_bytes per class: the f64 centres, four M x ld work matrices, the diagonal
    blocks' inverses, and the packed f16 splits the A factor's products run on from 4096 centres on (2 M roundup(M, 64) units for
    T and, later, a merge level's four operands; two 512-column panels) — the scratch is sized for every M."""
    ld = _round_up(M, 2)
    split = (2 * M * _round_up(M, 64) + 2 * M * 512) // 2 + 2
    ldz = _round_up(D, 2)
    ldz += 64 if ldz % 128 == 0 else 0           # (rows of the f64 centres are kept off a power-of-two stride)
    return (M * ldz + ld + 4 * M * ld + 2 * ((M + 127) // 128) * 128 * 128 + split) * 8


def factor_bytes(M):
    """The four M x ld f64 inverse factors of one class."""
    return 4 * M * _round_up(M, 2) * 8


def divisors_desc(world):
    return [b for b in range(world, 0, -1) if world % b == 0]


def owners_of_batch(k, size, world, b):
    """Ranks owning positions 0 .. size - 1 of lock-step batch k: the round's k-th block of b ranks."""
    R = world // b
    base = (k % R) * b
    return [base + j for j in range(size)]


def lockstep_batches(classes, world, b):
    """[(classes of the batch, their owner ranks)] for `classes` in order, batches of b."""
    classes = list(classes)
    return [(classes[k0:k0 + b], owners_of_batch(k0 // b, len(classes[k0:k0 + b]), world, b)) for k0 in range(0, len(classes), b)]


def precond_groups(n_units, G):
    """Units (rounds of batches) grouped for the class-batched preconditioner chains: sizes 1, 2, 3 (each only while smaller
    than G), then G — nothing but the first chain is waited for at the start of a job, every later group is built while the
    group before it is being fitted.  Returns a list of lists of unit indices."""
    groups, g0 = [], 0
    for size in (1, 2, 3):
        if g0 < n_units and size < G:
            groups.append(list(range(g0, min(g0 + size, n_units))))
            g0 += size
    while g0 < n_units:
        groups.append(list(range(g0, min(g0 + G, n_units))))
        g0 += G
    return groups


@dataclass
class JobPlan:
    world: int
    b: int                      # classes per lock-step batch
    g: int                      # classes per rank and preconditioner chain (rounds per chain group)
    n_loc: int
    knm_format: str
    feasible: bool
    budget_bytes: float
    total_bytes: float
    parts: dict = field(default_factory=dict)

    @property
    def rounds_per_batch_cycle(self):
        return self.world // self.b

    def summary(self):
        gb = {k: round(v / 1e9, 2) for k, v in self.parts.items()}
        return "world %d: b = %d, g = %d, %s K_nM, %.1f of %.1f GB per rank %s" % (
            self.world, self.b, self.g, self.knm_format, self.total_bytes / 1e9, self.budget_bytes / 1e9, gb)


def _parts(N, D, M, C, world, b, g, fmt, gauss="h2", knm_bytes=None, rows_resident=True, lookahead=1, chain_sides=1):
    n_loc = (int(N) + world - 1) // world
    kb = knm_bytes(n_loc, M) if knm_bytes is not None else knm_bytes_rule(n_loc, M, fmt)
    ldx = _round_up(D, 4)
    parts = {}
    if rows_resident:
        parts["rows_f32"] = n_loc * ldx * 4 + n_loc * 4
        parts["rows_packed"] = n_loc * _round_up(D, 64) * 4 + (n_loc * (_round_up(D, 128) + 4) if gauss == "f8" else 0)
    parts["labels"] = 2 * b * n_loc * 8
    parts["scores"] = n_loc * C * 4
    parts["knm_shards"] = b * kb
    # (lookahead + 1 groups of factors: the one being fitted and the ones being built ahead of it; a chain workspace per side
    # stream the chains run on)
    parts["factors_two_groups"] = (lookahead + 1) * g * factor_bytes(M)
    parts["chain_workspace"] = chain_sides * g * chain_workspace_bytes(M, D)
    # the centres (rows, packed split, norms) of every class of two chain groups: every rank builds the K_nM shard of
    # every class of a batch, so it holds all of a group's centres, not only those of the classes it owns
    parts["centres_two_groups"] = (lookahead + 1) * g * world * M * (ldx * 4 + _round_up(D, 64) * 4 + 4)
    parts["kernel_workspaces"] = (2 * n_loc * ((M + 511) // 512) * 8              # fused scoring: f64 partials per column group
                                  + ((n_loc + 255) // 256) * _round_up(M, 4) * 8   # right-hand side out of the build: one slab row per row block
                                  + 2 * 512 * _round_up(M, 4) * 8                  # pass slabs
                                  + 4 * world * _round_up(M, 2) * 8 * 2)           # exchanged (world, M) matrices
    parts["reserve"] = RESERVE_BYTES
    return n_loc, parts


def plan_lockstep(N, D, M, C, world, hbm_bytes=HBM_BYTES_MI355X, budget_fraction=BUDGET_FRACTION, storage="auto", gauss="h2",
                  knm_format=None, knm_bytes=None, batch=0, chain=0, rows_resident=True, lookahead=1, chain_sides=1):
    """Choose (b, g) for C classes on N rows sharded over `world` ranks: the largest divisor b of `world` — then the largest
    g <= 6 — whose planned bytes per rank stay within budget_fraction x hbm_bytes.  `batch` / `chain` > 0 pin b / g (the plan
    then only reports whether they fit).  knm_format / knm_bytes: the backend's own rule and byte count (HipBackend), the
    documented rule otherwise.  Returns a JobPlan; `feasible` False when not even b = g = 1 fits (the job needs more ranks)."""
    world, C = int(world), int(C)
    n_loc = (int(N) + world - 1) // world
    fmt = knm_format(n_loc, M) if knm_format is not None else knm_format_rule(n_loc, M, storage)
    budget = budget_fraction * hbm_bytes
    bs = [int(batch)] if batch else divisors_desc(world)
    if any(world % b for b in bs):
        raise ValueError("plan_lockstep: the lock-step batch (%d) must divide the world size (%d)" % (bs[0], world))
    last = None
    for b in bs:
        n_rounds = max(1, ((C + b - 1) // b + world // b - 1) // (world // b))
        gs = [int(chain)] if chain else list(range(max(1, min(MAX_CHAIN_CLASSES, n_rounds)), 0, -1))
        for g in gs:
            n_loc, parts = _parts(N, D, M, C, world, b, g, fmt, gauss, knm_bytes, rows_resident, lookahead, chain_sides)
            total = float(sum(parts.values()))
            last = JobPlan(world, b, g, n_loc, fmt, total <= budget, budget, total, parts)
            if last.feasible:
                return last
    return last


def device_hbm_bytes(device=None):
    """Total memory of the HIP device the job runs on (the planner's budget base); the MI355X figure without one."""
    try:
        import torch
        if torch.cuda.is_available():
            return float(torch.cuda.get_device_properties(torch.cuda.current_device() if device is None else device).total_memory)
    except Exception:
        pass
    return HBM_BYTES_MI355X
