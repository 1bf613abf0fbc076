"""The one options table of the product (round-5 review, item 7).

Every switch the package has lives here, with the value the committed benchmark ran on as its default.  The table is filled
ONCE — when `odx.get_backend()` first creates the backend — from the environment variables named in `ENV` (kept so that a
deployment can still be configured from outside the process; nothing else in the package or the library reads `os.environ` /
`getenv` for behaviour), pushed into libodx through `odx_set_option`, and from then on changed only through `set()` /
`override()`.  `as_dict()` is what `bench.py` prints into its `config`.  tests/test_abi.py pins the defaults.

Not options: `ODX_SAMPLES` / `ODX_MODEL` (odx/providers.py: how an UNMODIFIED reference driver, which passes no cfg_options, is
handed its image stream and network — the integration contract) and `ODX_LIB_PATH` (where hip.py finds
libodx.so).
"""
import contextlib
import os
from dataclasses import asdict, dataclass, fields


@dataclass
class Options:
    # ---- FALKON kernels (odx/backend.py, csrc/gauss*.hip, knm_pass*.hip, dense_f64.hip)
    gauss: str = "h2"                # "h2": X Z' on the f16 matrix cores via the two-term split (f32 accuracy) | "f32": all-f32 MFMA
    #                                  | "f8": BASELINE config 5's e4m3 contraction (throughput only)
    knm_storage: str = "auto"        # stored K_nM: "auto" (24-bit fixed point where the passes are HBM-bound, f32 below) | "f32" | "u24"
    #                                  | "bf16" (config 2's throughput-only storage)
    precond: str = "auto"            # A factor of the preconditioner on the split-f16 core: "auto" (from 4096 centres on) | "f64" | "split"
    h2_tile: int = 0                 # tile core of the split kernels: 0 automatic | 128 | 256
    chain_helpers: int = -1          # helper streams of the factorisation chains: -1 automatic (from 4096 rows on) | 0 | 1
    chain_split_min: int = 4         # classes from which fit_batch splits a class-batched chain into two half chains
    # ---- feature forward (odx/extract.py, odx/fpn.py)
    trunk: str = "rows"              # "rows": trunk stages / RPN head as row GEMMs on the library's tile cores | "conv": convolution library
    rows_min_positions: int = 3600   # stride-16 positions per call from which the row-GEMM trunk is used (below: convolution library)
    trunk_graph: bool = True         # replay the trunk from a HIP graph per image size
    group_graph: bool = True         # replay a group's whole forward from one HIP graph
    # ---- host plumbing
    stream_probe: str = "host"       # how odx/streams.py decides whether two streams share a hardware queue: "host" | "events"
    staged_uploads: bool = False     # small host tensors through a page-locked staging block (odx/harvest.py)
    # ---- Minibootstrap schedule (odx/region_classifier.py; each also an opts[...] key of trainRegionClassifier)
    class_batch: int = 0             # classes advanced together with batched chains (0: the default rule)
    class_streams: int = 0           # classes on streams of their own
    reference_order: str = "auto"    # "auto" | "sequential": the reference's class-by-class loop
    class_shard: bool = False        # classes round-robin over the ranks of torch.distributed


# environment variable -> field, read once by load()
ENV = {
    "ODX_GAUSS": "gauss", "ODX_KNM": "knm_storage", "ODX_PRECOND": "precond", "ODX_H2_TILE": "h2_tile", "ODX_CHAIN_HELPERS": "chain_helpers",
    "ODX_CHAIN_SPLIT_MIN": "chain_split_min", "ODX_TRUNK": "trunk", "ODX_ROWS_MIN_POSITIONS": "rows_min_positions",
    "ODX_TRUNK_GRAPH": "trunk_graph", "ODX_GROUP_GRAPH": "group_graph", "ODX_STREAM_PROBE": "stream_probe",
    "ODX_STAGED_UPLOADS": "staged_uploads", "ODX_CLASS_BATCH": "class_batch", "ODX_CLASS_STREAMS": "class_streams",
    "ODX_REFERENCE_ORDER": "reference_order", "ODX_CLASS_SHARD": "class_shard",
}

_CHOICES = {"gauss": ("h2", "f32", "f8"), "knm_storage": ("auto", "f32", "u24", "bf16"), "precond": ("auto", "f64", "split"),
            "h2_tile": (0, 128, 256), "chain_helpers": (-1, 0, 1), "trunk": ("rows", "conv"), "stream_probe": ("host", "events"),
            "reference_order": ("auto", "sequential")}
_PRECOND_CODE = {"auto": 0, "f64": 1, "split": 2}

_current = Options()
_loaded = False


def _coerce(name, value):
    kind = type(getattr(Options(), name))
    if kind is bool:
        if isinstance(value, str):
            value = value.strip().lower() not in ("", "0", "false", "no", "off")
        value = bool(value)
    elif kind is int:
        value = int(value)
    else:
        value = str(value)
        if name == "precond" and value[:1] in ("f", "s"):          # (historical spellings: anything starting with f / s)
            value = "f64" if value[0] == "f" else "split"
    if name in _CHOICES and value not in _CHOICES[name]:
        raise ValueError("odx.options: %s must be one of %r, got %r" % (name, _CHOICES[name], value))
    return value


def current():
    """The live table (an Options instance).  Read fields from it; change them with set() / override()."""
    return _current


def load(environ=None):
    """Fill the table from the environment variables of ENV (once: later calls return the table as it is)."""
    global _loaded
    if not _loaded:
        env = os.environ if environ is None else environ
        for var, name in ENV.items():
            if env.get(var, "") != "":
                setattr(_current, name, _coerce(name, env[var]))
        _loaded = True
    return _current


def reset():
    """Back to the defaults, environment forgotten (tests)."""
    global _current, _loaded
    _current, _loaded = Options(), False
    _push_library()


def _push_library():
    """The library-side options follow the table (only when libodx is loaded: nothing here loads it)."""
    from . import hip
    lib = hip._LIB
    if lib is None:
        return
    for name, value in (("h2_tile", _current.h2_tile), ("precond", _PRECOND_CODE[_current.precond]), ("chain_helpers", _current.chain_helpers)):
        hip.check(lib.odx_set_option(name.encode(), int(value)), "odx_set_option(%s)" % name)


def apply():
    """Push the library-side options of the table into libodx (h2_tile, precond, chain_helpers)."""
    _push_library()


def set(**kw):            # noqa: A001 — odx.options.set(...) reads as what it does
    """Change options of the live table (validated) and push them to the library / the live backend."""
    for name, value in kw.items():
        if name not in {f.name for f in fields(Options)}:
            raise AttributeError("odx.options: no option %r" % name)
        setattr(_current, name, _coerce(name, value))
    _push_library()
    from . import backend as _b
    be = _b._BACKEND
    if be is not None and getattr(be, "name", "") == "hip-gfx950":       # (gauss / knm_storage live on the backend object)
        for name in ("gauss", "knm_storage"):
            if name in kw:
                setattr(be, name, getattr(_current, name))


@contextlib.contextmanager
def override(**kw):
    """`with odx.options.override(precond="split"): ...` — the options changed inside the block, restored behind it."""
    old = {k: getattr(_current, k) for k in kw}
    set(**kw)
    try:
        yield _current
    finally:
        set(**old)


def library_hook(name, value):
    """The library's TEST HOOKS (include/odx.h: rls_force_nt_gram, rls_force_inverse_solve) — not options of the product."""
    from . import hip
    hip.check(hip.load().odx_set_option(name.encode(), int(value)), "odx_set_option(%s)" % name)


def as_dict():
    return asdict(_current)
