"""HIP backend: torch tensors in, raw pointers into libodx.so, torch tensors out.

This is the ONLY compute backend of the product.  PyTorch-ROCm is plumbing here (device
memory, the current HIP stream, torch.distributed); no torch arithmetic stands in for a
kernel.  ``tests/`` drive the same host logic (solver.py, dist.py, the drop-in modules) with
a numpy-oracle backend of the same interface to exercise it on CPU / gloo.

Interface used by the host logic (any backend provides exactly this):
    features(X) -> Features          rows(F, idx) -> Features
    precond(Zf, sigma, lam, eps) -> Precond
    knm(F, Zf, sigma) -> Knm         ktk(K, v=None, w=None) -> vec(M) f64
    trmv(P, name, x, alpha=1, beta=0, z=None) -> vec(M) f64
    cg_init / cg_step / cg_finish / axpby on f64 vectors with a device-side state
    mmv(F, Zf, sigma, V, ranges) -> (n, T) f32
    vec(x) / zeros(n) f64 vectors on the backend's device
"""
import ctypes
import os

import torch

from . import hip


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class Features:
    """Row-major f32 rows (n x D, leading dimension a multiple of 4) plus their squared norms and, once a
    Gaussian-kernel call has needed it, the packed two-term f16 split of the rows (P, meta: odx_split_f16)."""
    __slots__ = ("X", "sq", "n", "D", "ld", "P", "meta", "own_pack", "P8", "meta8", "sq8", "zero_row")

    def __init__(self, X, sq, D, P=None, meta=None):
        self.X, self.sq, self.n, self.D, self.ld = X, sq, X.shape[0], D, X.stride(0) if X.shape[0] else X.shape[1]
        self.P, self.meta = P, meta
        self.zero_row = False           # P is followed by one all-zero row in memory (what the tap-gathering products read outside the map)
        self.own_pack = P is None       # False: P was gathered from another matrix's split and carries ITS scale
        self.P8 = self.meta8 = self.sq8 = None     # e4m3 packing (odx_split_f8), made on demand by the throughput-only fp8 kernels


class Rows16:
    """A 16-bit (bf16 / f16) row matrix as odx_gemm_b16 takes it: buf (n, ld), ld a multiple of 128 elements, columns K .. ld
    zero.  `dense` is the (n, K) view."""
    __slots__ = ("buf", "n", "K", "zero_row")

    def __init__(self, buf, K, zero_row=False):
        self.buf, self.n, self.K = buf, buf.shape[0], K
        self.zero_row = zero_row            # buf is followed by one all-zero row in memory (odx_gemm_b16_taps reads it outside the map)

    @property
    def dense(self):
        return self.buf[:, :self.K]


class PackedRows:
    """An activation matrix (n, D) of a chain of GEMM layers: P / meta = its packed two-term f16 split as a layer's epilogue
    wrote it (odx_gemm_h2_chain_f32; meta[0] the scale, meta[1] max |x|), X = the f32 matrix when the layer wrote one (a block's
    output: the next block's identity branch), zero_row = P is followed by one all-zero row in memory (odx_gemm_h2_taps_f32)."""
    __slots__ = ("X", "n", "D", "P", "meta", "zero_row")

    def __init__(self, X, n, D, P, meta, zero_row=False):
        self.X, self.n, self.D, self.P, self.meta, self.zero_row = X, n, D, P, meta, zero_row


class Precond:
    """Inverse Cholesky factors of the FALKON preconditioner, f64, row-major (M x ld)."""
    __slots__ = ("LTi", "LTit", "LAi", "LAit", "M", "ld", "info", "block_rows")


class Knm:
    """A stored K_nM shard.  fmt "f32": K (n, ld) f32.  "u24": 24-bit fixed point — K (n, ld) int16 holds q >> 8 and lo
    (n, ld) uint8 holds q & 255, q = round(K 2^24).  "bf16": K (n, ld) int16 holds the bf16 bit patterns.  (include/odx.h,
    "compact storage of the stored K_nM")."""
    __slots__ = ("K", "n", "M", "ld", "fmt", "lo")

    def __init__(self):
        self.fmt, self.lo = "f32", None

    def dense(self):
        """The block as an (n, M) f32 tensor (tests, diagnostics; exact for every format)."""
        if self.fmt == "f32":
            return self.K[:, :self.M]
        if self.fmt == "u24":
            q = ((self.K[:, :self.M].to(torch.int32) & 0xFFFF) << 8) | self.lo[:, :self.M].to(torch.int32)
            return q.to(torch.float32) * (2.0 ** -24)
        return ((self.K[:, :self.M].to(torch.int32) & 0xFFFF) << 16).view(torch.float32)

    def rows(self, lo, hi):
        """The sub-block of rows [lo, hi) as a view."""
        sub = Knm()
        sub.n, sub.M, sub.ld, sub.fmt = hi - lo, self.M, self.ld, self.fmt
        sub.K = self.K[lo:hi]
        sub.lo = self.lo[lo:hi] if self.lo is not None else None
        return sub


_KNM_CODE = {"f32": hip.KNM_F32, "u24": hip.KNM_U24, "bf16": hip.KNM_BF16}


class HipBackend:
    name = "hip-gfx950"

    def __init__(self, device=None):
        self.lib = hip.require_gpu()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._ws = {}
        self._meta_pools = {}
        self._capture_meta_pool = None
        # "h2": X Z' of the Gaussian kernels on the f16 matrix cores via the two-term split (f32 accuracy);
        # "f32": the all-f32 MFMA chain.  Both are HIP paths of libodx; there is no other.
        # ("f8": BASELINE config 5's e4m3 contraction — throughput only, K entries ~1e-3 off; never a default)
        from . import options as _options
        opts = _options.load()                 # the one options table (odx/options.py): filled once, here
        self.gauss = opts.gauss
        # Storage of the K_nM block the CG passes stream (include/odx.h): "f32" always f32; "u24" / "bf16" always that compact
        # format (f16-split kernels only); "auto" (default): 24-bit fixed point for the blocks whose passes are HBM-bound
        # (>= 2^27 entries and at least 1024 centres on the wide tile core: the headline, config 5's shards, configs 2 and
        # 4), f32 below — the fits of a Minibootstrap round (<= 22 000 x 2000) gain nothing.  The one-call / class-batched CG
        # loops stream either (odx_falkon_cg_batched_f64 / _q_f64).  tools/precision_storage_study.py: alpha against the f64 evaluation is the same
        # with u24 as with f32 storage; bf16 is BASELINE config 2's throughput-only storage (alpha off by 1e-2..6e-1).
        self.knm_storage = opts.knm_storage
        _options.apply()                       # h2_tile / precond / chain_helpers -> libodx (odx_set_option)
        # the process's side streams are picked NOW, while the process is fresh (odx/streams.py: which hardware queue a stream
        # lands on depends on the streams the process has used before; picked behind a big job they landed worse — the
        # Minibootstrap read 0.51 instead of 0.43 s, the harvest loop 4.5 instead of 3.6 ms per image; ~50 ms, once)
        # ... and two of them become the library's helper streams (odx_set_helper_streams: a helper the library creates by itself
        # may land on the main stream's hardware queue, which serialises a chain's GEMMs behind every K_nM build)
        self.helper_streams = []
        try:
            from . import streams as _streams
            own = _streams.of_default(3, self.device)
            if len(own) >= 3:
                self.helper_streams = own[1:3]
                hip.check(self.lib.odx_set_helper_streams(ctypes.c_void_p(own[1].cuda_stream), ctypes.c_void_p(own[2].cuda_stream)),
                          "odx_set_helper_streams")
        except hip.OdxError:
            raise
        except Exception:          # noqa: BLE001 — a heuristic: never a reason not to start
            pass

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, key, nbytes):
        """Scratch buffer of a call site, one per stream it has been used on (fits issued on different streams must not
        share scratch; the main stream keeps the bare key)."""
        nbytes = max(int(nbytes), 16)
        if torch.cuda.is_current_stream_capturing():
            # inside a HIP-graph capture (extract.GraphedCall): the kernels being recorded keep this address for as long as the graph
            # is replayed, so the buffer must belong to the GRAPH — a fresh allocation from the capture's private pool — and never
            # to the shared table, whose entries are replaced when a later call asks for more (use after free on the next replay)
            return torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        if cur != torch.cuda.default_stream(self.device):
            key = (key, cur.cuda_stream)
        buf = self._ws.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = None
            self._ws.pop(key, None)
            buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[key] = buf
        return buf

    def release_workspaces(self):
        """Drops the scratch buffers and the library's internal helper streams of this thread (both come back on demand): what
        a process calls between its fits and latency-bound work (odx_release_helper_streams says why)."""
        self._ws.clear()
        self.release_helper_streams()

    def release_helper_streams(self):
        """The library's internal helper streams of this thread's factorisation chains are destroyed (no wait; remade on demand):
        called when a fit / a training step has been queued — left alive they slow every later small launch of the process
        (odx_release_helper_streams)."""
        hip.check(self.lib.odx_release_helper_streams(), "odx_release_helper_streams")

    def vec(self, x):
        return torch.as_tensor(x, dtype=torch.float64, device=self.device).contiguous()

    def zeros(self, n, dtype=torch.float64):
        return torch.zeros(n, dtype=dtype, device=self.device)

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------ features
    def row_matrix(self, X):
        """features() without the squared norms: the layout contract only, for the callers that never form a kernel block
        of these rows (the RLS box regressors: a read of all rows saved)."""
        return self.features(X, norms=False)

    def features(self, X, norms=True):
        """Adopt (n, D) rows as kernel operands: f32, row stride a multiple of 4 elements,
        16-byte aligned (copied only when the caller's tensor is not already so)."""
        if not torch.is_tensor(X):
            X = torch.as_tensor(X)
        X = X.to(device=self.device, dtype=torch.float32)
        if X.dim() != 2:
            raise ValueError("features: expected a 2-D (n, D) tensor, got shape %s" % (tuple(X.shape),))
        n, D = X.shape
        ok = X.stride(1) == 1 and (n == 0 or (X.stride(0) % 4 == 0 and X.stride(0) >= D)) and X.data_ptr() % 16 == 0
        if not ok:
            ld = (D + 3) // 4 * 4
            buf = torch.zeros((n, ld), dtype=torch.float32, device=self.device)
            buf[:, :D] = X
            X = buf[:, :D]
        if not norms:
            return Features(X, None, D)
        sq = torch.empty(n, dtype=torch.float32, device=self.device)
        if n and self.gauss == "h2":
            # the norms and, from the same read of the rows, the maximum the f16 split scales by (pack() then only packs)
            meta = self._meta_slot()
            hip.check(self.lib.odx_row_sqnorm_absmax_f32(_p(X), X.stride(0), n, D, _p(sq), _p(meta), self._stream()),
                      "odx_row_sqnorm_absmax_f32")
            F = Features(X, sq, D)
            F.meta = meta
            return F
        if n:
            hip.check(self.lib.odx_row_sqnorm_f32(_p(X), X.stride(0), n, D, _p(sq), self._stream()), "odx_row_sqnorm_f32")
        return Features(X, sq, D)

    def masked_stream(self, cus, complement=False):
        """A stream confined to `cus` compute units, spread evenly over the XCDs (logical CU i sits on XCC i % 8, so the
        first `cus` bits are cus / 8 per XCD); the library's internal helper streams created from now on get the same
        mask.  complement: the stream gets the OTHER total - cus units instead (and the helper streams' mask is left alone) —
        the MFMA-bound Gaussian kernels on the part of the chip the factorisation chains are not confined to.  Kept alive by
        the backend."""
        total = int(self.lib.odx_device_cus())
        cus = max(8, min(int(cus), total - 8 if complement else total))
        words = (total + 31) // 32
        arr = (ctypes.c_uint32 * words)()
        for b in (range(cus, total) if complement else range(cus)):
            arr[b // 32] |= 1 << (b % 32)
        if not complement:
            hip.check(self.lib.odx_set_side_stream_cu_mask(arr, words), "odx_set_side_stream_cu_mask")
        raw = ctypes.c_void_p()
        hip.check(self.lib.odx_stream_create_cu_mask(arr, words, ctypes.byref(raw)), "odx_stream_create_cu_mask")
        st = torch.cuda.ExternalStream(raw.value, device=self.device)
        self._masked_streams = getattr(self, "_masked_streams", []) + [(st, raw)]
        return st

    def set_helper_cus(self, cus):
        """The CU mask the library's internal helper streams are CREATED with from now on (first `cus` compute units; 0: the
        whole device).  Streams that exist keep theirs (odx_set_side_stream_cu_mask)."""
        cus = int(cus)
        if cus <= 0:
            hip.check(self.lib.odx_set_side_stream_cu_mask(None, 0), "odx_set_side_stream_cu_mask")
            return
        total = int(self.lib.odx_device_cus())
        words = (total + 31) // 32
        arr = (ctypes.c_uint32 * words)()
        for b in range(max(8, min(cus, total))):
            arr[b // 32] |= 1 << (b % 32)
        hip.check(self.lib.odx_set_side_stream_cu_mask(arr, words), "odx_set_side_stream_cu_mask")

    def _meta_slot(self):
        """Two zeroed floats (scale, max |x| bits) for one matrix, cut from a pool that is zeroed once per 4096 matrices —
        a fill launch per matrix otherwise (a Minibootstrap round: ~250 of them).  A slot is handed out once; the slices
        keep their pool alive."""
        if torch.cuda.is_current_stream_capturing():          # (see _workspace: memory a captured graph refers to is the graph's own)
            # one zero fill per 256 matrices of the capture (a trunk run as row GEMMs asks for ~60 slots per forward); the pool
            # is the graph's own memory and its fill a node of that graph: extract.GraphedCall drops it around every capture
            pool = self._capture_meta_pool
            if pool is None or pool[1] + 2 > pool[0].numel():
                pool = self._capture_meta_pool = [torch.zeros(1024, dtype=torch.float32, device=self.device), 0]
            at = pool[1]
            pool[1] = at + 4
            return pool[0][at:at + 2]
        cur = torch.cuda.current_stream(self.device)
        key = cur.cuda_stream
        pool = self._meta_pools.get(key)
        if pool is None or pool[1] + 2 > pool[0].numel():
            pool = self._meta_pools[key] = [torch.zeros(8192, dtype=torch.float32, device=self.device), 0]
        at = pool[1]
        pool[1] = at + 4                      # 16-byte slots
        return pool[0][at:at + 2]

    def reset_capture_pools(self):
        """Called by whoever captures a HIP graph, before and after the capture: pooled memory handed out during one capture
        must not serve the next one (it belongs to the first graph, and so does the launch that zeroes it)."""
        self._capture_meta_pool = None

    def rows(self, F, idx):
        """MyCenterSelector.select: gather rows (their norms, and their packed split if it exists) by index."""
        idx = torch.as_tensor(idx, dtype=torch.int64, device=self.device).reshape(-1)
        ld = (F.D + 3) // 4 * 4
        buf = torch.zeros((idx.numel(), ld), dtype=torch.float32, device=self.device)
        buf[:, :F.D] = F.X.index_select(0, idx)
        P = F.P.index_select(0, idx) if F.P is not None else None
        Z = Features(buf[:, :F.D], F.sq.index_select(0, idx), F.D, P, F.meta if P is not None else None)
        if F.P8 is not None:
            Z.P8, Z.meta8, Z.sq8 = F.P8.index_select(0, idx), F.meta8, F.sq8.index_select(0, idx)
        return Z

    def pack8(self, F):
        """Make sure F carries its e4m3 packing (odx_split_f8); returns F."""
        if F.P8 is None:
            ldp8 = (F.D + 127) // 128 * 128
            F.P8 = torch.empty((F.n, ldp8), dtype=torch.uint8, device=self.device)
            F.meta8 = torch.zeros(2, dtype=torch.float32, device=self.device)
            F.sq8 = torch.empty(F.n, dtype=torch.float32, device=self.device)
            hip.check(self.lib.odx_split_f8(_p(F.X), F.ld, F.n, F.D, _p(F.P8), ldp8, _p(F.meta8), _p(F.sq8), self._stream()),
                      "odx_split_f8")
        return F

    def pack(self, F, zero_row=False):
        """Make sure F carries its packed f16 split (odx_split_f16); returns F.  zero_row: the packed rows are followed by
        one all-zero row in memory (what gemm_h2_taps reads for a position outside the map)."""
        if F.P is not None and zero_row and not F.zero_row:
            # a packing that exists but lacks the trailing zero row (advisor, round 5: it was silently handed on and the tap
            # gather read past its end): packed again with the row when the split is this matrix's own, refused otherwise
            if not F.own_pack:
                raise ValueError("pack(zero_row=True): the packed rows were gathered from another matrix and have no zero row behind them")
            F.P = None
        if F.P is None:
            ldp = (F.D + 63) // 64 * 64
            if zero_row:
                buf = torch.empty((F.n + 1, ldp), dtype=torch.int32, device=self.device)
                buf[F.n].zero_()
                F.P = buf[:F.n]
                F.zero_row = True
            else:
                F.P = torch.empty((F.n, ldp), dtype=torch.int32, device=self.device)
            if F.meta is not None and F.n:            # features() already left max |x| in meta[1]
                hip.check(self.lib.odx_split_f16_premax(_p(F.X), F.ld, F.n, F.D, _p(F.P), ldp, _p(F.meta), self._stream()),
                          "odx_split_f16_premax")
            else:
                F.meta = torch.zeros(2, dtype=torch.float32, device=self.device)
                hip.check(self.lib.odx_split_f16(_p(F.X), F.ld, F.n, F.D, _p(F.P), ldp, _p(F.meta), self._stream()),
                          "odx_split_f16")
        return F

    # ------------------------------------------------------------------ FALKON pieces
    def precond(self, Zf, sigma, lam, eps, out=None, ws_key="precond"):
        """`out`: optional preallocated (4, M, ld) f64 tensor (e.g. owned by another stream); `ws_key` names the
        scratch buffer, so that factorisations in flight on different streams do not share one."""
        M, D = Zf.n, Zf.D
        ld = (M + 1) // 2 * 2
        P = Precond()
        P.M, P.ld = M, ld
        mats = out if out is not None else torch.empty((4, M, ld), dtype=torch.float64, device=self.device)
        P.LTi, P.LTit, P.LAi, P.LAit = mats[0], mats[1], mats[2], mats[3]
        P.block_rows = M                     # the four factors are one (4, M, ld) block: a class batch of one (cg_solve_batched)
        P.info = torch.zeros(1, dtype=torch.int32, device=self.device)
        nbytes = self.lib.odx_falkon_precond_workspace_bytes(M, D)
        ws = self._workspace(ws_key, nbytes)
        hip.check(self.lib.odx_falkon_precond_f64(_p(Zf.X), Zf.ld, M, D, float(sigma), float(lam), float(eps),
                                                  _p(P.LTi), _p(P.LTit), _p(P.LAi), _p(P.LAit), ld, _p(P.info),
                                                  _p(ws), ws.numel(), self._stream()), "odx_falkon_precond_f64")
        return P

    MAX_CLASS_BATCH = 32        # ODX_MAX_ZBATCH of libodx

    def precond_batched(self, Zfs, sigma, lam, eps, out=None, ws_key="precond_batched", Mmax=None):
        """The preconditioners of len(Zfs) <= 32 independent classes with ONE chain of launches
        (odx_falkon_precond_batched_f64): returns one Precond per class, views into a shared (B, 4, Mmax, ld) f64 block
        (`out` when given) whose leading M_b x M_b blocks equal, bit for bit, what precond() makes for that class.  Mmax: rows
        of a class's slot when that is more than this call's largest class (two calls filling halves of one block)."""
        B = len(Zfs)
        if not 1 <= B <= self.MAX_CLASS_BATCH:
            raise ValueError("precond_batched: 1..%d classes per call, got %d" % (self.MAX_CLASS_BATCH, B))
        D = Zfs[0].D
        if any(z.D != D for z in Zfs):
            raise ValueError("precond_batched: every class needs the same feature dimension")
        Ms = [int(z.n) for z in Zfs]
        if Mmax is not None and int(Mmax) < max(Ms):
            raise ValueError("precond_batched: Mmax (%d) below the largest class (%d)" % (int(Mmax), max(Ms)))
        Mmax = max(Ms) if Mmax is None else int(Mmax)
        ld = (Mmax + 1) // 2 * 2
        if out is None:
            out = torch.empty((B, 4, Mmax, ld), dtype=torch.float64, device=self.device)
        if out.dtype != torch.float64 or out.dim() != 4 or out.shape[0] < B or tuple(out.shape[1:]) != (4, Mmax, ld) or not out.is_contiguous():
            raise ValueError("precond_batched: out must be a contiguous (>= %d, 4, %d, %d) f64 tensor" % (B, Mmax, ld))
        info = torch.zeros(B, dtype=torch.int32, device=self.device)
        ws = self._workspace(ws_key, self.lib.odx_falkon_precond_batched_workspace_bytes(Mmax, D, B))
        zp = (ctypes.c_void_p * B)(*[z.X.data_ptr() for z in Zfs])
        zl = (ctypes.c_int64 * B)(*[int(z.ld) for z in Zfs])
        zm = (ctypes.c_int64 * B)(*Ms)
        hip.check(self.lib.odx_falkon_precond_batched_f64(zp, zl, zm, B, Mmax, D, float(sigma), float(lam), float(eps), _p(out), ld,
                                                          4 * Mmax * ld, _p(info), _p(ws), ws.numel(), self._stream()),
                  "odx_falkon_precond_batched_f64")
        Ps = []
        for b, M in enumerate(Ms):
            P = Precond()
            P.M, P.ld = M, ld
            P.LTi, P.LTit, P.LAi, P.LAit = (out[b, k, :M] for k in range(4))
            P.info = info[b:b + 1]
            P.block_rows = Mmax                  # rows of each factor's slot in the shared block (cg_solve_batched)
            Ps.append(P)
        return Ps

    def check_precond(self, P):
        self.check_info(P.info)

    def check_info(self, info):
        """One host synchronisation: raise if a Cholesky of the preconditioner met a non-positive pivot."""
        info = int(info.item())
        if info != 0:
            raise hip.OdxError("FALKON preconditioner: non-positive pivot at index %d (Cholesky failed)" % (info - 1))

    def knm_format(self, n, M):
        """Storage format a K_nM block of this shape gets (see __init__)."""
        if self.gauss not in ("h2", "f8") or n <= 0 or M <= 0:
            return "f32"
        if self.knm_storage in ("u24", "bf16"):
            return self.knm_storage if self.lib.odx_knm_fwd_bwd_q_workspace_bytes(n, M, _KNM_CODE[self.knm_storage]) >= 0 else "f32"
        if self.knm_storage == "auto" and n * M >= (1 << 27) and M >= 1024 and (self.gauss == "f8" or self.lib.odx_gauss_h2_tile(n, M) == 256) \
                and self.lib.odx_knm_fwd_bwd_q_workspace_bytes(n, M, hip.KNM_U24) >= 0:
            return "u24"
        return "f32"

    def knm_bytes(self, n, M):
        """Bytes of the K_nM block knm() / knm_rhs() make for this shape (for preallocated `out` buffers)."""
        return int(self.lib.odx_knm_bytes(max(n, 0), M, _KNM_CODE[self.knm_format(n, M)]))

    def _knm_block(self, n, M, fmt, out):
        K = Knm()
        K.n, K.M, K.fmt = n, M, fmt
        K.ld = ld = int(self.lib.odx_knm_ld(M, _KNM_CODE[fmt]))
        per = {"f32": 4, "u24": 3, "bf16": 2}[fmt]
        need = n * ld * per
        if out is not None:
            raw = out.view(-1).view(torch.uint8)
            if raw.numel() < need or raw.data_ptr() % 16:
                raise ValueError("knm: out buffer too small (%d < %d bytes) or not 16-byte aligned" % (raw.numel(), need))
        else:
            raw = torch.empty(max(need, 16), dtype=torch.uint8, device=self.device)
        if fmt == "f32":
            K.K = raw[: n * ld * 4].view(torch.float32).view(n, ld)
        else:
            K.K = raw[: n * ld * 2].view(torch.int16).view(n, ld)
            if fmt == "u24":
                K.lo = raw[n * ld * 2: n * ld * 3].view(n, ld)
        return K

    def knm(self, F, Zf, sigma, out=None):
        n, M = F.n, Zf.n
        fmt = self.knm_format(n, M)
        if fmt != "f32" or self.gauss == "f8":
            return self._knm_store(F, Zf, sigma, fmt, None, out, None)[0]
        K = self._knm_block(n, M, "f32", out)
        ld = K.ld
        if self.direct_small(n, M, F.D):
            # a toy-sized block: every entry from direct differences summed in f64, exactly rounded (odx_gauss_knm_direct_f32)
            hip.check(self.lib.odx_gauss_knm_direct_f32(_p(F.X), F.ld, n, _p(Zf.X), Zf.ld, M, F.D, float(sigma), _p(K.K), ld,
                                                        self._stream()), "odx_gauss_knm_direct_f32")
            return K
        if self.gauss == "h2":
            self.pack(F), self.pack(Zf)
            hip.check(self.lib.odx_gauss_knm_h2(_p(F.P), F.P.stride(0), _p(F.meta), _p(F.sq), n, _p(Zf.P), Zf.P.stride(0),
                                                _p(Zf.meta), _p(Zf.sq), M, F.D, float(sigma), _p(K.K), ld, self._stream()),
                      "odx_gauss_knm_h2")
        else:
            hip.check(self.lib.odx_gauss_knm_f32(_p(F.X), F.ld, _p(F.sq), n, _p(Zf.X), Zf.ld, _p(Zf.sq), M, F.D,
                                                 float(sigma), _p(K.K), ld, self._stream()), "odx_gauss_knm_f32")
        return K

    DIRECT_MAX_MACS = 1 << 24

    def direct_small(self, n, M, D):
        """Whether an (n, M) block over D features is built by direct differences (exactly rounded entries): the default
        kernels (gauss "h2", tile core not pinned) and at most 2^24 multiply-adds — far below every problem of the reference's
        regime (its smallest fits are 1e4 rows x 500 centres x 256 features = 1.3e9)."""
        from . import options as _options
        return (self.gauss == "h2" and _options.current().h2_tile == 0 and n > 0 and M > 0 and self.knm_storage in ("auto", "f32")
                and int(n) * int(M) * int(D) <= self.DIRECT_MAX_MACS)

    def _knm_store(self, F, Zf, sigma, fmt, w, out, rhs_out):
        """Build on the wide tile core into storage format `fmt` (odx_gauss_knm_h2_store), with the fused right-hand side
        when w is given."""
        n, M = F.n, Zf.n
        K = self._knm_block(n, M, fmt, out)
        f8 = self.gauss == "f8"
        (self.pack8(F), self.pack8(Zf)) if f8 else (self.pack(F), self.pack(Zf))
        ws = None
        if w is not None:
            w = w.to(dtype=torch.float64, device=self.device).contiguous()
            if rhs_out is None:
                rhs_out = torch.empty(M, dtype=torch.float64, device=self.device)
            ws = self._workspace("knm_rhs", self.lib.odx_gauss_knm_h2_rhs_workspace_bytes(n, M))
        fn, PX, mx, PZ, mz = ((self.lib.odx_gauss_knm_f8_store, F.P8, F.meta8, Zf.P8, Zf.meta8) if f8 else
                              (self.lib.odx_gauss_knm_h2_store, F.P, F.meta, Zf.P, Zf.meta))
        sqx, sqz = (F.sq8, Zf.sq8) if f8 else (F.sq, Zf.sq)
        hip.check(fn(_p(PX), PX.stride(0), _p(mx), _p(sqx), n, _p(PZ), PZ.stride(0), _p(mz), _p(sqz), M, F.D, float(sigma),
                     _KNM_CODE[fmt], _p(K.K), K.ld, _p(K.lo), K.ld, _p(w), _p(rhs_out if w is not None else None), _p(ws),
                     ws.numel() if ws is not None else 0, self._stream()), "odx_gauss_knm_f8_store" if f8 else "odx_gauss_knm_h2_store")
        return K, rhs_out

    def knm_rhs(self, F, Zf, sigma, w, out=None, rhs_out=None):
        """K_nM and this shard's K_nM' w (the right-hand side of the fit) in one go.  With the f16-split kernels on the
        wide tile core the column sums come out of the build itself; otherwise (small blocks, ODX_GAUSS=f32) the build is
        followed by one pass over the stored block.  Returns (K, K'w)."""
        n, M = F.n, Zf.n
        if rhs_out is None:
            rhs_out = torch.empty(M, dtype=torch.float64, device=self.device)
        fmt = self.knm_format(n, M)
        if (fmt != "f32" or self.gauss == "f8") and n > 0:
            return self._knm_store(F, Zf, sigma, fmt, w, out, rhs_out)
        if not (self.gauss == "h2" and n > 0 and self.lib.odx_gauss_h2_tile(n, M) == 256) or self.direct_small(n, M, F.D):
            K = self.knm(F, Zf, sigma, out=out)
            return K, self.ktk(K, w=w, out=rhs_out)
        return self._knm_store(F, Zf, sigma, "f32", w, out, rhs_out)

    def pin_gauss_tile(self, tile):
        """Pin the tile core of the f16-split Gaussian kernels (128 or 256); 0 = chosen per launch (default)."""
        from . import options as _options
        _options.set(h2_tile=int(tile))

    def reserve_cus_during_passes(self, cus):
        """Leave `cus` CUs free while the persistent CG pass kernel runs, for work queued on other streams."""
        hip.check(self.lib.odx_set_pass_reserved_cus(int(cus)), "odx_set_pass_reserved_cus")

    def ktk(self, K, v=None, w=None, out=None):
        """out = K' (K v + w) over this shard (f64)."""
        if out is None:
            out = torch.empty(K.M, dtype=torch.float64, device=self.device)
        if K.fmt != "f32":
            code = _KNM_CODE[K.fmt]
            nbytes = self.lib.odx_knm_fwd_bwd_q_workspace_bytes(max(K.n, 1), K.M, code)
            if nbytes < 0:
                raise hip.OdxError("odx_knm_fwd_bwd_q: M = %d is outside the supported range" % K.M)
            ws = self._workspace("ktk", nbytes)
            hip.check(self.lib.odx_knm_fwd_bwd_q(_p(K.K), K.ld, _p(K.lo), K.ld, code, K.n, K.M, _p(v), _p(w), _p(out), _p(ws),
                                                 ws.numel(), self._stream()), "odx_knm_fwd_bwd_q")
            return out
        nbytes = self.lib.odx_knm_fwd_bwd_workspace_bytes(max(K.n, 1), K.M)
        if nbytes < 0:
            raise hip.OdxError("odx_knm_fwd_bwd: M = %d is outside the supported range" % K.M)
        ws = self._workspace("ktk", nbytes)
        hip.check(self.lib.odx_knm_fwd_bwd(_p(K.K), K.ld, K.n, K.M, _p(v), _p(w), _p(out), _p(ws), ws.numel(),
                                           self._stream()), "odx_knm_fwd_bwd")
        return out

    def _ktk2_bytes(self, K):
        if K.fmt != "f32":
            return self.lib.odx_knm_fwd_bwd2_q_workspace_bytes(max(K.n, 1), K.M, _KNM_CODE[K.fmt])
        return self.lib.odx_knm_fwd_bwd2_workspace_bytes(max(K.n, 1), K.M)

    def can_ktk2(self, K):
        """Whether the two-vector pass exists at this block's width (both vectors must fit in LDS: M <= 10 000)."""
        return self._ktk2_bytes(K) >= 0

    def ktk2(self, K, v1, v2, out1=None, out2=None):
        """out1 = K' (K v1), out2 = K' (K v2) over this shard from ONE read of K (odx_knm_fwd_bwd2[_q])."""
        if out1 is None:
            out1 = torch.empty(K.M, dtype=torch.float64, device=self.device)
        if out2 is None:
            out2 = torch.empty(K.M, dtype=torch.float64, device=self.device)
        nbytes = self._ktk2_bytes(K)
        if nbytes < 0:
            raise hip.OdxError("odx_knm_fwd_bwd2: M = %d is outside the two-vector configurations" % K.M)
        ws = self._workspace("ktk", nbytes)
        if K.fmt != "f32":
            hip.check(self.lib.odx_knm_fwd_bwd2_q(_p(K.K), K.ld, _p(K.lo), K.ld, _KNM_CODE[K.fmt], K.n, K.M, _p(v1), _p(v2),
                                                  _p(out1), _p(out2), _p(ws), ws.numel(), self._stream()), "odx_knm_fwd_bwd2_q")
            return out1, out2
        hip.check(self.lib.odx_knm_fwd_bwd2(_p(K.K), K.ld, K.n, K.M, _p(v1), _p(v2), _p(out1), _p(out2), _p(ws), ws.numel(),
                                            self._stream()), "odx_knm_fwd_bwd2")
        return out1, out2

    def cg_residual(self, B, AX, AP, state, R):
        """R = B - (AX + a AP), a = the step cg_step has just taken (state[3])."""
        hip.check(self.lib.odx_cg_residual(_p(B), _p(AX), _p(AP), _p(state), _p(R), R.numel(), self._stream()), "odx_cg_residual")

    def cg_solve(self, K, P, b0, n_total, lam, maxiter, opt):
        """The CG loop of an unsharded fit in one library call (odx_falkon_cg_f64); returns alpha (M,) f64.  f32-stored
        blocks only: the library loop streams K as floats (compact formats go through solver.falkon_fit's loop)."""
        if K.fmt != "f32":
            raise hip.OdxError("cg_solve: the one-call CG loop needs an f32-stored K_nM block, got %r" % K.fmt)
        alpha = torch.empty(K.M, dtype=torch.float64, device=self.device)
        nbytes = self.lib.odx_falkon_cg_workspace_bytes(max(K.n, 1), K.M)
        if nbytes < 0:
            raise hip.OdxError("odx_falkon_cg_f64: M = %d is outside the supported range" % K.M)
        ws = self._workspace("cg_solve", nbytes)
        hip.check(self.lib.odx_falkon_cg_f64(_p(K.K), K.ld, K.n, K.M, _p(P.LTi), _p(P.LTit), _p(P.LAi), _p(P.LAit), P.ld, _p(b0),
                                             float(n_total), float(lam), int(maxiter), int(opt.cg_full_gradient_every),
                                             float(opt.cg_epsilon), float(opt.cg_tolerance), _p(alpha), _p(ws), ws.numel(),
                                             self._stream()), "odx_falkon_cg_f64")
        return alpha

    def cg_batched_supported(self, ns, Ms, fmt="f32"):
        """Whether cg_solve_batched has a lock-step loop for blocks of these shapes stored as `fmt` (one pass configuration
        for all of them) — asked BEFORE the blocks are built."""
        B = len(ns)
        if not 1 <= B <= self.MAX_CLASS_BATCH:
            return False
        n = (ctypes.c_int64 * B)(*[int(v) for v in ns])
        M = (ctypes.c_int64 * B)(*[int(v) for v in Ms])
        if fmt == "f32":
            return self.lib.odx_falkon_cg_batched_workspace_bytes(B, n, M) >= 0
        return self.lib.odx_falkon_cg_batched_q_workspace_bytes(B, n, M, _KNM_CODE[fmt]) >= 0

    def cg_solve_batched(self, Ks, Ps, b0s, n_totals, lam, maxiter, opt):
        """The CG loops of len(Ks) <= 32 independent fits in lock step, one launch sequence for all of them
        (odx_falkon_cg_batched_f64).  Ps: the Preconds of ONE precond_batched call, in order (they share a block);
        b0s: (B, vstride) f64.  Returns alpha (B, vstride) f64 — row b's first M_b entries are class b's alpha, bit for
        bit what cg_solve gives — or None when the classes do not share a pass configuration / storage format.  Blocks in
        a compact format (24-bit fixed point, bf16) go through odx_falkon_cg_batched_q_f64."""
        B = len(Ks)
        fmt = Ks[0].fmt
        if any(k.fmt != fmt for k in Ks):           # one storage format per batch
            return None
        n = (ctypes.c_int64 * B)(*[int(k.n) for k in Ks])
        M = (ctypes.c_int64 * B)(*[int(k.M) for k in Ks])
        if fmt == "f32":
            nbytes = self.lib.odx_falkon_cg_batched_workspace_bytes(B, n, M)
        else:
            nbytes = self.lib.odx_falkon_cg_batched_q_workspace_bytes(B, n, M, _KNM_CODE[fmt])
        if nbytes < 0:
            return None
        base = Ps[0].LTi
        p_rows, ldp = Ps[0].block_rows, Ps[0].ld
        p_stride = 4 * p_rows * ldp
        for b, P in enumerate(Ps):
            if P.LTi.data_ptr() != base.data_ptr() + b * p_stride * 8 or P.ld != ldp:
                raise ValueError("cg_solve_batched: the preconditioners must be consecutive members of one precond_batched block")
        kp = (ctypes.c_void_p * B)(*[k.K.data_ptr() for k in Ks])
        kl = (ctypes.c_int64 * B)(*[int(k.ld) for k in Ks])
        nt = (ctypes.c_double * B)(*[float(x) for x in n_totals])
        alpha = torch.zeros_like(b0s)
        ws = self._workspace("cg_solve_batched", nbytes)
        if fmt != "f32":
            lp = (ctypes.c_void_p * B)(*[(k.lo.data_ptr() if k.lo is not None else 0) for k in Ks])
            hip.check(self.lib.odx_falkon_cg_batched_q_f64(B, kp, kl, lp, kl, _KNM_CODE[fmt], n, M, _p(base), ldp, p_rows, p_stride, _p(b0s),
                                                           b0s.stride(0), nt, float(lam), int(maxiter), int(opt.cg_full_gradient_every),
                                                           float(opt.cg_epsilon), float(opt.cg_tolerance), _p(alpha), _p(ws), ws.numel(),
                                                           self._stream()), "odx_falkon_cg_batched_q_f64")
            return alpha
        hip.check(self.lib.odx_falkon_cg_batched_f64(B, kp, kl, n, M, _p(base), ldp, p_rows, p_stride, _p(b0s), b0s.stride(0), nt,
                                                     float(lam), int(maxiter), int(opt.cg_full_gradient_every), float(opt.cg_epsilon),
                                                     float(opt.cg_tolerance), _p(alpha), _p(ws), ws.numel(), self._stream()),
                  "odx_falkon_cg_batched_f64")
        return alpha

    _TRI = {"LTi": 0, "LTit": 1, "LAi": 0, "LAit": 1}

    def trmv(self, P, name, x, alpha=1.0, beta=0.0, z=None, out=None):
        """out = alpha * P.<name> x + beta * z   (name in LTi / LTit / LAi / LAit)."""
        if out is None:
            out = torch.empty(P.M, dtype=torch.float64, device=self.device)
        hip.check(self.lib.odx_trmv_f64(_p(getattr(P, name)), P.ld, P.M, self._TRI[name], _p(x), float(alpha),
                                        float(beta), _p(z), _p(out), self._stream()), "odx_trmv_f64")
        return out

    def cg_init(self, B, X, R, Pv, state):
        hip.check(self.lib.odx_cg_init(_p(B), _p(X), _p(R), _p(Pv), _p(state), B.numel(), self._stream()), "odx_cg_init")

    def cg_step(self, X, R, Pv, AP, state, cg_eps, full_grad):
        hip.check(self.lib.odx_cg_step(_p(X), _p(R), _p(Pv), _p(AP), _p(state), float(cg_eps), int(bool(full_grad)),
                                       X.numel(), self._stream()), "odx_cg_step")

    def cg_finish(self, R, Pv, state, cg_eps, tol):
        hip.check(self.lib.odx_cg_finish(_p(R), _p(Pv), _p(state), float(cg_eps), float(tol), R.numel(),
                                         self._stream()), "odx_cg_finish")

    def axpby(self, a, x, b, y):
        hip.check(self.lib.odx_axpby_f64(float(a), _p(x), float(b), _p(y), y.numel(), self._stream()), "odx_axpby_f64")

    # ------------------------------------------------------------------ scoring
    def mmv(self, F, Zf, sigma, V, ranges=None, out=None, max_range=None):
        """(n, T) f32 = K(F, Zf) @ V with V (Mtot, T) f64; ``ranges`` (T, 2) int32 row ranges of the
        non-zero block of each column (None = dense); ``max_range``: an upper bound of the range lengths when the
        caller knows one (sizes the launch; default Mtot)."""
        V = V.to(device=self.device, dtype=torch.float64)
        if V.dim() == 1:
            V = V[:, None]
        V = V.contiguous()
        Mtot, T = V.shape
        if Mtot != Zf.n:
            raise ValueError("mmv: V has %d rows but there are %d centres" % (Mtot, Zf.n))
        if ranges is None:
            key = ("dense_ranges", Mtot, T)          # a host -> device copy per call otherwise (a predict is ~50 us of GPU work)
            ranges = self._ws.get(key)
            if ranges is None:
                ranges = self._ws[key] = torch.tensor([[0, Mtot]] * T, dtype=torch.int32, device=self.device)
        ranges = ranges.to(device=self.device, dtype=torch.int32).contiguous()
        if out is None:
            out = torch.empty((F.n, T), dtype=torch.float32, device=self.device)
        if F.n == 0 or T == 0:
            return out
        if Mtot == 0:
            return out.zero_()
        if self.gauss == "f8":
            self.pack8(F), self.pack8(Zf)
            mr = Mtot if max_range is None else max(1, min(int(max_range), Mtot))
            ws = self._workspace("mmv", self.lib.odx_gauss_mmv_h2_workspace_bytes(F.n, mr, T))
            hip.check(self.lib.odx_gauss_mmv_f8(_p(F.P8), F.P8.stride(0), _p(F.meta8), _p(F.sq8), F.n, _p(Zf.P8), Zf.P8.stride(0),
                                                _p(Zf.meta8), _p(Zf.sq8), mr, F.D, float(sigma), _p(V), V.stride(0), _p(ranges), T,
                                                _p(out), out.stride(0), _p(ws), ws.numel(), self._stream()), "odx_gauss_mmv_f8")
        elif self.gauss == "h2":
            self.pack(F), self.pack(Zf)
            mr = Mtot if max_range is None else max(1, min(int(max_range), Mtot))
            ws = self._workspace("mmv", self.lib.odx_gauss_mmv_h2_workspace_bytes(F.n, mr, T))
            hip.check(self.lib.odx_gauss_mmv_h2(_p(F.P), F.P.stride(0), _p(F.meta), _p(F.sq), F.n, _p(Zf.P), Zf.P.stride(0),
                                                _p(Zf.meta), _p(Zf.sq), mr, F.D, float(sigma), _p(V), V.stride(0), _p(ranges), T,
                                                _p(out), out.stride(0), _p(ws), ws.numel(), self._stream()), "odx_gauss_mmv_h2")
        else:
            hip.check(self.lib.odx_gauss_mmv_f32(_p(F.X), F.ld, _p(F.sq), F.n, _p(Zf.X), Zf.ld, _p(Zf.sq), F.D,
                                                 float(sigma), _p(V), V.stride(0), _p(ranges), T, _p(out), out.stride(0),
                                                 self._stream()), "odx_gauss_mmv_f32")
        return out

    # ------------------------------------------------------------------ RLS (A7)
    def rls_gram(self, F, idx, Yt, G, XtY):
        """G (D1 x D1 lower) += [X 1]'[X 1],  XtY (4 x D1) += Yt [X 1]  over rows ``idx`` of F."""
        nc = idx.numel()
        if nc == 0:
            return
        nbytes = self.lib.odx_rls_gram_workspace_bytes(nc, F.D)
        ws = self._workspace("rls_gram", nbytes)
        hip.check(self.lib.odx_rls_gram_f64(_p(F.X), F.ld, F.D, _p(idx), nc, _p(Yt), Yt.stride(0), _p(G), G.stride(0),
                                            _p(XtY), XtY.stride(0), _p(ws), ws.numel(), self._stream()),
                  "odx_rls_gram_f64")

    def rls_solve(self, G, D, lam, XtY):
        D1 = D + 1
        W = torch.empty((4, D1), dtype=torch.float64, device=self.device)
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        nbytes = self.lib.odx_rls_solve_workspace_bytes(D)
        ws = self._workspace("rls_solve", nbytes)
        hip.check(self.lib.odx_rls_solve_f64(_p(G), G.stride(0), D, float(lam), _p(XtY), XtY.stride(0), _p(W),
                                             W.stride(0), _p(info), _p(ws), ws.numel(), self._stream()),
                  "odx_rls_solve_f64")
        return W, info

    def rls_rows_form(self, F):
        """Whether the Grams of F's rows can be formed on their own (rls_gram_begin): see odx_rls_rows_form."""
        return bool(F.n) and bool(self.lib.odx_rls_rows_form(_p(F.X), F.ld, F.D))

    def rls_gram_zeros(self, F, C):
        D1 = F.D + 1
        return torch.zeros((C, D1, (D1 + 1) // 2 * 2), dtype=torch.float64, device=self.device)

    def rls_gram_begin(self, F, idx_pad, seg_off, seg_len, G):
        """Queue the Grams of the classes alone (they need the rows, not the targets) into G = rls_gram_zeros(F, C), which
        rls_train_batched(begun=G) completes.  Needs rls_rows_form(F)."""
        C, D = len(seg_len), F.D
        npad = int(idx_pad.numel())
        if not npad:
            return G
        D1 = D + 1
        ld = (D1 + 1) // 2 * 2
        so = (ctypes.c_int64 * C)(*[int(v) for v in seg_off])
        sl = (ctypes.c_int64 * C)(*[int(v) for v in seg_len])
        ws = self._workspace("rls_gram_batched", self.lib.odx_rls_gram_batched_workspace_bytes(npad, D))
        hip.check(self.lib.odx_rls_gram_batched_f64(_p(F.X), F.ld, D, _p(idx_pad), npad, so, sl, C, None, 0, _p(G), ld, D1 * ld, None, 0, 0,
                                                    _p(ws), ws.numel(), self._stream()), "odx_rls_gram_batched_f64")
        return G

    def rls_pad_index(self, run, seg_off, seg_len, npad):
        """(idx_pad (npad), gid, pos, dest (total each), lens (C)) for the row ids `run` of a class batch, one launch
        (odx_rls_pad_index)."""
        C, total = len(seg_len), int(run.numel())
        run = run.contiguous()
        idx_pad = torch.empty(npad, dtype=torch.int64, device=self.device)
        maps = torch.empty(3 * max(total, 1) + C, dtype=torch.int64, device=self.device)
        m = maps[:3 * max(total, 1)].view(3, -1)
        lens = maps[3 * max(total, 1):]
        so = (ctypes.c_int64 * C)(*[int(v) for v in seg_off])
        sl = (ctypes.c_int64 * C)(*[int(v) for v in seg_len])
        hip.check(self.lib.odx_rls_pad_index(_p(run), total, so, sl, C, int(npad), _p(idx_pad), _p(m[0]), _p(m[1]), _p(m[2]), _p(lens),
                                             self._stream()), "odx_rls_pad_index")
        return idx_pad, m[0, :total], m[1, :total], m[2, :total], lens

    def rls_gram_raw_begin(self, F, idx_pad, seg_off, seg_len, G, Yraw):
        """rls_gram_begin that also forms the RAW targets' products in the same sweep (odx_rls_gram_raw_batched_f64): Yraw (n, 4)
        f32 by row id -> O5 (C, 5, ld) f64 = [Y 1]' X, which rls_train_batched(raw=(O5, stats, cnt)) turns into the whitened
        targets' X' Yw once the statistics exist — no second sweep over the rows behind the Grams."""
        C, D = len(seg_len), F.D
        D1 = D + 1
        ld = (D1 + 1) // 2 * 2
        O5 = torch.empty((C, 5, ld), dtype=torch.float64, device=self.device)
        so = (ctypes.c_int64 * C)(*[int(v) for v in seg_off])
        sl = (ctypes.c_int64 * C)(*[int(v) for v in seg_len])
        hip.check(self.lib.odx_rls_gram_raw_batched_f64(_p(F.X), F.ld, D, _p(idx_pad), int(idx_pad.numel()), so, sl, C, _p(Yraw), Yraw.stride(0),
                                                        _p(G), ld, D1 * ld, _p(O5), ld, self._stream()), "odx_rls_gram_raw_batched_f64")
        return O5

    def rls_train_batched(self, F, idx_pad, seg_off, seg_len, Yt, lam, allreduce=None, begun=None, after=None, raw=None):
        """The RLS solves of len(seg_len) <= 32 classes with one launch chain (odx_rls_gram_batched_f64 +
        odx_rls_solve_batched_f64).  idx_pad: device int64 row ids class after class, every segment starting at a multiple
        of 16 and padded with -1; seg_off / seg_len: python lists; Yt (4, ldy) f64 whitened targets in the same padded
        order.  allreduce: optional callable summing a tensor over row shards (applied to the Grams and X'Y).  begun: the G
        block of rls_gram_begin for the same rows, queued on stream `after`: only Yt [X 1] is formed here (beside the Grams —
        it writes other entries of G than they do), and the solves wait for `after`.  Returns W (C, 4, ldw) f64 and info
        (C,) int32."""
        C, D = len(seg_len), F.D
        D1 = D + 1
        ld = (D1 + 1) // 2 * 2
        npad = int(idx_pad.numel())
        G = begun if begun is not None else torch.zeros((C, D1, ld), dtype=torch.float64, device=self.device)
        XtY = torch.zeros((C, 4, ld), dtype=torch.float64, device=self.device)
        W = torch.empty((C, 4, ld), dtype=torch.float64, device=self.device)
        info = torch.zeros(C, dtype=torch.int32, device=self.device)
        so = (ctypes.c_int64 * C)(*[int(v) for v in seg_off])
        sl = (ctypes.c_int64 * C)(*[int(v) for v in seg_len])
        if raw is not None:
            # (the Grams and [Y 1]' X are on stream `after`: the fold needs both, and writes the Gram's bias row)
            O5, stats, cnt = raw
            torch.cuda.current_stream(self.device).wait_stream(after)
            hip.check(self.lib.odx_rls_fold_whitened_f64(_p(O5), O5.stride(1), D, C, _p(stats), _p(cnt), _p(G), ld, D1 * ld, _p(XtY), ld, 4 * ld,
                                                         self._stream()), "odx_rls_fold_whitened_f64")
            after = None
        elif npad:
            ws = self._workspace("rls_gram_batched", self.lib.odx_rls_gram_batched_workspace_bytes(npad, D))
            fn, name = ((self.lib.odx_rls_xty_batched_f64, "odx_rls_xty_batched_f64") if begun is not None
                        else (self.lib.odx_rls_gram_batched_f64, "odx_rls_gram_batched_f64"))
            hip.check(fn(_p(F.X), F.ld, D, _p(idx_pad), npad, so, sl, C, _p(Yt), Yt.stride(0), _p(G), ld, D1 * ld, _p(XtY), ld, 4 * ld,
                         _p(ws), ws.numel(), self._stream()), name)
        if after is not None:
            torch.cuda.current_stream(self.device).wait_stream(after)
        if allreduce is not None:
            allreduce(G)
            allreduce(XtY)
        ws = self._workspace("rls_solve_batched", self.lib.odx_rls_solve_batched_workspace_bytes(D, C))
        hip.check(self.lib.odx_rls_solve_batched_f64(_p(G), ld, D1 * ld, D, C, float(lam), _p(XtY), ld, 4 * ld, _p(W), ld, 4 * ld,
                                                     _p(info), _p(ws), ws.numel(), self._stream()), "odx_rls_solve_batched_f64")
        return W, info

    def rls_predict_rows(self, F, idx, W, out=None):
        nc = F.n if idx is None else idx.numel()
        Pm = torch.empty((nc, 4), dtype=torch.float64, device=self.device) if out is None else out
        if tuple(Pm.shape) != (nc, 4) or Pm.dtype != torch.float64 or not Pm.is_contiguous():
            raise ValueError("rls_predict_rows: out must be a contiguous (%d, 4) f64 tensor" % nc)
        if nc:
            hip.check(self.lib.odx_rls_predict_rows_f64(_p(F.X), F.ld, F.D, _p(idx), nc, _p(W), W.stride(0), _p(Pm), 4,
                                                        self._stream()), "odx_rls_predict_rows_f64")
        return Pm


    def rls_predict_rows_batched(self, F, idx_all, starts, W, out):
        """out (total, 4) f64 = [X 1] w for the rows of len(starts) <= 32 classes with one launch: idx_all holds the row ids
        class after class, class c's first at starts[c]; W (C, 4, ldw)."""
        total = int(idx_all.numel())
        C = len(starts)
        if total == 0:
            return out
        st = (ctypes.c_int64 * C)(*[int(v) for v in starts])
        hip.check(self.lib.odx_rls_predict_rows_batched_f64(_p(F.X), F.ld, F.D, _p(idx_all), st, C, total, _p(W), W.stride(1),
                                                            W.stride(0), _p(out), 4, self._stream()), "odx_rls_predict_rows_batched_f64")
        return out

    # ------------------------------------------------------------------ dense f32 product (RLS apply)
    def gemm_nt(self, Fa, Fb):
        """(n, m) f32 = Fa.X (n, D) @ Fb.X (m, D)' on the f32 MFMA core (exact f32 fmaf chain)."""
        out = torch.empty((Fa.n, Fb.n), dtype=torch.float32, device=self.device)
        if Fa.n and Fb.n:
            hip.check(self.lib.odx_gemm_nt_f32(_p(Fa.X), Fa.ld, _p(Fb.X), Fb.ld, _p(out), Fb.n, Fa.n, Fb.n, Fa.D,
                                               1.0, 0.0, 0, self._stream()), "odx_gemm_nt_f32")
        return out

    def gemm_nt_f64(self, A, B):
        """(m, n) f64 = A (m, k) @ B (n, k)' on the f64 matrix cores (odx_gemm_nt_f64); the result's rows are an even number
        of doubles apart (what the RLS solve asks of its operands): returns the (m, n) view of an (m, ldn) block."""
        A, B = (t.to(device=self.device, dtype=torch.float64) for t in (A, B))
        m, k = A.shape
        n = B.shape[0]
        if B.shape[1] != k:
            raise ValueError("gemm_nt_f64: inner dimensions differ (%d, %d)" % (k, B.shape[1]))

        def even(t):
            if t.stride(1) == 1 and t.stride(0) % 2 == 0 and t.data_ptr() % 16 == 0:
                return t
            buf = torch.zeros((t.shape[0], (t.shape[1] + 1) // 2 * 2), dtype=torch.float64, device=self.device)
            buf[:, :t.shape[1]] = t
            return buf[:, :t.shape[1]]
        A, B = even(A), even(B)
        out = torch.zeros((m, (n + 1) // 2 * 2), dtype=torch.float64, device=self.device)
        if m and n and k:
            hip.check(self.lib.odx_gemm_nt_f64(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), m, n, k, 1.0, 0.0, 0,
                                               self._stream()), "odx_gemm_nt_f64")
        return out[:, :n]

    # ------------------------------------------------------------------ feature-forward ops (A11)
    def roi_align(self, feat, rois, spatial_scale, output_size, sampling_ratio=0):
        """maskrcnn_benchmark.layers.ROIAlign forward: feat (N, C, H, W) f32, rois (R, 5)."""
        feat = feat.to(device=self.device, dtype=torch.float32).contiguous()
        rois = rois.to(device=self.device, dtype=torch.float32).contiguous()
        N, C, H, W = feat.shape
        PH, PW = output_size
        R = rois.shape[0]
        out = torch.empty((R, C, PH, PW), dtype=torch.float32, device=self.device)
        hip.check(self.lib.odx_roi_align_fwd_f32(_p(feat), N, C, H, W, _p(rois), R, float(spatial_scale), PH, PW,
                                                 int(sampling_ratio), _p(out), self._stream()), "odx_roi_align_fwd_f32")
        return out

    def roi_align_fpn(self, feats, rois, scales, output_size, sampling_ratio=2, return_levels=False):
        """maskrcnn_benchmark's Pooler over an FPN pyramid with one launch: feats = list of (N, C, H_l, W_l) f32 maps,
        scales = their 1 / strides; every RoI (R, 5) is pooled from the level LevelMapper assigns it.  -> (R, C, PH, PW)."""
        feats = [f.to(device=self.device, dtype=torch.float32).contiguous() for f in feats]
        rois = rois.to(device=self.device, dtype=torch.float32).contiguous()
        L = len(feats)
        N, C = feats[0].shape[:2]
        if any(f.shape[0] != N or f.shape[1] != C for f in feats):
            raise ValueError("roi_align_fpn: every level needs the same batch size and channel count")
        PH, PW = output_size
        R = rois.shape[0]
        out = torch.empty((R, C, PH, PW), dtype=torch.float32, device=self.device)
        lv = torch.empty(R, dtype=torch.int32, device=self.device) if return_levels else None
        if R:
            fp = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
            hs = (ctypes.c_int * L)(*[int(f.shape[2]) for f in feats])
            ws = (ctypes.c_int * L)(*[int(f.shape[3]) for f in feats])
            sc = (ctypes.c_float * L)(*[float(v) for v in scales])
            hip.check(self.lib.odx_roi_align_fpn_f32(fp, hs, ws, sc, L, N, C, _p(rois), R, PH, PW, int(sampling_ratio), _p(out),
                                                     _p(lv), self._stream()), "odx_roi_align_fpn_f32")
        return (out, lv) if return_levels else out

    def roi_align_fpn_rows(self, feats, rois, scales, output_size, sampling_ratio=2):
        """roi_align_fpn for a pyramid handed over as channels-last views of NHWC row matrices (what the pyramid's row GEMMs
        write): -> (R, PH PW C) f32, the crops flattened in (ph, pw, c) order (odx_roi_align_fpn_nhwc_f32: no NCHW copy of the
        levels).  Levels that are not such views are taken through a channels-last copy."""
        # (16-bit levels — a pyramid run natively in bf16 / f16 — are read as they are: odx_roi_align_fpn_nhwc_16 decodes the samples)
        dt = feats[0].dtype if feats[0].dtype in (torch.bfloat16, torch.float16) and all(f.dtype == feats[0].dtype for f in feats) else torch.float32
        fs = []
        for f in feats:
            f = f.to(device=self.device, dtype=dt)
            if f.is_contiguous() or not f.is_contiguous(memory_format=torch.channels_last) or f.data_ptr() % 16 != 0:
                f = f.contiguous(memory_format=torch.channels_last)
            fs.append(f)
        rois = rois.to(device=self.device, dtype=torch.float32).contiguous()
        L = len(fs)
        N, C = fs[0].shape[:2]
        if any(f.shape[0] != N or f.shape[1] != C for f in fs) or C % 4 != 0:
            raise ValueError("roi_align_fpn_rows: every level needs the same batch size and a channel count in fours")
        PH, PW = output_size
        R = rois.shape[0]
        out = torch.empty((R, PH * PW * C), dtype=torch.float32, device=self.device)
        if R:
            fp = (ctypes.c_void_p * L)(*[f.data_ptr() for f in fs])
            hs = (ctypes.c_int * L)(*[int(f.shape[2]) for f in fs])
            ws = (ctypes.c_int * L)(*[int(f.shape[3]) for f in fs])
            sc = (ctypes.c_float * L)(*[float(v) for v in scales])
            if dt == torch.float32:
                hip.check(self.lib.odx_roi_align_fpn_nhwc_f32(fp, hs, ws, sc, L, N, C, _p(rois), R, PH, PW, int(sampling_ratio), _p(out), None,
                                                              self._stream()), "odx_roi_align_fpn_nhwc_f32")
            else:
                hip.check(self.lib.odx_roi_align_fpn_nhwc_16(fp, int(dt == torch.bfloat16), hs, ws, sc, L, N, C, _p(rois), R, PH, PW,
                                                             int(sampling_ratio), _p(out), None, self._stream()), "odx_roi_align_fpn_nhwc_16")
        return out

    def packed(self, X, meta=None, zero_row=False):
        """X (rows, K) f32 as a GEMM operand of gemm_h2: its packed two-term f16 split (no row norms).  meta: the two meta
        words a producer already left for X (gemm_h2(..., with_max=True): max |X| in meta[1]) — no maximum pass then; a bound
        on max |X| (the maximum of a matrix X's rows were cut from) serves as well."""
        X = X.to(device=self.device, dtype=torch.float32)
        if X.stride(1) != 1 or X.stride(0) % 4 != 0 or X.data_ptr() % 16 != 0:
            ld = (X.shape[1] + 3) // 4 * 4
            buf = torch.zeros((X.shape[0], ld), dtype=torch.float32, device=self.device)
            buf[:, :X.shape[1]] = X
            X = buf[:, :X.shape[1]]
        return self.pack(Features(X, None, X.shape[1], meta=meta), zero_row=zero_row)

    @staticmethod
    def weight_bounds(w, bias=None):
        """(bound_w, bound_add) of odx_gemm_h2_chain_f32 for the weight matrix w (n, K) and bias: sqrt(K) max_j |w_j|_2 and
        max |bias|, rounded up (host numbers, computed once per layer when its weights are packed)."""
        w = w.detach().double()
        bw = float(w.norm(dim=1).max()) * (w.shape[1] ** 0.5) * (1 + 1e-6) if w.numel() else 0.0
        ba = float(bias.detach().abs().max()) * (1 + 1e-6) if bias is not None and bias.numel() else 0.0
        return bw, ba

    def _chain_out(self, m, n, f32_out, zero_row):
        ldop = (n + 63) // 64 * 64
        buf = torch.empty((m + 1 if zero_row else m, ldop), dtype=torch.int32, device=self.device)
        if zero_row:
            buf[m].zero_()
        out = torch.empty((m, n), dtype=torch.float32, device=self.device) if f32_out else None
        return out, buf[:m], self._meta_slot()

    def chain_gemm(self, A, B, bias=None, residual=None, residual_meta=None, relu=False, bounds=(0.0, 0.0), f32_out=True, zero_row=False):
        """A layer of a chain: act(A B' + bias + residual) written AS the next layer's packed operand by the product's own
        epilogue (odx_gemm_h2_chain_f32) — PackedRows with P / meta, and X when f32_out.  A: Features / PackedRows; bounds =
        weight_bounds(B's matrix, bias); residual_meta: the meta words of the residual (its maximum in [1])."""
        m, n, K = A.n, B.n, A.D
        if B.D != K:
            raise ValueError("chain_gemm: inner dimensions differ (%d, %d)" % (K, B.D))
        if residual is not None and residual_meta is None:
            raise ValueError("chain_gemm: a residual needs its meta words")
        out, P, meta = self._chain_out(m, n, f32_out, zero_row)
        if m and n:
            if bias is not None:
                bias = bias.to(device=self.device, dtype=torch.float32).contiguous()
            ldr = 0
            if residual is not None:
                residual = residual.to(device=self.device, dtype=torch.float32)
                if residual.stride(1) != 1:
                    residual = residual.contiguous()
                ldr = residual.stride(0)
            hip.check(self.lib.odx_gemm_h2_chain_f32(_p(A.P), A.P.stride(0), _p(A.meta), m, _p(B.P), B.P.stride(0), _p(B.meta), n, K,
                                                     _p(bias), _p(residual), ldr, int(bool(relu)), _p(out), n, _p(meta), _p(P), P.stride(0),
                                                     float(bounds[0]), float(bounds[1]), _p(residual_meta), self._stream()), "odx_gemm_h2_chain_f32")
        return PackedRows(out, m, n, P, meta, zero_row)

    def chain_conv3x3(self, A, R, H, W, B, bias=None, relu=False, bounds=(0.0, 0.0), f32_out=False, zero_row=False):
        """chain_gemm for a 3 x 3 convolution (padding 1) over the packed NHWC rows A (R H W, C): the taps gathered inside the
        product's operand loads where the library does that (A.zero_row required), else gathered from the packed rows into the
        packed neighbourhood matrix (odx_taps3x3_packed) and multiplied."""
        C, n, m = A.D, B.n, R * H * W
        if A.n != m or B.D != 9 * C:
            raise ValueError("chain_conv3x3: %d rows of %d channels against R H W = %d, K = %d" % (A.n, C, m, B.D))
        if A.zero_row and self.lib.odx_gemm_h2_taps_supported(m, n, C, A.P.stride(0)):
            out, P, meta = self._chain_out(m, n, f32_out, zero_row)
            if bias is not None:
                bias = bias.to(device=self.device, dtype=torch.float32).contiguous()
            hip.check(self.lib.odx_gemm_h2_taps_f32(_p(A.P), A.P.stride(0), _p(A.meta), R, H, W, C, _p(B.P), B.P.stride(0), _p(B.meta), n,
                                                    _p(bias), None, 0, int(bool(relu)), _p(out), n, _p(meta), _p(P), P.stride(0),
                                                    float(bounds[0]), float(bounds[1]), None, self._stream()), "odx_gemm_h2_taps_f32")
            return PackedRows(out, m, n, P, meta, zero_row)
        D = 9 * C
        T = PackedRows(None, m, D, torch.empty((m, (D + 63) // 64 * 64), dtype=torch.int32, device=self.device), A.meta)
        if m:
            hip.check(self.lib.odx_taps3x3_packed(_p(A.P), A.P.stride(0), R, H, W, C, _p(T.P), T.P.stride(0), self._stream()), "odx_taps3x3_packed")
        return self.chain_gemm(T, B, bias=bias, relu=relu, bounds=bounds, f32_out=f32_out, zero_row=zero_row)

    def stem_pool_rows(self, y, bias):
        """The stem's tail in one pass (odx_stem_pool_rows_f32 / _16): y (B, C, H, W) the 7 x 7 convolution's output WITHOUT its
        bias, contiguous, f32 / bf16 / f16 -> relu(y + bias), max-pooled 3 x 3 / 2 / 1, as NHWC rows.  f32: (rows (B Ho Wo, C),
        (B, Ho, Wo), meta words with max |rows|); 16 bits: (Rows16 with a zero row, (B, Ho, Wo), None)."""
        B, C, H, W = y.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        if not y.is_contiguous():
            raise ValueError("stem_pool_rows: a contiguous (B, C, H, W) map expected")
        bias = bias.to(device=self.device, dtype=y.dtype).contiguous()
        if y.dtype == torch.float32:
            rows = torch.empty((B * Ho * Wo, C), dtype=torch.float32, device=self.device)
            om = self._meta_slot()
            hip.check(self.lib.odx_stem_pool_rows_f32(_p(y), _p(bias), B, C, H, W, _p(rows), C, _p(om), self._stream()), "odx_stem_pool_rows_f32")
            return rows, (B, Ho, Wo), om
        ld = (C + 127) // 128 * 128
        buf = torch.zeros((B * Ho * Wo + 1, ld), dtype=y.dtype, device=self.device)         # (pad columns and the zero row)
        hip.check(self.lib.odx_stem_pool_rows_16(_p(y), _p(bias), int(y.dtype == torch.bfloat16), B, C, H, W, _p(buf), ld, self._stream()),
                  "odx_stem_pool_rows_16")
        return Rows16(buf[:B * Ho * Wo], C, True), (B, Ho, Wo), None

    def upsample_add_rows(self, lat, top, B, H, W, Hp, Wp):
        """The top-down step of a feature pyramid on NHWC rows, in place: lat (B H W, C) f32 += top (B Hp Wp, C) at the nearest-
        neighbour source row (odx_upsample_add_rows_f32).  Returns the meta words holding max |sum| — what `packed` / `conv3x3_rows`
        take as `meta`."""
        C = lat.shape[1]
        if (lat.dtype != torch.float32 or top.dtype != torch.float32 or lat.shape[0] != B * H * W or top.shape != (B * Hp * Wp, C)
                or lat.stride(1) != 1 or top.stride(1) != 1):
            raise ValueError("upsample_add_rows: (B H W, C) and (B Hp Wp, C) f32 row matrices expected")
        om = self._meta_slot()
        hip.check(self.lib.odx_upsample_add_rows_f32(_p(lat), lat.stride(0), _p(top), top.stride(0), B, H, W, Hp, Wp, C, _p(om), self._stream()),
                  "odx_upsample_add_rows_f32")
        return om

    def upsample_add_rows16(self, lat, top, B, H, W, Hp, Wp):
        """upsample_add_rows for 16-bit rows (Rows16, in place in lat's buffer): added in f32, rounded once."""
        C, dt = lat.K, lat.buf.dtype
        if top.buf.dtype != dt or top.K != C or lat.n != B * H * W or top.n != B * Hp * Wp:
            raise ValueError("upsample_add_rows16: Rows16 of one type, (B H W) and (B Hp Wp) rows of the same channels expected")
        hip.check(self.lib.odx_upsample_add_rows_16(_p(lat.buf), lat.buf.stride(0), _p(top.buf), top.buf.stride(0), int(dt == torch.bfloat16),
                                                    B, H, W, Hp, Wp, C, self._stream()), "odx_upsample_add_rows_16")
        return lat

    def conv3x3_rows(self, Y, R, H, W, B, bias=None, relu=False, meta=None, with_max=False):
        """act(3 x 3 convolution, padding 1) of the NHWC rows Y (R H W, C) f32 with the packed weights B (n, 9 C: ky kx c) as
        ONE product: where the library gathers the taps inside the product's operand loads (odx_gemm_h2_taps_f32: wide
        layers on the 256 x 256 core) Y is packed once and its 9 x neighbourhood matrix never written; elsewhere that matrix
        is written in packed form (packed_taps3x3) and multiplied (gemm_h2).  Same sums either way.  meta / with_max as for
        `packed` / `gemm_h2`."""
        C, n, m = Y.shape[1], B.n, R * H * W
        ok = (Y.dtype == torch.float32 and Y.stride(1) == 1 and Y.stride(0) % 4 == 0 and Y.data_ptr() % 16 == 0 and B.D == 9 * C
              and self.lib.odx_gemm_h2_taps_supported(m, n, C, (C + 63) // 64 * 64))
        if not ok:
            return self.gemm_h2(self.packed_taps3x3(Y, R, H, W, meta=meta), B, bias=bias, relu=relu, with_max=with_max)
        Fy = self.packed(Y, meta=meta, zero_row=True)
        out = torch.empty((m, n), dtype=torch.float32, device=self.device)
        if bias is not None:
            bias = bias.to(device=self.device, dtype=torch.float32).contiguous()
        om = self._meta_slot() if with_max else None
        hip.check(self.lib.odx_gemm_h2_taps_f32(_p(Fy.P), Fy.P.stride(0), _p(Fy.meta), R, H, W, C, _p(B.P), B.P.stride(0), _p(B.meta), n,
                                                _p(bias), None, 0, int(bool(relu)), _p(out), n, _p(om), None, 0, 0.0, 0.0, None,
                                                self._stream()), "odx_gemm_h2_taps_f32")
        return (out, om) if with_max else out

    def packed_taps3x3(self, Y, R, H, W, meta=None):
        """The packed operand of a 3 x 3 convolution (padding 1) as a GEMM over the NHWC rows Y (R * H * W, C): K = 9 C,
        tap after tap; written in the packed form directly (odx_split_f16_taps3x3; with the meta words of Y's producer,
        as for `packed`: odx_split_f16_taps3x3_premax)."""
        C = Y.shape[1]
        if C % 8 != 0 or Y.stride(1) != 1 or Y.stride(0) % 4 != 0 or Y.data_ptr() % 16 != 0 or Y.dtype != torch.float32:
            yp = torch.nn.functional.pad(Y.reshape(R, H, W, C), (0, 0, 1, 1, 1, 1))
            return self.packed(torch.cat([yp[:, ky:ky + H, kx:kx + W, :] for ky in range(3) for kx in range(3)], dim=3).reshape(R * H * W, 9 * C))
        n, D = R * H * W, 9 * C
        Fp = Features(Y, None, C)
        Fp.n, Fp.D = n, D                              # the operand the GEMM sees; X stays the (n, C) source
        Fp.P = torch.empty((n, (D + 63) // 64 * 64), dtype=torch.int32, device=self.device)
        if meta is not None:
            Fp.meta = meta
            hip.check(self.lib.odx_split_f16_taps3x3_premax(_p(Y), Y.stride(0), R, H, W, C, _p(Fp.P), Fp.P.stride(0), _p(meta), self._stream()),
                      "odx_split_f16_taps3x3_premax")
            return Fp
        Fp.meta = torch.empty(2, dtype=torch.float32, device=self.device)
        hip.check(self.lib.odx_split_f16_taps3x3(_p(Y), Y.stride(0), R, H, W, C, _p(Fp.P), Fp.P.stride(0), _p(Fp.meta), self._stream()),
                  "odx_split_f16_taps3x3")
        return Fp

    def gemm_h2(self, A, B, bias=None, residual=None, relu=False, with_max=False):
        """(m, n) f32 = act(A.X B.X' + bias + residual) for packed operands (`packed`): the f32 product at f32 accuracy on
        the f16 matrix cores (odx_gemm_h2_f32).  with_max: returns (out, meta) — the launch leaves max |out| in meta[1]
        (odx_gemm_h2_max_f32), the words `packed` / `packed_taps3x3` take for the next layer."""
        m, n, K = A.n, B.n, A.D
        if B.D != K:
            raise ValueError("gemm_h2: inner dimensions differ (%d, %d)" % (K, B.D))
        out = torch.empty((m, n), dtype=torch.float32, device=self.device)
        if m == 0 or n == 0:
            return (out, self._meta_slot()) if with_max else out
        if bias is not None:
            bias = bias.to(device=self.device, dtype=torch.float32).contiguous()
        ldr = 0
        if residual is not None:
            residual = residual.to(device=self.device, dtype=torch.float32)
            if residual.stride(1) != 1:
                residual = residual.contiguous()
            ldr = residual.stride(0)
        if with_max:
            meta = self._meta_slot()
            hip.check(self.lib.odx_gemm_h2_max_f32(_p(A.P), A.P.stride(0), _p(A.meta), m, _p(B.P), B.P.stride(0), _p(B.meta), n, K,
                                                   _p(bias), _p(residual), ldr, int(bool(relu)), _p(out), n, _p(meta), self._stream()),
                      "odx_gemm_h2_max_f32")
            return out, meta
        hip.check(self.lib.odx_gemm_h2_f32(_p(A.P), A.P.stride(0), _p(A.meta), m, _p(B.P), B.P.stride(0), _p(B.meta), n, K,
                                           _p(bias), _p(residual), ldr, int(bool(relu)), _p(out), n, self._stream()), "odx_gemm_h2_f32")
        return out

    # ------------------------------------------------------------------ 16-bit layers (a forward run in bf16 / f16)
    def rows16(self, X, dtype=None, zero_row=False):
        """X (rows, K) as the operand of gemm_b16: a bf16 / f16 row-major block whose rows are a multiple of 128 elements
        long, zero beyond K (Rows16).  A matrix that already is one (right dtype, contiguous, K % 128 == 0) is taken as it is.
        zero_row: the block is followed by one all-zero row in memory (conv3x3_rows16 gathers its taps in the product then)."""
        if isinstance(X, Rows16) and (X.zero_row or not zero_row) and (dtype is None or X.buf.dtype == dtype):
            return X
        if isinstance(X, Rows16):            # (no zero row where one is wanted, or another 16-bit type than asked for: made again)
            X = X.dense
        dtype = dtype or (X.dtype if X.dtype in (torch.bfloat16, torch.float16) else torch.bfloat16)
        n, K = X.shape
        ld = (K + 127) // 128 * 128
        if not zero_row and X.is_cuda and X.dtype == dtype and K == ld and X.stride(1) == 1 and X.stride(0) == ld and X.data_ptr() % 16 == 0:
            return Rows16(X, K)
        buf, _ = self._rows16_out(n, K, dtype, zero_row)
        buf[:, :K].copy_(X)
        return Rows16(buf, K, zero_row)

    def gemm_b16(self, A, B, bias=None, residual=None, relu=False, out_f32=False, zero_row=False):
        """act(A B' + bias (+ residual)) for Rows16 operands A (m, K), B (n, K) of one 16-bit type (odx_gemm_b16: one MFMA term
        per product, f32 sums, one rounding).  Returns a Rows16 of the operands' type — ready to be the next layer's operand,
        its pad columns zero — or, with out_f32, an (m, n) f32 tensor.  residual: Rows16 of the operands' type (f32 tensor
        with out_f32) of the result's shape."""
        m, n, K = A.n, B.n, A.K
        dt = A.buf.dtype
        if B.K != K or B.buf.dtype != dt:
            raise ValueError("gemm_b16: operands differ in inner dimension (%d, %d) or type" % (K, B.K))
        if out_f32:
            out = torch.empty((m, n), dtype=torch.float32, device=self.device)
            ldo, res, ldr = n, None, 0
            if residual is not None:
                res = residual.to(device=self.device, dtype=torch.float32)
                res = res if res.stride(1) == 1 else res.contiguous()
                ldr = res.stride(0)
        else:
            out, ldo = self._rows16_out(m, n, dt, zero_row)
            res, ldr = None, 0
            if residual is not None:
                if not isinstance(residual, Rows16) or residual.buf.dtype != dt or residual.n != m or residual.K != n:
                    raise ValueError("gemm_b16: the residual must be a Rows16 of the result's shape and type")
                res, ldr = residual.buf, residual.buf.stride(0)
        if bias is not None:
            bias = bias.to(device=self.device, dtype=torch.float32).contiguous()
        if m and n:
            hip.check(self.lib.odx_gemm_b16(_p(A.buf), A.buf.stride(0), m, _p(B.buf), B.buf.stride(0), n, K, 1 if dt == torch.bfloat16 else 0,
                                            _p(bias), _p(res), ldr, int(bool(relu)), _p(out), ldo, 0 if out_f32 else 1, self._stream()),
                      "odx_gemm_b16")
        return out if out_f32 else Rows16(out, n, zero_row)

    def _rows16_out(self, m, n, dt, zero_row):
        """(m, roundup(n, 128)) 16-bit output block with zero pad columns, followed by an all-zero row when asked."""
        ldo = (n + 127) // 128 * 128
        buf = torch.empty((m + 1 if zero_row else m, ldo), dtype=dt, device=self.device)
        if zero_row:
            buf[m].zero_()
        out = buf[:m]
        if ldo > n:
            out[:, n:].zero_()
        return out, ldo

    def conv3x3_rows16(self, Y, R, H, W, B, bias=None, relu=False, zero_row=False):
        """act(3 x 3 convolution, padding 1) of the 16-bit NHWC rows Y (Rows16, R H W rows of C channels) with the weights B
        (Rows16 (n, 9 C): ky kx c) as one product: the taps gathered inside the product's operand loads where the library serves
        the shape (odx_gemm_b16_taps: Y.zero_row, C % 64 == 0, a wide layer), else from the written-out neighbourhood matrix
        (taps3x3_16 + gemm_b16).  Returns a Rows16."""
        C, n, m = Y.K, B.n, R * H * W
        dt = Y.buf.dtype
        if Y.n != m or B.K != 9 * C or B.buf.dtype != dt:
            raise ValueError("conv3x3_rows16: %d rows of %d channels against R H W = %d, K = %d" % (Y.n, C, m, B.K))
        if not (Y.zero_row and m and self.lib.odx_gemm_b16_taps_supported(m, n, C, Y.buf.stride(0))):
            return self.gemm_b16(self.taps3x3_16(Y, R, H, W), B, bias=bias, relu=relu, zero_row=zero_row)
        out, ldo = self._rows16_out(m, n, dt, zero_row)
        if bias is not None:
            bias = bias.to(device=self.device, dtype=torch.float32).contiguous()
        hip.check(self.lib.odx_gemm_b16_taps(_p(Y.buf), Y.buf.stride(0), R, H, W, C, _p(B.buf), B.buf.stride(0), n, 1 if dt == torch.bfloat16 else 0,
                                             _p(bias), None, 0, int(bool(relu)), _p(out), ldo, 1, self._stream()), "odx_gemm_b16_taps")
        return Rows16(out, n, zero_row)

    def taps3x3_16(self, Y, R, H, W):
        """The operand of a 3 x 3 convolution (padding 1) as a GEMM over the 16-bit NHWC rows Y (Rows16, R * H * W rows of C
        channels, C % 8 == 0): K = 9 C, tap after tap (odx_taps3x3_16)."""
        C = Y.K
        if Y.n != R * H * W or C % 8 != 0:
            raise ValueError("taps3x3_16: %d rows of %d channels for R, H, W = %d, %d, %d (C %% 8 == 0 required)" % (Y.n, C, R, H, W))
        ld = (9 * C + 127) // 128 * 128
        P = torch.empty((Y.n, ld), dtype=Y.buf.dtype, device=self.device)
        hip.check(self.lib.odx_taps3x3_16(_p(Y.buf), Y.buf.stride(0), R, H, W, C, _p(P), ld, self._stream()), "odx_taps3x3_16")
        return Rows16(P, 9 * C)

    def roi_align_rows(self, feat, rois, spatial_scale, output_size, sampling_ratio=0, step=2):
        """RoIAlign for a head that starts with a stride-`step` 1 x 1 convolution: the bins that convolution reads only,
        as an (R * OH * OW, C) row matrix (NHWC).  Returns (rows, (R, OH, OW))."""
        rois = rois.to(device=self.device, dtype=torch.float32).contiguous()
        N, C, H, W = feat.shape
        PH, PW = output_size
        R = rois.shape[0]
        OH, OW = (PH + step - 1) // step, (PW + step - 1) // step
        out = torch.empty((R * OH * OW, C), dtype=torch.float32, device=self.device)
        feat = feat.to(device=self.device, dtype=torch.float32)
        if (C % 4 == 0 and not feat.is_contiguous() and feat.is_contiguous(memory_format=torch.channels_last) and feat.data_ptr() % 16 == 0):
            # the map is the NHWC row matrix of a trunk run as row GEMMs (a channels-last view): pooled straight from it
            hip.check(self.lib.odx_roi_align_rows_nhwc_f32(_p(feat), C, N, C, H, W, _p(rois), R, float(spatial_scale), PH, PW,
                                                           int(sampling_ratio), int(step), _p(out), self._stream()), "odx_roi_align_rows_nhwc_f32")
            return out, (R, OH, OW)
        feat = feat.contiguous()
        hip.check(self.lib.odx_roi_align_rows_f32(_p(feat), N, C, H, W, _p(rois), R, float(spatial_scale), PH, PW,
                                                  int(sampling_ratio), int(step), _p(out), self._stream()), "odx_roi_align_rows_f32")
        return out, (R, OH, OW)

    # ------------------------------------------------------------------ harvest labelling (odx/harvest.py)
    LABEL_MAX_GT = 64

    def rpn_label(self, gt, anchors, cls, A, neg_thr, pos_thr):
        """Labels of the visible anchors of an image against its ground-truth boxes in two launches (odx_rpn_label_f32): returns
        (ious (n,), assoc (n, 4), neg_mask (n,) bool, over (n,) bool, extra (G, n) bool, counters (2 A + 2 G + G A,) int32 in
        the order RPNHarvester reads them)."""
        gt = gt.to(device=self.device, dtype=torch.float32).contiguous()
        anchors = anchors.to(device=self.device, dtype=torch.float32).contiguous()
        cls = cls.to(device=self.device, dtype=torch.int64).contiguous()
        G, n = int(gt.shape[0]), int(anchors.shape[0])
        ious = torch.empty(n, dtype=torch.float32, device=self.device)
        assoc = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        flags = torch.empty((2 + G, n), dtype=torch.uint8, device=self.device)
        counters = torch.empty(2 * A + 2 * G + G * A + G, dtype=torch.int32, device=self.device)       # (+ G words of scratch)
        nc = 2 * A + 2 * G + G * A
        hip.check(self.lib.odx_rpn_label_f32(_p(gt), G, _p(anchors), _p(cls), n, int(A), float(neg_thr), float(pos_thr), _p(ious), _p(assoc),
                                             _p(flags[0]), _p(flags[1]), _p(flags[2:]), _p(counters), _p(counters[nc:]), self._stream()),
                  "odx_rpn_label_f32")
        fb = flags.view(torch.bool)
        return ious, assoc, fb[0], fb[1], fb[2:], counters[:nc]

    def det_label(self, gt, labels0, proposals, C, img_size, reg_min, neg_thr, in_image):
        """Labels of the proposals of an image against its ground-truth boxes in one launch (odx_det_label_f32): returns (prop
        (R, 4) clamped to the image, overlap (R, C), sel (G, R) bool, cmask (R, n_in) bool or None, counters (G + n_in,) int32).
        labels0 / in_image: DEVICE int32 tensors (the boxes' 0-based classes; the classes whose candidates are wanted)."""
        gt = gt.to(device=self.device, dtype=torch.float32).contiguous()
        proposals = proposals.to(device=self.device, dtype=torch.float32).contiguous()
        G, R, n_in = int(gt.shape[0]), int(proposals.shape[0]), int(in_image.numel())
        prop = torch.empty((R, 4), dtype=torch.float32, device=self.device)
        overlap = torch.empty((R, int(C)), dtype=torch.float32, device=self.device)
        sel = torch.empty((G, R), dtype=torch.uint8, device=self.device)
        cmask = torch.empty((R, n_in), dtype=torch.uint8, device=self.device) if n_in else None
        counters = torch.empty(G + n_in, dtype=torch.int32, device=self.device)
        hip.check(self.lib.odx_det_label_f32(_p(gt), _p(labels0), G, _p(proposals), R, int(C), float(img_size[0]), float(img_size[1]),
                                             float(reg_min), float(neg_thr), _p(in_image) if n_in else None, n_in, _p(prop), _p(overlap), _p(sel),
                                             _p(cmask), _p(counters), self._stream()), "odx_det_label_f32")
        return prop, overlap, sel.view(torch.bool), None if cmask is None else cmask.view(torch.bool), counters

    def box_targets(self, examples, targets):
        """(n, 4) box-regression targets of n (example, target) pairs in one launch (odx_box_targets_f32)."""
        examples = examples.to(device=self.device, dtype=torch.float32).contiguous()
        targets = targets.to(device=self.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(examples)
        hip.check(self.lib.odx_box_targets_f32(_p(examples), _p(targets), int(examples.shape[0]), _p(out), self._stream()), "odx_box_targets_f32")
        return out

    BIAS_ACT_DTYPES = (torch.float32, torch.bfloat16, torch.float16)

    def bias_act_(self, y, bias, residual=None, relu=True):
        """y = act(y + bias[c] (+ residual)) in place over a contiguous NCHW map, f32 / bf16 / f16 (odx_bias_act_nchw_f32 /
        _16): the epilogue of a trunk convolution as one pass.  Returns y."""
        dt = y.dtype
        if dt not in self.BIAS_ACT_DTYPES or not y.is_contiguous() or y.dim() != 4 or bias.dtype != dt or not bias.is_contiguous():
            raise ValueError("bias_act_: contiguous f32 / bf16 / f16 NCHW map and a bias of its dtype expected")
        if residual is not None and (residual.shape != y.shape or residual.dtype != dt or not residual.is_contiguous()):
            raise ValueError("bias_act_: the residual must be a contiguous tensor of the map's shape and dtype")
        N, C, H, W = y.shape
        if bias.numel() != C:
            raise ValueError("bias_act_: %d bias values for %d channels" % (bias.numel(), C))
        rp = _p(residual) if residual is not None else None
        if dt == torch.float32:
            hip.check(self.lib.odx_bias_act_nchw_f32(_p(y), _p(bias), rp, N, C, H * W, 1 if relu else 0, self._stream()),
                      "odx_bias_act_nchw_f32")
        else:
            hip.check(self.lib.odx_bias_act_nchw_16(_p(y), _p(bias), rp, 1 if dt == torch.bfloat16 else 0, N, C, H * W,
                                                    1 if relu else 0, self._stream()), "odx_bias_act_nchw_16")
        return y

    def paste_masks(self, masks, boxes, im_h, im_w, thresh=0.5, padding=1):
        """Masker: masks (R, S, S) f32, boxes (R, 4) -> (R, im_h, im_w) bool."""
        masks = masks.to(device=self.device, dtype=torch.float32).contiguous()
        boxes = boxes.to(device=self.device, dtype=torch.float32).contiguous()
        R, S = masks.shape[0], masks.shape[-1]
        out = torch.empty((R, im_h, im_w), dtype=torch.uint8, device=self.device)
        if R:
            hip.check(self.lib.odx_paste_masks_u8(_p(masks), _p(boxes), R, S, int(im_h), int(im_w), float(thresh), int(padding),
                                                  _p(out), self._stream()), "odx_paste_masks_u8")
        return out.bool()

    def nms(self, boxes, scores, iou_threshold, max_keep=0, sorted_desc=False):
        """Indices of the boxes kept by greedy NMS, in descending score order (maskrcnn_benchmark
        layers.nms contract).  max_keep > 0: only the first max_keep of them (= nms(...)[:max_keep], without visiting the
        candidates behind the last one); sorted_desc: the caller's boxes already are in descending score order (top-k
        output) — no second sort, no gather."""
        boxes = boxes.to(device=self.device, dtype=torch.float32)
        R = boxes.shape[0]
        if R == 0:
            return torch.empty(0, dtype=torch.int64, device=self.device)
        if sorted_desc:
            order, sb = None, boxes.contiguous()
        else:
            order = torch.argsort(scores.to(self.device), descending=True, stable=True)
            sb = boxes.index_select(0, order).contiguous()
        keep = torch.empty(R, dtype=torch.uint8, device=self.device)
        ws = self._workspace("nms", self.lib.odx_nms_workspace_bytes(R))
        if max_keep and max_keep > 0:
            hip.check(self.lib.odx_nms_first_f32(_p(sb), R, float(iou_threshold), int(max_keep), _p(keep), _p(ws), ws.numel(),
                                                 self._stream()), "odx_nms_first_f32")
        else:
            hip.check(self.lib.odx_nms_f32(_p(sb), R, float(iou_threshold), _p(keep), _p(ws), ws.numel(), self._stream()),
                      "odx_nms_f32")
        kept = keep.bool().nonzero().reshape(-1)
        return kept if order is None else order[kept]

    def rpn_topk_decode(self, logits, deltas, anchors, k, img_size, delta_clamp):
        """The k best RPN candidates of every image of a batch by objectness, sorted, decoded and clipped — one launch
        (odx_rpn_topk_decode_f32): logits (B, A, H, W), deltas (B, 4 A, H, W), anchors (H W A, 4); img_size = (width, height).
        Returns boxes (B, k, 4), scores (B, k) = sigmoid(logit), index (B, k) int32 (flat (h W + w) A + a)."""
        logits = logits.to(device=self.device, dtype=torch.float32).contiguous()
        deltas = deltas.to(device=self.device, dtype=torch.float32).contiguous()
        anchors = anchors.to(device=self.device, dtype=torch.float32).contiguous()
        B, A, H, W = logits.shape
        if tuple(deltas.shape) != (B, 4 * A, H, W) or tuple(anchors.shape) != (H * W * A, 4):
            raise ValueError("rpn_topk_decode: deltas %s / anchors %s do not belong to logits %s" % (tuple(deltas.shape), tuple(anchors.shape), tuple(logits.shape)))
        k = int(k)
        if not 0 < k <= min(8192, A * H * W):
            raise ValueError("rpn_topk_decode: k = %d outside 1 .. min(8192, A H W = %d)" % (k, A * H * W))
        boxes = torch.empty((B, k, 4), dtype=torch.float32, device=self.device)
        scores = torch.empty((B, k), dtype=torch.float32, device=self.device)
        index = torch.empty((B, k), dtype=torch.int32, device=self.device)
        hip.check(self.lib.odx_rpn_topk_decode_f32(_p(logits), _p(deltas), _p(anchors), B, A, H, W, k, float(img_size[0]), float(img_size[1]),
                                                   float(delta_clamp), _p(boxes), _p(scores), _p(index), self._stream()), "odx_rpn_topk_decode_f32")
        return boxes, scores, index

    def nms_compact(self, boxes_sorted, keep, P, counts=None):
        """The first P kept boxes of each set as a dense (B, P, 4) block and their counts (B,) int32 (odx_nms_compact_f32)."""
        B, Rmax = int(boxes_sorted.shape[0]), int(boxes_sorted.shape[1])
        boxes_sorted = boxes_sorted.to(device=self.device, dtype=torch.float32).contiguous()
        keep8 = keep.to(device=self.device).to(torch.uint8).contiguous()
        out = torch.empty((B, int(P), 4), dtype=torch.float32, device=self.device)
        n = torch.zeros(B, dtype=torch.int32, device=self.device)
        if counts is not None:
            counts = counts.to(device=self.device, dtype=torch.int32).contiguous()
        if B and P and Rmax:
            hip.check(self.lib.odx_nms_compact_f32(_p(boxes_sorted), _p(keep8), _p(counts), Rmax, B, int(P), _p(out), _p(n), self._stream()),
                      "odx_nms_compact_f32")
        return out, n

    def nms_batched(self, boxes_sorted, counts, iou_threshold, max_keep=0, as_bool=True):
        """Greedy NMS of B independent box sets with one launch pair: boxes_sorted (B, Rmax, 4) f32, set b's counts[b]
        boxes first in its slot, sorted by descending score; counts (B,) int32 ON THE DEVICE.  Returns keep (B, Rmax) bool
        (as_bool=False: the kernel's own uint8 flags).  max_keep > 0: at most that many survivors per set, the first ones (the
        walk of a set stops there)."""
        B, Rmax = int(boxes_sorted.shape[0]), int(boxes_sorted.shape[1])
        keep = torch.empty((B, Rmax), dtype=torch.uint8, device=self.device)
        if B == 0 or Rmax == 0:
            return keep.bool()
        boxes_sorted = boxes_sorted.to(device=self.device, dtype=torch.float32).contiguous()
        counts = counts.to(device=self.device, dtype=torch.int32).contiguous()
        ws = self._workspace("nms_batched", self.lib.odx_nms_batched_workspace_bytes(Rmax, B))
        if max_keep > 0:
            hip.check(self.lib.odx_nms_batched_first_f32(_p(boxes_sorted), _p(counts), Rmax, B, float(iou_threshold), int(max_keep),
                                                         _p(keep), _p(ws), ws.numel(), self._stream()), "odx_nms_batched_first_f32")
        else:
            hip.check(self.lib.odx_nms_batched_f32(_p(boxes_sorted), _p(counts), Rmax, B, float(iou_threshold), _p(keep), _p(ws),
                                                   ws.numel(), self._stream()), "odx_nms_batched_f32")
        return keep.bool() if as_bool else keep


_BACKEND = None


def get_backend():
    """The process-wide HIP backend (created on first use; raises loudly without GPU / library)."""
    global _BACKEND
    if _BACKEND is None:
        _BACKEND = HipBackend()
    return _BACKEND


def set_backend(b):
    """Tests only: install a backend object (e.g. the numpy-oracle backend of tests/)."""
    global _BACKEND
    _BACKEND = b
