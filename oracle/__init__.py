"""CPU oracle for the hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import anything from this package, and only as the checker / the timed CPU
baseline.  The product path (``online-detection_amd/``) never imports it and fails
loudly when its HIP library is missing.

Parity pin status:
  * ``rls_ref``  (A7/A8)  — pinned: checked against golden vectors produced by the
    reference's own RegionRefinerTrainer / RegionPredictor (tests/golden/make_golden.py).
  * ``bootstrap_ref`` state machine (A6), ``helpers_ref`` (A1/A2/A10) — pinned the
    same way (reference code executed in the build container, fixtures committed).
  * ``falkon_ref`` (A3/A4/A5/A9) — **parity unpinned by the reference**: the
    arithmetic lives in third-party FalkonML/falkon @ 0d96c685 which is not vendored
    in the reference and not installable offline.  The restatement follows the
    published algorithm (FALKON, NeurIPS'17, Alg. 1) with the wrapper constants of
    the reference's call sites, and is cross-checked against an independent dense
    f64 Nystroem-KRR solve.
"""
