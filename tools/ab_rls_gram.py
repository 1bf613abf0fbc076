#!/usr/bin/env python3
"""A/B of the RLS call's forms (development aid): the Gram kernel's k-tile (ODX_RLS_GRAM_BK = 32: floats in LDS, the default;
16: round 4's f64 form) x the targets' products (ODX_RLS_RAW_TARGETS = 1: inside the Gram sweep; 0: a second sweep).  Each
combination runs in a child process (the knobs are read once per process) and prints the bench extras' RLS figure (best of 2 x 8).
python tools/ab_rls_gram.py [BK:RAW ...]     default: 32:1 16:1 32:0"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = """
import sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import odx; odx.get_backend()
import bench_extras as b
r = b.rls_extra(cpu=False)
r2 = b.rls_extra(cpu=False)
print(json.dumps({"ms": min(r["ms"], r2["ms"])}))
""" % (ROOT, os.path.join(ROOT, "online-detection_amd"), os.path.join(ROOT, "tools"))

combos = [tuple(c.split(":")) for c in sys.argv[1:]] or [("32", "1"), ("16", "1"), ("32", "0")]
for bk, raw in combos:
    env = dict(os.environ, ODX_RLS_GRAM_BK=bk, ODX_RLS_RAW_TARGETS=raw)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print("k-tile", bk, "raw targets", raw, line[-1] if line else out.stderr[-400:], flush=True)
