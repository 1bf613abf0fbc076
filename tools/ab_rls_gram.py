#!/usr/bin/env python3
"""A/B of the RLS Gram launch (development aid): k-split (ODX_RLS_GRAM_SPLIT) x LDS padding (ODX_RLS_GRAM_LDS_PAD_KB: fewer
resident workgroups per CU, room for the target kernels beside the Grams).  Each combination in a child process (the knobs are
read once per process): the bench extras' RLS figure, best of 8."""
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

combos = [tuple(c.split(":")) for c in sys.argv[1:]] or [("1", "0"), ("2", "0"), ("1", "20"), ("2", "20")]
for split, pad in combos:
    env = dict(os.environ, ODX_RLS_GRAM_SPLIT=split, ODX_RLS_GRAM_LDS_PAD_KB=pad)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print("split", split, "pad_kb", pad, line[-1] if line else out.stderr[-400:], flush=True)
