import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "online-detection_amd"))
import torch
from tools import bench_extras as bx
c2 = dict(C=30, n=100_000, D=1024, M=2000, sigma=15.0, lam=1e-5, classes_run=30, seed=1234 + 2)
for name, kw in (("auto", {}), ("f32", dict(storage="f32")), ("u24", dict(storage="u24")), ("bf16", dict(storage="bf16"))):
    r = bx.falkon_config_extra(name="config 2 " + name, **c2, **kw)
    print(name, {k: r[k] for k in r if k in ("s_classes_run", "samples_per_s", "dtype")}, flush=True)
    for k in ("roofline_hbm", "roofline_mfma", "pass", "build"):
        if k in r: print("   ", k, r[k])
r = bx.falkon_config_extra(name="config 4", C=21, n=500_000, D=256, M=2000, sigma=10.0, lam=1e-5, classes_run=21, labels="pixels", seed=1234 + 4)
print("config4", {k: r[k] for k in r if k in ("s_classes_run", "samples_per_s", "dtype")})
r = bx.falkon_config_extra(name="config 4 f32", C=21, n=500_000, D=256, M=2000, sigma=10.0, lam=1e-5, classes_run=21, labels="pixels", seed=1234 + 4, storage="f32")
print("config4 f32", {k: r[k] for k in r if k in ("s_classes_run", "samples_per_s", "dtype")})
