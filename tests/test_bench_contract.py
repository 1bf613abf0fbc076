"""The bench line the driver parses (round prompt, "Maintain bench.py" and tier item 4): the committed line of the last
profiled run (profiles/r01_bench_n1.json, written by bench.py on the GPU box) carries every contracted key with the
contracted meaning, and bench.py's command line accepts the driver's flags."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contracted_shape():
    line = open(os.path.join(ROOT, "profiles", "r01_bench_n1.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].split(" at ")[0] == base["metric"].split(" at ")[0]
    assert d["unit"] == "samples/s" and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["steps"] >= 1 and d["warmup"] >= 0 and d["scaling"] in ("weak", "strong")
    assert d["vs_baseline"] is None                       # BASELINE.md publishes no number for this metric
    assert abs(d["value"] - d["config"]["N"] * 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]


def test_bench_accepts_the_driver_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
