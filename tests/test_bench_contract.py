"""The bench line the driver parses (round prompt, "Maintain bench.py" and tier item 4): the committed line of the last
profiled run (profiles/r01_bench_n1.json, written by bench.py on the GPU box) carries every contracted key with the
contracted meaning, and bench.py's command line accepts the driver's flags."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_bench_line():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")))[-1]


def test_committed_bench_line_has_the_contracted_shape():
    line = open(_newest_bench_line()).read().strip().splitlines()[-1]
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
    if os.path.basename(_newest_bench_line()) >= "r04":
        # `roofline` is the kernel family with MORE device time in the timed region (not the family of the single biggest kernel)
        fam = {"mfma": d["roofline_mfma"], "hbm": d["roofline_hbm"]}
        other = fam["hbm" if r["bound"] == "mfma" else "mfma"]
        assert r["family_ms_per_step"] >= other["family_ms_per_step"]
        assert r["kernel"] == fam[r["bound"]]["kernel"] and r["achieved"] == fam[r["bound"]]["achieved"]
        # the whole step against SURVEY 8(d)'s roofline time: F_K / P_mfma + B_CG / BW_hbm + F_pc / P_mfma(f64)
        st, cf = d["roofline_step"], d["config"]
        N, D, M, C = cf["N"], cf["D"], cf["M"], cf["classes"]
        assert abs(st["F_K_flop"] - 2 * 2.0 * N * M * D * C) <= 1e-9 * st["F_K_flop"]
        assert abs(st["F_pc_flop"] - (2.0 * M * M * D + float(M) ** 3) * C) <= 1e-9 * st["F_pc_flop"]
        assert abs(st["B_CG_bytes"] - 21.0 * N * M * st["s_K_bytes_per_entry"] * C) <= 2e-3 * st["B_CG_bytes"]
        t = (st["F_K_flop"] / (st["peaks"]["mfma_TFLOPs"] * 1e12) + st["B_CG_bytes"] / (st["peaks"]["hbm_GBps"] * 1e9)
             + st["F_pc_flop"] / (st["peaks"]["mfma_f64_TFLOPs"] * 1e12)) / d["n_gpus"]
        assert abs(t - st["roofline_time_s"]) < 2e-3 * t
        assert abs(st["achieved"] - st["roofline_time_s"] / (d["ms_per_step"] / 1e3)) < 2e-3 and 0 < st["achieved"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]


def test_bench_accepts_the_driver_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_gpus_n_without_a_launcher_starts_n_ranks(monkeypatch, capsys):
    """`python bench.py --gpus 8` (the driver's command) must start 8 ranks itself: the parent builds ONE
    `python -m torch.distributed.run --nproc-per-node 8 ... bench.py <same flags>` child on 127.0.0.1, relays the
    result line and exits with the child's status — checked here with a stand-in for the child process."""
    import importlib.util
    import io
    import pytest
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class FakeProc:
        def __init__(self, cmd, stdout=None, env=None, text=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = io.StringIO('noise\n{"metric": "m", "n_gpus": 8}\n')
            self.rc = seen.get("rc", 0)

        def wait(self):
            return self.rc

    import subprocess as sp
    monkeypatch.setattr(sp, "Popen", FakeProc)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    bench.main()
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert cmd[-7].endswith("bench.py") and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert capsys.readouterr().out.strip() == '{"metric": "m", "n_gpus": 8}'
    seen["rc"] = 7                                       # a failed child is a failed bench: no retry, same status
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7


def test_roofline_traffic_is_only_quoted_from_passes_of_this_trees_kernel_sources(tmp_path, monkeypatch):
    """`roofline.traffic` comes from committed PMC passes (the counters cannot be read inside the benched process): the passes
    carry the fingerprint of the kernel sources they were taken on (profiles/rNN_pmc_meta.json, tools/profile_round.sh) and a
    tree whose csrc differs — a kernel changed, the profile forgotten — gets null with the reason, not stale bytes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sha = bench.csrc_sha16()
    assert len(sha) == 16 and sha == bench.csrc_sha16()
    prof = tmp_path / "profiles"
    prof.mkdir()
    for counter, v in (("FETCH_SIZE", 1.0e7), ("WRITE_SIZE", 5.0e6)):
        with open(prof / ("r77_pmc_%s_counter_collection.csv" % counter), "w") as f:
            f.write("Kernel_Name,Counter_Name,Counter_Value\n")
            f.write('"void odx::knm_passq_stag_kernel<10, 2, 1>(odx::Args)",%s,%f\n' % (counter, v))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_sha16", lambda: sha)
    got, why = bench.profiled_traffic_gb("knm_passq_stag_kernel<10,2,1>")
    assert got is None and "without a recorded fingerprint" in why
    json.dump({"csrc_sha16": "0" * 16}, open(prof / "r77_pmc_meta.json", "w"))
    got, why = bench.profiled_traffic_gb("knm_passq_stag_kernel<10,2,1>")
    assert got is None and "0000000000000000" in why and sha in why
    json.dump({"csrc_sha16": sha}, open(prof / "r77_pmc_meta.json", "w"))
    got, why = bench.profiled_traffic_gb("knm_passq_stag_kernel<10,2,1>")
    assert got == 25.6 and sha in why
