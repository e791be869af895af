"""bench.py's output contract, checked on the line committed with the round's profiles (profiles/r02_bench.json, produced
on an MI355X by tools/profile_round.sh) and on the tool that reduces the PMC passes."""
import csv
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(REPO, "profiles", "r02_bench.json")))
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    assert d["metric"].split(" (")[0] in base["metric"] and d["unit"] == "frames/s"
    for k, t in (("value", float), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float), ("higher_is_better", bool),
                 ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] in ("weak", "strong")
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    # whole-job value = streams x steps / time
    assert abs(d["value"] - d["config"]["streams_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01


def test_bench_help_and_pmc_summary(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout
    # two launches of one kernel: FETCH_SIZE 100 and 300 KiB, WRITE_SIZE 10 and 30 KiB -> (2 x 200 + 20) KiB per launch
    for name, vals in (("FETCH_SIZE", (100, 300)), ("WRITE_SIZE", (10, 30))):
        d = tmp_path / ("pmc_r1_" + name)
        d.mkdir()
        with open(d / "pmc_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
            for v in vals:
                w.writerow(["some_kernel(int)", name, v])
    res = tmp_path / "out.json"
    run = subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_summary.py"), str(tmp_path), str(res)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    k = json.load(open(res))["kernels"]["some_kernel(int)"]
    assert k["launches"] == 2 and k["hbm_bytes_per_launch"] == (2 * 200 + 20) * 1024
