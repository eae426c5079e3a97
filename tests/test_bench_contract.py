"""The committed bench line (profiles/r02/bench_default.json, produced by `python bench.py` on an MI355X) keeps the driver's contract."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    line = open(os.path.join(ROOT, "profiles", "r02", "bench_default.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                 ("cpu_baseline", dict)):
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert "workload" in d["config"] and not {"model", "global_batch", "seq_len", "encoder"} & set(d["config"])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].split(",")[0] in base["metric"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert abs(d["value"] - d["config"]["frames_per_step_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # round 2: the default workload is BASELINE's metric row (M = 10 000) and says how honest the synthetic detections are
    assert d["config"]["memory_instances"] == 10000 and d["config"]["workload"].startswith("T:")
    assert d["det_points_after_outlier_mean"] >= 4000 and d["assignment_correct_rate"] >= 0.95 and d["registered_given_correct_assignment"] >= 0.95


def test_bench_gpus_2_starts_two_ranks_and_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` (no launcher, no WORLD_SIZE) must start two ranks as child processes and exit non-zero when they fail --
    on this GPU-less box every rank stops with the "needs an MI355X" message (--gpus was parsed and never read)"""
    import subprocess
    import sys
    import pytest
    import torch
    if torch.cuda.device_count() > 0:      # on a GPU box the run would start real bench ranks next to this process: nothing to check here
        pytest.skip("checks the GPU-less failure path")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "starting 2 ranks" in out.stderr and "--nproc-per-node=2" in out.stderr
    # torchrun terminates the second rank as soon as the first one exits, so only ONE message is guaranteed
    assert out.stderr.count("needs an MI355X") >= 1, out.stderr[-2000:]
    assert not out.stdout.strip().startswith("{")           # no JSON line from a failed run


def test_every_config_preset_parses_and_fails_loudly_without_a_gpu():
    """`bench.py --config X` resolves its preset (model, memory, points, frames / steps defaults, flags) and then stops with the no-GPU message:
    no preset may die in argument handling before it reaches the device"""
    import subprocess
    import sys
    import pytest
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("checks the GPU-less failure path")
    for cfg in ("C1", "C2", "C3", "C4", "C5", "T"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg], capture_output=True, text=True, timeout=120)
        assert out.returncode != 0 and "needs an MI355X" in out.stderr, (cfg, out.stderr[-500:])
