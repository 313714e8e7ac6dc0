"""The driver's contract with `bench.py`: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line whose keys are the
ones the round prompt names (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling /
vs_baseline / dtype / data / config.workload) plus the `roofline` and `cpu_baseline` objects of this tier, measured live.  A
short run of the real script as a child process (no side legs but the two required objects)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3", "--no-sub", "--no-loop",
           "--no-stress", "--no-live-pmc", "--cpu-seconds", "2"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert base["metric"].startswith(j["metric"]) and j["unit"] == "check-ins/s"      # (BASELINE's metric string also names the attention roofline: `roofline*`)
    assert j["n_gpus"] == 1 and j["steps"] == 10 and j["warmup"] == 3 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["data"].startswith("synthetic") and isinstance(j["dtype"], str)
    assert j["value"] > 0 and j["ms_per_step"] > 0 and "workload" in j["config"] and "model" not in j["config"]
    # 16 check-ins per step: value and ms_per_step describe the same timed region
    assert abs(j["value"] * j["ms_per_step"] / 1e3 - 16.0) < 0.05 * 16.0
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0 and r["achieved"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == j["unit"] and c["sample"]
    h = j["host_stalls"]
    assert h["max_step_gap_ms"] >= 0 and isinstance(h["gaps_over_threshold"], list)
    # a short timed region is honoured exactly, with a 200-step measurement of the same loop beside it
    assert j["long_run"] is not None and j["long_run"]["steps"] == 200
    assert j["parity"]["worst_max_abs_logit_err"] < 2e-2
