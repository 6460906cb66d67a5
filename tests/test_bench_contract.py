"""bench.py's output contract (one JSON line on stdout with the driver's keys plus `roofline` and `cpu_baseline`),
checked by running it for two short steps on the GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_emits_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-clips", "2",
                          "--cpu-threads", "8", "--batch", "8", "--profile-steps", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["unit"] == "clips/s" and r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["dtype"] == "f32" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] > 0 and abs(r["value"] - 8 * 2 / (r["ms_per_step"] * 2e-3)) / r["value"] < 0.01
    rf = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0
