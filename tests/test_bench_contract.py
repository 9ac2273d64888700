"""bench.py keeps the driver's contract: ONE JSON line, last on stdout, with the required keys, `roofline` and
`cpu_baseline` (small batch / few steps so that it runs in seconds)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64",
                          "--cpu-frames", "24", "--pipelined", "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])                                   # the JSON line is the LAST thing on stdout
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "u8" and d["data"] == "synthetic"
    assert d["unit"] == "frames/s" and d["value"] > 1000 and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["achieved"] > 0
    assert r["traffic"] is None or r["traffic"] > 0              # committed PMC figure only for the default batch
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 1 and c["unit"] == "frames/s" and c["sample"]
    assert d["cpu_baseline_all_cores"]["cores"] >= 1
