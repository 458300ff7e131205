"""GPU: bench.py prints exactly one JSON line with the fields the driver reads
(metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better /
scaling / vs_baseline / dtype / data / config.workload, plus roofline and cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "3",
                        "--minutes", "2", "--cpu-seconds", "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"].startswith("Msamples/s scanned") and d["unit"] == "Msamples/s"
    assert d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # value = samples per step / time per step
    samples = d["config"]["frames_per_gpu"] * d["config"]["channels"]
    assert abs(d["value"] - samples / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms_mean"] * 1e-3) / 1e9) / rf["achieved"] < 1e-2
    assert rf["algorithmic_bytes_per_launch"] == samples * 4
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "Msamples/s" and cb["value"] > 0 and cb["sample"]
