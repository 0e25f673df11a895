"""bench.py prints ONE JSON line with the fields the driver reads (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload + roofline + cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "16384", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert abs(d["value"] - 16384.0 * 16384.0 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
