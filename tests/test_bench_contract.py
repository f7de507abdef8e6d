"""bench.py prints ONE JSON line with the driver's contract keys (plus the roofline / cpu_baseline objects)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic"}


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5", "--envs-per-gpu", "256"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert REQUIRED <= set(d) and ROOFLINE <= set(d["roofline"])
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 5 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "env-steps/s" and d["value"] > 1e6 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0 and 0 < d["roofline"]["frac"] < 1
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
