"""bench.py's output contract (the driver parses the LAST stdout line as JSON): every required key with the right
type, one short run on the GPU box, CPU-baseline leg included."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["metric"] == "mel_frames_per_sec" and d["unit"] == "frames/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert isinstance(d["value"], float) and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["workload"].startswith("cfg2_b16") and "model" not in d["config"]
    # value is whole-job valid frames / wall
    assert abs(d["value"] - d["config"]["valid_frames_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    ro = d["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s")
    assert ro["peak"] > 0 and ro["achieved"] > 0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    assert ro["traffic"] is None or ro["traffic"] > 0
    assert ro["launches"] == 3 * 4  # the dominant kernel runs once per decoder layer per timed step
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "frames/s" and cb["sample"]
    chk = d["check_vs_oracle"]
    assert chk["durations_equal"] is True and chk["postnet_max_abs_buckets_pinned"] < 1e-3
