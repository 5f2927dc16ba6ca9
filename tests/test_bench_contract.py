"""bench.py prints ONE JSON line with the keys the driver depends on (plus `roofline` and `cpu_baseline`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--frames", "4096", "--cpu-budget", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["unit"] == "Msymbols/s" and d["dtype"] == "u8"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0
    assert d["value"] > 0 and d["bit_errors"] >= 0
    # value is consistent with ms_per_step and the workload size
    syms = 4096 * (2048 + 6) * 2
    assert abs(d["value"] - syms / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.02


@pytest.mark.gpu
def test_bench_default_line_carries_every_single_gpu_config():
    """With no workload flags the one JSON line holds BASELINE.json configs[1] as the headline and configs[2] (K=15 x 4096
    frames) and configs[3] (K=24, one 2048-bit frame) under extra.configs, each with its own value / kernel rates /
    roofline / cpu_baseline / bit errors."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--cpu-budget", "1"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "65536 frames" in d["config"]["workload"] and d["steps"] == 3
    extras = d["extra"]["configs"]
    assert [("K=15" in e["config"]["workload"], "K=24" in e["config"]["workload"]) for e in extras] == [(True, False), (False, True)]
    for e in extras:
        assert "error" not in e, e
        for k in ("value", "ms_per_step", "update_msym_s", "chainback_mbit_s", "roofline", "cpu_baseline", "bit_errors", "steps", "warmup"):
            assert k in e, k
        assert e["value"] > 0 and e["roofline"]["frac"] > 0 and e["cpu_baseline"]["value"] > 0
        assert e["roofline"]["bytes_per_launch"] > 0 and e["roofline"]["algorithmic_bytes_per_launch"] > 0
    assert "4096 frames" in extras[0]["config"]["workload"] and extras[0]["dtype"] == "i16"
    assert "1 frames" in extras[1]["config"]["workload"] and extras[1]["bit_errors"] == 0  # K=24 at 4 dB: error free
    # the reference's own methodology (one frame per call, host buffers) through the five-function C ABI, per decoder
    one = d["extra"]["one_frame_handles"]
    assert "error" not in one, one
    assert [r["code"] for r in one["decoders"]] == ["27", "47", "29", "49", "615", "224"]
    for r in one["decoders"]:
        assert "error" not in r, r
        assert r["decodes"] >= 3 and r["update_msym_s"] > 0 and r["chainback_mbit_s"] > 0
    assert one["decoders"][-1]["bit_errors"] == 0
