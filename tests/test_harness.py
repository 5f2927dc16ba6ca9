"""The stand-alone C++ harness (harness/viterbi_bench): reference-style flags and JSON schema (src/main.cpp:80-118,
300-316), decoding on the GPU."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "harness", "viterbi_bench")
REF_KEYS = ["name", "K", "R", "poly", "total_input_bytes", "total_transmit_bits", "total_output_symbols", "sampling_time",
            "minimum_samples", "total_samples", "init_ns", "update_ns", "chainback_ns", "total_bits", "total_bit_errors",
            "bit_error_rate"]


def test_flag_validation_matches_reference():
    if not os.path.exists(BIN):
        pytest.skip("harness not built")
    r = subprocess.run([BIN, "-t", "-1"], capture_output=True, text=True)
    assert r.returncode == 1 and "Sampling time must be positive" in r.stderr  # src/main.cpp:344-347


@pytest.mark.gpu
def test_harness_json_schema_and_ber(tmp_path):
    assert os.path.exists(BIN), "build it: make -C harness"
    out = tmp_path / "bench.json"
    r = subprocess.run([BIN, "-t", "0.05", "-n", "2", "-o", str(out), "--frames", "256", "--payload-bytes", "64", "--hard",
                        "--codes", "27,47,29,49,615"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    entries = json.load(open(out))
    assert [e["K"] for e in entries] == [7, 7, 9, 9, 15]
    for e in entries:
        for k in REF_KEYS:
            assert k in e, k
        assert e["name"] == "hip" and e["frames"] == 256
        assert e["total_samples"] == len(e["update_ns"]) == len(e["chainback_ns"]) == len(e["init_ns"]) >= 3
        assert e["total_bit_errors"] == 0  # noise-free input decodes without error
    r = subprocess.run([BIN, "-t", "0.05", "-n", "1", "-o", str(out), "--payload-bytes", "8", "--hard", "--codes", "224"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    e = json.load(open(out))[0]
    assert e["K"] == 24 and e["total_bit_errors"] == 0  # BER from the nbits+K-1 call


@pytest.mark.gpu
def test_reference_style_bindings_decode():
    """harness/adapter_check: the reference's test_third_party call sequence over host buffers, through
    include/hip_interface.h and through a ka9q_interface-shaped five-function adapter."""
    exe = os.path.join(ROOT, "harness", "adapter_check")
    assert os.path.exists(exe), "build it: make -C harness"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all bindings ok" in r.stdout


@pytest.mark.gpu
def test_harness_reference_methodology_mode(tmp_path):
    """--host-api: the reference's own loop (src/main.cpp:257-280) -- one frame per handle, host buffers, the blocking
    reference-shaped calls -- at the reference's frame sizes; same JSON schema, noise-free input decodes without error."""
    out = tmp_path / "host.json"
    r = subprocess.run([BIN, "-t", "0.05", "-n", "2", "-o", str(out), "--host-api", "--hard", "--codes", "27,49,615,224"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    entries = json.load(open(out))
    assert [(e["K"], e["R"]) for e in entries] == [(7, 2), (9, 4), (15, 6), (24, 2)]
    for e in entries:
        for k in REF_KEYS:
            assert k in e, k
        assert e["name"] == "hip_host_1frame" and e["frames"] == 1
        assert e["total_input_bytes"] == {7: 1024, 9: 512, 15: 256, 24: 8}[e["K"]]  # src/main.cpp:366,395,404,414
        assert e["total_samples"] == len(e["update_ns"]) >= 3 and e["total_bit_errors"] == 0


def test_harness_rejects_zero_time_and_zero_samples():
    """src/main.cpp:344-351: sampling time must be positive, minimum samples non-zero (checked before any device use)."""
    for flags in (["-t", "0"], ["-t", "-1"], ["-n", "0"]):
        r = subprocess.run([BIN] + flags, capture_output=True, text=True, timeout=60)
        assert r.returncode == 1, (flags, r.returncode, r.stderr)


@pytest.mark.gpu
def test_harness_gpus_argument(tmp_path):
    """--gpus N shards the frames over N devices (one host thread and one stream per device, hipSetDevice per shard).  One GPU is
    what the test box has: --gpus 1 runs the sharded path with a single shard (same JSON, `gpus` 1, frames_per_gpu = frames) and
    asking for more devices than are visible is refused before anything is decoded."""
    out = tmp_path / "g.json"
    r = subprocess.run([BIN, "-t", "0.05", "-n", "2", "-o", str(out), "--gpus", "1", "--frames", "512", "--payload-bytes", "32", "--hard",
                        "--codes", "27,615"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for e in json.load(open(out)):
        assert e["gpus"] == 1 and e["frames"] == 512 and e["frames_per_gpu"] == 512 and e["total_bit_errors"] == 0
    r = subprocess.run([BIN, "-t", "0.05", "-n", "2", "-o", str(out), "--gpus", "64", "--codes", "27"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "device(s) visible" in r.stderr
