"""The K=15 and K=24 production kernels keep their metrics in registers: no VGPR spills, no scratch.  (hipcc once hoisted 84
loop-invariant masks out of the K=15 step loop and spilled them; every trellis step then began with scratch reloads behind an
s_waitcnt vmcnt(0).)  Compiles the kernel sources to assembly for gfx950 -- no GPU needed -- and reads the kernel metadata."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ka9q_viterbi_comparison_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def kernel_metadata(source):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S",
                        "-o", out, os.path.join(CSRC, source), "-I", CSRC], check=True, capture_output=True)
        text = open(out).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        body = m.group(2)
        get = lambda key: int(re.search(key + r":\s+(\d+)", body).group(1))
        meta[m.group(1)] = dict(scratch=get(r"\.private_segment_fixed_size"), vgpr_spill=get(r"\.vgpr_spill_count"), vgprs=get(r"\.vgpr_count"))
    return meta


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("source,pattern,min_kernels", [
    ("acs_k15.hip", r"acs_k15_kernel|decode_windowed_k15_kernel", 4),
    ("acs_k24t.hip", r"acs_k24t_pass_[hl]_kernelILb[01]ELi0E", 4),
    # the latency kernels park rows / row batches in registers indexed at compile time: an indexed access would put them into scratch
    ("acs_wave.hip", r"acs_wave_kernel|chainback_wave_kernel", 4),
    ("chainback.hip", r"chainback_rows_seg_kernel", 2),
])
def test_production_kernels_do_not_spill(source, pattern, min_kernels):
    meta = {k: v for k, v in kernel_metadata(source).items() if re.search(pattern, k)}
    assert len(meta) >= min_kernels, sorted(meta)
    for name, m in meta.items():
        assert m["vgpr_spill"] == 0 and m["scratch"] == 0, (name, m)
        assert m["vgprs"] <= 256, (name, m)
