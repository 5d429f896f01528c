"""Occupancy guards for the sweep kernel, read from the compiler's own metadata (`hipcc -S`, no GPU): the full-output
instantiations must fit three waves per SIMD (<= 168 VGPRs, three 4-wave workgroups per CU in 160 KB of LDS), the
instantiations without lists four (<= 128 VGPRs, <= 40 KB).  A change that silently costs a wave per SIMD costs ~10 %."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_sweep_kernel_register_and_lds_budgets(tmp_path):
    out = str(tmp_path / "fo_sweep.s")
    csrc = os.path.join(ROOT, "frenetix-occlusion_amd", "csrc")
    subprocess.check_call([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                           "-S", "--cuda-device-only", "-o", out, os.path.join(csrc, "fo_sweep.hip")],
                          stderr=subprocess.DEVNULL)
    txt = open(out).read()
    seen = {}
    for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count: (\d+)",
                         txt, re.S):
        if "fo_sweep_queue_kernel" in m.group(2):
            flags = re.search(r"queue_kernelI(Lb\dELb\dELb\dELb\dE)", m.group(2)).group(1)
            pair, lists, allm, split = [c == "1" for c in re.findall(r"Lb(\d)E", flags)]
            seen[(pair, lists, allm, split)] = (int(m.group(1)), int(m.group(3)))
    assert len(seen) == 12                                   # (full, pair, reduced) x (all metrics, subset) x (split, not)
    for (pair, lists, allm, split), (lds, vgpr) in seen.items():
        if lists or split:
            assert vgpr <= 168 and 3 * lds <= 160 * 1024, (pair, lists, allm, split, lds, vgpr)
        else:
            assert vgpr <= 128 and 4 * lds <= 160 * 1024, (pair, lists, allm, split, lds, vgpr)
