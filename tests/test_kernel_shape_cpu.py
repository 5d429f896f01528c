"""Occupancy guards for the sweep kernel, read from the compiler's own metadata (`hipcc -S`, no GPU): every
instantiation must fit three waves per SIMD (<= 168 VGPRs, three 4-wave workgroups per CU in 160 KB of LDS).  (Until
the middle of round 3 the instantiations without lists ran in a four-wave shape; with the scalar-instruction diet the
three-wave shape became the faster one for them as well.)  A change that silently costs a wave per SIMD costs ~10 %."""
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
    import __graft_entry__ as g
    subprocess.check_call([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + csrc] +
                          g.HIP_SOURCES["fo_sweep.hip"] +       # the flags the library is built with
                          ["-S", "--cuda-device-only", "-o", out, os.path.join(csrc, "fo_sweep.hip")],
                          stderr=subprocess.DEVNULL)
    txt = open(out).read()
    seen = {}
    for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count: (\d+)",
                         txt, re.S):
        if "fo_sweep_queue_kernel" in m.group(2):
            f = re.search(r"queue_kernelILb(\d)ELi(\d)ELb(\d)ELb(\d)E", m.group(2))
            pair, lists, allm, split = f.group(1) == "1", int(f.group(2)), f.group(3) == "1", f.group(4) == "1"
            seen[(pair, lists, allm, split)] = (int(m.group(1)), int(m.group(3)))
    # (float64 lists, float32 lists, pair scalars, reduced) x (all metrics, subset) x (split, not) + the one instantiation
    # with float32 storage of float64 list entries (FO_LISTS_F32_EXACT: default metric set, full grid)
    assert len(seen) == 17 and (True, 3, True, False) in seen
    for (pair, lists, allm, split), (lds, vgpr) in seen.items():
        assert vgpr <= 168 and 3 * lds <= 160 * 1024, (pair, lists, allm, split, lds, vgpr)
