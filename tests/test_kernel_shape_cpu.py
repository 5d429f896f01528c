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
    # (float64 lists, float32 lists, float32 storage of float64 list entries -- FO_LISTS_F32_EXACT, the headline format, which
    # has every form the others have since round 6 --, pair scalars, reduced) x (all metrics, subset) x (split, not)
    assert len(seen) == 20 and all((True, 3, allm, split) in seen for allm in (False, True) for split in (False, True))
    for (pair, lists, allm, split), (lds, vgpr) in seen.items():
        assert vgpr <= 168 and 3 * lds <= 160 * 1024, (pair, lists, allm, split, lds, vgpr)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_scene_kernels_have_no_private_segment(tmp_path):
    """A kernel with a private segment (a stack or spills in scratch memory) is dispatched late: the rule kernel once kept its
    by-value RuleView argument on a stack -- 284 B per lane, more than the runtime keeps allocated between dispatches at 1 024
    threads per workgroup -- and every launch paid ~11 us for it.  None of the scene-stage kernels may
    have one; the horizon-split instantiations of the sweep (the small planning step) may hold a few spilled registers at most."""
    csrc = os.path.join(ROOT, "frenetix-occlusion_amd", "csrc")
    import __graft_entry__ as g
    for src in ("fo_scene.hip", "fo_sweep.hip"):
        out = str(tmp_path / (src + ".s"))
        subprocess.check_call([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + csrc] +
                              g.HIP_SOURCES[src] + ["-S", "--cuda-device-only", "-o", out, os.path.join(csrc, src)],
                              stderr=subprocess.DEVNULL)
        txt = open(out).read()
        n = 0
        for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+)", txt, re.S):
            name, scratch = m.group(1), int(m.group(2))
            if src == "fo_scene.hip":
                assert scratch == 0, (name, scratch)
                n += 1
            elif "fo_sweep_queue_kernel" in name and re.search(r"queue_kernelILb0ELi0ELb\dELb1E", name):   # reduced outputs, split
                # (round 5: the plain -O3 allocation leaves these instantiations seven spilled VGPRs = 32 B of scratch per lane,
                # where round 4's backend options left none.  Measured on the step they serve, one box, back to back: 0.0735 ms
                # with the 32 B, 0.0743 ms without -- a segment this small at 256 threads per workgroup stays allocated between
                # dispatches; what cost the rule kernel 11 us was 284 B per lane at 1 024 threads.  The bound keeps it small.)
                assert scratch <= 48, (name, scratch)
                n += 1
        assert n >= (12 if src == "fo_scene.hip" else 2), (src, n)
