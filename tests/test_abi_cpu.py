"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/fo_hip.h
declares; the host layer refuses to run without a GPU (no silent fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "fo_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fo_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as g
    g.build()
    from frenetix_occlusion import _native
    return _native


def test_library_exports_every_declared_symbol(native):
    lib = native.load()
    declared = _declared()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fo_hip.h but not exported"
    assert sorted(native.EXPORTS) == declared
    assert lib.fo_abi_version() == 12
    import __graft_entry__ as g
    assert native.build_id() == g.source_id()      # the library in the tree is the tree's


def test_enums_match_header(native):
    txt = open(os.path.join(ROOT, "include", "fo_hip.h")).read()
    assert f"FO_NPF = {native.NPF}" in txt and f"FO_NPI = {native.NPI}" in txt
    assert f"FO_NL = {native.NL}" in txt and f"FO_NC = {native.NC}" in txt


def test_no_gpu_means_loud_failure(native):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.NativeError):
        native.Context(0)
    from frenetix_occlusion.sweep import MetricSweep
    with pytest.raises(RuntimeError):
        MetricSweep((4.5, 1.6, 1.4, 1000.0, 11.5), 0.1)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "frenetix-occlusion_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert "fo_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_golden_fixture_type_codes_are_the_product_codes():
    import golden_util
    from frenetix_occlusion import _native as native
    assert golden_util.TYPE_CODES == native.TYPE_CODES


def test_c_example_compiles_as_plain_c(tmp_path):
    """examples/sweep_from_c.c (a host with nothing but a C compiler and the HIP runtime's C API) builds and links
    against the header and the library; without a GPU it must stop at fo_create with its message (no CPU path)."""
    import subprocess
    lib = os.path.join(ROOT, "frenetix-occlusion_amd", "lib")
    exe = str(tmp_path / "sweep_from_c")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-O2", os.path.join(ROOT, "examples", "sweep_from_c.c"),
                           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-L" + lib, "-lfo_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe])
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "fo_create failed" in r.stderr


def test_ctypes_structures_have_the_layout_of_the_header(native, tmp_path):
    """the ctypes mirrors of fo_step_t / fo_spawn_rule_params_t / fo_rule_agent_types_t (frenetix_occlusion/_native.py) against
    what a C compiler makes of include/fo_hip.h: size and the offset of every member"""
    import ctypes as C
    import subprocess
    fields = [n for n, *_ in native.Step._fields_]
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "fo_hip.h"', 'int main(void) {',
             'printf("%zu %zu %zu\\n", sizeof(fo_step_t), sizeof(fo_spawn_rule_params_t), sizeof(fo_rule_agent_types_t));']
    lines += ['printf("%%zu\\n", offsetof(fo_step_t, %s));' % f for f in fields]
    lines += ['return 0; }']
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", str(src), "-I" + os.path.join(ROOT, "include"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()
    assert [int(v) for v in out[:3]] == [C.sizeof(native.Step), C.sizeof(native.SpawnRuleParams), C.sizeof(native.RuleAgentTypes)]
    assert [int(v) for v in out[3:]] == [getattr(native.Step, f).offset for f in fields]
