"""CPU-side checks of the drop-in boundary: the shared library builds, loads, and exports exactly the symbols that
include/boojum_hip.h declares; without a GPU the product fails loudly instead of falling back to CPU code."""
import os
import re

import pytest

import era_boojum_amd as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "boojum_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bj_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    from era_boojum_amd import build
    build.build()
    lib = E.load_library()
    declared = header_symbols()
    assert declared, "no declarations parsed from the header"
    for name in declared:
        assert hasattr(lib, name), "libboojum_hip.so does not export %s" % name
    # the python binding covers the whole header, nothing more
    assert declared == E.exported_symbols()
    assert lib.bj_abi_version() == E.binding.ABI_VERSION == 6


def test_status_strings():
    lib = E.load_library()
    assert lib.bj_status_string(0) == b"ok"
    assert b"device" in lib.bj_status_string(-2)


def test_no_cpu_fallback_without_device():
    lib = E.load_library()
    if lib.bj_device_count() > 0:
        pytest.skip("a HIP device is present; the no-device path is exercised on CPU-only hosts")
    with pytest.raises(E.BoojumHipError):
        E.Context(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure; the product tree must not reference it."""
    pkg = os.path.join(ROOT, "era_boojum_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp", ".inc")):
                txt = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "liboracle", "oracle.h", "orc_"):
                    assert needle not in txt, (f, needle)


def test_generated_sources_are_current():
    """csrc/p2_asm.inc, csrc/gl_asm.inc and csrc/jit_headers.inc are generated (tools/gen_p2_asm.py, tools/gen_gl_asm.py,
    era_boojum_amd/build.py::write_jit_headers) and committed so that a checkout builds with hipcc alone: the committed files must
    be what the generators emit today — in particular the run-time compiler must get the same gl.h the library was built with."""
    import importlib.util
    from era_boojum_amd import build as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "tools", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    assert open(os.path.join(B.CSRC, "p2_asm.inc")).read() == load("gen_p2_asm").generate(), "run python tools/gen_p2_asm.py"
    before = open(os.path.join(B.CSRC, "gl_asm.inc")).read()
    load("gen_gl_asm").main()
    assert open(os.path.join(B.CSRC, "gl_asm.inc")).read() == before, "csrc/gl_asm.inc was stale (regenerated now): commit it"
    inc = open(B.JIT_HEADERS_INC).read()
    B.write_jit_headers()
    assert open(B.JIT_HEADERS_INC).read() == inc, "csrc/jit_headers.inc was stale (regenerated now): commit it"
    for h in B.JIT_HEADERS:
        assert open(os.path.join(B.CSRC, h)).read()[:2000] in inc.replace(')BJRAW"\n    R"BJRAW(', "")
