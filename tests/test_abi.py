"""CPU tests of the boundary: the shared library loads without a GPU and exports exactly what
include/plonk_hip.h declares; entry points that need no device behave."""
import ctypes as C
import os
import re

import pytest

from plonk_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "plonk_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(plk_[a-z0-9_]+)\s*\(", text)) - {"plk_builder_fn"})


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in plonk_hip.h but not exported"


def test_no_device_entry_points():
    lib = _lib.load()
    assert lib.plk_strerror(0) == b"ok"
    assert b"domain size" in lib.plk_strerror(_lib.PLK_ERR_LENGTH)
    assert lib.plk_strerror(-12345) == b"unknown error"
    m, one, r2 = ((C.c_uint64 * 4)() for _ in range(3))
    lib.plk_field_constants(m, one, r2)
    p = sum(int(m[i]) << (64 * i) for i in range(4))
    assert p == 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    assert sum(int(one[i]) << (64 * i) for i in range(4)) == (1 << 256) % p
    assert sum(int(r2[i]) << (64 * i) for i in range(4)) == (1 << 512) % p
    assert lib.plk_ctx_create(0, None) == _lib.PLK_ERR_ARG
    assert lib.plk_last_error(None) == b"" and lib.plk_last_error_row(None) == -1
    lib.plk_ctx_destroy(None)  # no-op


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libplonk_hip.so"))
    with pytest.raises(_lib.LibraryMissing):
        _lib.load()


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "plonk_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "plonk_oracle" not in src and "c_oracle" not in src and "liboracle" not in src, f


def test_limits_in_the_binding_are_the_header_s():
    """Numbers the Python binding repeats from plonk_hip.h (the library ties its own copies to the header with
    static_asserts)."""
    text = open(os.path.join(ROOT, "include", "plonk_hip.h")).read()
    shift = int(re.search(r"#define PLK_NOALLOC_MAX_ROWS \(\(size_t\)1 << (\d+)\)", text).group(1))
    assert _lib.NOALLOC_MAX_ROWS == 1 << shift
    assert _lib.MAX_BATCH == int(re.search(r"#define PLK_MAX_BATCH (\d+)", text).group(1))
    assert _lib.NUM_SELECTORS == int(re.search(r"#define PLK_NUM_SELECTORS (\d+)", text).group(1))


def test_field_instantiations_share_no_symbols_of_their_own():
    """The library holds one compiled instantiation of its entry points per field.  Apart from the C++ runtime's
    and the standard library's own (identical wherever they are instantiated), no external symbol may be defined by
    both objects: the linker would keep one definition for both, silently, and which one is a matter of inlining."""
    import shutil
    import subprocess
    build = os.path.join(ROOT, "plonk_amd", "csrc", "build")
    objs = [os.path.join(build, f"capi_f{k}.o") for k in (0, 1)]
    if not all(os.path.exists(o) for o in objs) or not shutil.which("nm") or not shutil.which("c++filt"):
        pytest.skip("the per-field objects are not in the tree (the library was built elsewhere)")

    def defined(obj):
        out = subprocess.run(["nm", "--defined-only", obj], capture_output=True, text=True, check=True).stdout
        return {ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "WTVDBuR"}

    both = sorted(defined(objs[0]) & defined(objs[1]))
    names = subprocess.run(["c++filt"], input="\n".join(both), capture_output=True, text=True, check=True).stdout.splitlines()
    allowed = ("std::", "void std::", "bool std::", "__gnu_cxx", "operator new", "typeinfo", "vtable", "guard variable",
               "DW.ref.__gxx_personality_v0", "__clang_call_terminate", "decltype", "auto std::", "unsigned long std::", "int std::")
    own = [n for n in names if not n.startswith(allowed)]
    assert own == [], own


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C99 on its own (no C++ in it), and a C program that
    names every declared entry point must link against the library."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no C compiler")
    header = os.path.join(ROOT, "include", "plonk_hip.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wpedantic", "-Werror", "-fsyntax-only", "-x", "c", header], check=True)
    names = sorted(set(re.findall(r"\b(plk_[a-z0-9_]+)\s*\(", open(header).read())))
    src = '#include "plonk_hip.h"\n#include <stdio.h>\nint main(void) {\n  const void* fns[] = {' + ", ".join(f"(const void*){n}" for n in names) + \
          '};\n  printf("%d\\n", (int)(sizeof fns / sizeof fns[0]));\n  return 0;\n}\n'
    lib_dir = os.path.join(ROOT, "plonk_amd")
    if not os.path.exists(os.path.join(lib_dir, "libplonk_hip.so")):
        pytest.skip("library not built")
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "link_all.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "link_all")
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wno-pedantic", "-I", os.path.join(ROOT, "include"), c, "-o", exe, "-L", lib_dir, "-lplonk_hip",
                        f"-Wl,-rpath,{lib_dir}", "-Wl,--unresolved-symbols=ignore-in-shared-libs"], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
        assert int(out) == len(names) and len(names) >= 60


def test_generated_dispatch_is_up_to_date(tmp_path, capsys):
    """field_syms.h and dispatch.cpp are generated from the header: what is committed is what the generator writes now."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_field_dispatch", os.path.join(ROOT, "tools", "gen_field_dispatch.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gen.main(str(tmp_path))
    for name in ("field_syms.h", "dispatch.cpp"):
        assert open(tmp_path / name).read() == open(os.path.join(ROOT, "plonk_amd", "csrc", name)).read(), name
