"""Runs the C++ host mirror's own test program (tests/cpp/test_mirror.cpp, built in-tree by
__graft_entry__.build() / `make -C tests/cpp`) on the GPU: the seven reference-named functions, the fused
call and the two table-building steps, through include/plonk_hip.hpp, against the C oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_mirror")


@pytest.mark.gpu
def test_cpp_mirror_program():
    assert os.path.exists(EXE), "tests/cpp/test_mirror missing: run `make -C tests/cpp` (or __graft_entry__.build())"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "0 failure(s)" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_constraint_limb_arithmetic_on_the_device():
    """The 29-bit-limb arithmetic of the constraint kernels against the 32-bit field functions, on the GPU, stage by stage
    (it once differed from the host compile of the same source: the compiler narrowed multiplies of limbs it knew to be short)."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_constraints_dev")
    assert os.path.exists(exe), "tests/cpp/test_constraints_dev missing: run `make -C tests/cpp` (or __graft_entry__.build())"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "mismatch at" not in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


def test_cpp_mirror_header_compiles_alone(tmp_path):
    """the header is self-contained C++17 (syntax check only: no GPU, no library needed)."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "plonk_hip.hpp"\nint main() { plonk_hip::GeneralEvaluationDomain d(8); return (int)d.size() - 8; }\n')
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)], check=True,
                   capture_output=True, timeout=120)
