"""The two inversion routines of plonk_amd/csrc/field.cuh are `__host__ __device__`: this builds their
host side (hipcc, no GPU needed) and checks the batched binary GCD against the bit-at-a-time loop
and against x * x^-1 = 1 on 100k random and special values (tools/test_inverse.hip)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_batched_gcd_inverse_on_host(tmp_path):
    exe = str(tmp_path / "test_inverse")
    subprocess.run(["hipcc", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "test_inverse.hip")], check=True,
                   capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout
