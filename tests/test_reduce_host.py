"""fe_reduce_f9 (plonk_amd/csrc/fr29.cuh) estimates a quotient in floating point: its host side is checked against the
plain chain of conditional subtractions at every multiple of p below 32 p, their neighbours, and two million random
lazy limb vectors (tools/test_reduce.hip, built with hipcc, no GPU needed)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_quotient_estimate_reduction_on_host(tmp_path):
    exe = str(tmp_path / "test_reduce")
    subprocess.run(["hipcc", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "test_reduce.hip")], check=True,
                   capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr
