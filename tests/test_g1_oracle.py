"""CPU checks of the commitment oracles: the curve constants, the Python bucket method against naive
double-and-add, the C restatement against Python, and the host side of the device arithmetic (g1.cuh is
`__host__ __device__`) against Python."""
import os
import random
import shutil
import subprocess

import numpy as np
import pytest

import c_oracle as co
import g1_oracle as g1
from plonk_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def point_to_limbs(p):
    return np.zeros(12, dtype=np.uint64) if p is None else np.array(g1.fq_to_mont_limbs(p[0]) + g1.fq_to_mont_limbs(p[1]), dtype=np.uint64)


def limbs_to_point(a):
    a = [int(x) for x in a]
    return None if not any(a) else (g1.fq_from_mont_limbs(a[:6]), g1.fq_from_mont_limbs(a[6:]))


def test_curve_constants():
    assert g1.on_curve(g1.G1)
    assert g1.mul(g1.R_ORDER, g1.G1) is None                       # the generator has the scalar field's order
    assert g1.mul(g1.R_ORDER - 1, g1.G1) == g1.neg(g1.G1)
    import plonk_oracle as po
    assert g1.R_ORDER == po.P                                      # scalars of G1 are the field of the polynomial path
    assert g1.Q.bit_length() == 381 and (g1.Q * g1.Q_INV32 + 1) % (1 << 32) == 0


def test_bucket_method_is_the_sum():
    rng = random.Random(2)
    bases = [g1.mul(rng.randrange(g1.R_ORDER), g1.G1) for _ in range(20)] + [None]
    scalars = [rng.randrange(g1.R_ORDER) for _ in range(21)]
    scalars[0], scalars[1], scalars[2] = 0, 1, g1.R_ORDER - 1
    want = g1.msm_naive(scalars, bases)
    for window in (3, 8, 13):
        assert g1.msm(scalars, bases, window) == want
    got = co.g1_msm(synth.ints_to_mont(scalars), np.stack([point_to_limbs(b) for b in bases]))
    assert limbs_to_point(got) == want
    assert limbs_to_point(co.g1_msm(synth.ints_to_mont([5, g1.R_ORDER - 5]), np.stack([point_to_limbs(g1.G1)] * 2))) is None


def test_c_msm_closed_form_at_4096():
    """bases (i + 1) G built in Python by repeated addition; C bucket method against one scalar multiplication."""
    n = 4096
    pts, acc = [], None
    for _ in range(n):
        acc = g1.add(acc, g1.G1)
        pts.append(acc)
    s = synth.random_elements(5, n)
    got = co.g1_msm(s, np.stack([point_to_limbs(p) for p in pts]))
    total = sum(v * (i + 1) for i, v in enumerate(synth.mont_to_ints(s))) % g1.R_ORDER
    assert limbs_to_point(got) == g1.mul(total, g1.G1)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_device_point_arithmetic_on_the_host(tmp_path):
    exe = str(tmp_path / "test_g1_host")
    subprocess.run(["hipcc", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "test_g1_host.hip")], check=True,
                   capture_output=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, check=True).stdout
    want = {"1": 1, "2": 2, "3": 3, "5": 5, "dbl_via_add": 2, "dbl_via_madd": 2, "zero_via_madd": 0, "zero_via_add": 0,
            "inf_plus_g": 1, "k=0xdeadbeefcafe1234": 0xDEADBEEFCAFE1234, "sum": 7919 * 5050,
            # the 28-bit-limb working form (fq28.cuh), converted back through the ABI form
            "w1": 1, "w2": 2, "w3": 3, "w5": 5, "w_dbl_via_add": 2, "w_dbl_via_madd": 2, "w_zero_via_madd": 0, "w_zero_via_add": 0,
            "w_inf_plus_g": 1, "w_zero_lazy": 0, "w_sum": 20100, "w_2^300": 2 ** 300,
            "w_chain": 3 ** 50 * 2 ** 300 + (3 ** 50 - 1) // 2}
    seen = 0
    for line in out.splitlines():
        tag, *rest = line.split()
        p = g1.mul(want[tag] % g1.R_ORDER, g1.G1) if want[tag] else None
        got = None if rest[0] == "inf" else (int(rest[0], 16), int(rest[1], 16))
        assert got == p, tag
        seen += 1
    assert seen == len(want)


def test_host_point_sum_of_the_library():
    """plk_g1_sum runs on the host (it adds up the partial sums of a sharded multi-scalar multiplication): against the oracle."""
    import plonk_amd as pa
    from plonk_amd.api import g1_sum
    rng = random.Random(9)
    ks = [rng.randrange(1, 1 << 40) for _ in range(9)]
    pts = [g1.mul(k, g1.G1) for k in ks] + [None, g1.neg(g1.mul(ks[0], g1.G1)), g1.mul(ks[1], g1.G1)]   # infinity, an inverse, a repeat
    got = g1_sum(np.stack([point_to_limbs(p) for p in pts]))
    assert limbs_to_point(got) == g1.mul((sum(ks) - ks[0] + ks[1]) % g1.R_ORDER, g1.G1)
    assert limbs_to_point(g1_sum(np.zeros((0, 12), dtype=np.uint64))) is None
    assert limbs_to_point(g1_sum(np.stack([point_to_limbs(pts[0]), point_to_limbs(g1.neg(pts[0]))]))) is None
