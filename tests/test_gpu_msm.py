"""GPU tests of the G1 multi-scalar multiplication against the Python oracle (small sizes, arbitrary bases) and
against the closed form sum_i s_i (i + 1) * G for tables of known multiples (large sizes)."""
import numpy as np
import pytest
import torch

import g1_oracle as g1
import plonk_amd as pa
import plonk_oracle as po
from plonk_amd import synth
from plonk_amd.device import to_device, to_host

pytestmark = pytest.mark.gpu


def point_to_limbs(p):
    if p is None:
        return np.zeros(12, dtype=np.uint64)
    return np.array(g1.fq_to_mont_limbs(p[0]) + g1.fq_to_mont_limbs(p[1]), dtype=np.uint64)


def limbs_to_point(a):
    a = [int(x) for x in a]
    if not any(a):
        return None
    return g1.fq_from_mont_limbs(a[:6]), g1.fq_from_mont_limbs(a[6:])


G = point_to_limbs(g1.G1)


def test_multiples_table(prover):
    t = to_host(prover.g1_multiples(G, 1, 300))
    for i in (0, 1, 2, 17, 255, 299):
        assert limbs_to_point(t[i]) == g1.mul(i + 1, g1.G1)
    t0 = to_host(prover.g1_multiples(G, 0, 2))
    assert limbs_to_point(t0[0]) is None and limbs_to_point(t0[1]) == g1.G1
    big = to_host(prover.g1_multiples(G, 2 ** 63 + 12345, 1))
    assert limbs_to_point(big[0]) == g1.mul(2 ** 63 + 12345, g1.G1)


@pytest.mark.parametrize("n", [1, 2, 7, 64, 200])
def test_msm_vs_oracle_arbitrary_bases(ctx, n):
    rng = np.random.default_rng(n)
    ks = [int(x) for x in rng.integers(1, 2 ** 62, size=n)]
    bases = [g1.mul(k * 0x9E3779B97F4A7C15 + 3, g1.G1) for k in ks]
    if n >= 7:
        bases[3] = None                                    # a base at infinity
    scalars = synth.mont_to_ints(synth.random_elements(1000 + n, n))
    if n >= 7:
        scalars[1], scalars[2], scalars[5] = 0, 1, po.P - 1
    pa.msm_set_bases(np.stack([point_to_limbs(b) for b in bases]), ctx=ctx)
    got = pa.msm_g1(synth.ints_to_mont(scalars), ctx=ctx)
    assert limbs_to_point(got) == g1.msm(scalars, bases)


@pytest.mark.parametrize("log_n", [5, 8, 9, 11, 12, 13, 15, 18, 20])   # window widths 5 ... 16, every fold shape
def test_msm_closed_form(prover, log_n):
    """bases (i + 1) * G: the sum is (sum_i s_i (i + 1) mod r) * G, one scalar multiplication in the oracle.
    The sizes cross every window width the implementation picks (7, 10, 13, 16 bits)."""
    n = 1 << log_n
    prover.msm_set_bases(prover.g1_multiples(G, 1, n))
    s = synth.random_elements(77 + log_n, n)
    got = prover.msm(to_device(s, prover.device))
    total = sum(v * (i + 1) for i, v in enumerate(synth.mont_to_ints(s))) % g1.R_ORDER
    assert limbs_to_point(got) == g1.mul(total, g1.G1)
    # fewer scalars than bases: a prefix of the table
    m = n // 2 + 3
    got = prover.msm(to_device(s[:m], prover.device))
    total = sum(v * (i + 1) for i, v in enumerate(synth.mont_to_ints(s[:m]))) % g1.R_ORDER
    assert limbs_to_point(got) == g1.mul(total, g1.G1)


def test_msm_degenerate(prover, ctx):
    n = 1 << 10
    prover.msm_set_bases(prover.g1_multiples(G, 1, n))
    zero = torch.zeros((n, 4), dtype=torch.int64, device=prover.device)
    assert limbs_to_point(prover.msm(zero)) is None                                   # all scalars zero
    same = to_device(np.tile(synth.ints_to_mont([12345]), (n, 1)), prover.device)     # every scalar in the same buckets
    assert limbs_to_point(prover.msm(same)) == g1.mul(12345 * (n * (n + 1) // 2), g1.G1)
    # scalars that cancel: s * P + (r - s) * P = infinity
    pa.msm_set_bases(np.stack([G, G]), ctx=ctx)
    assert limbs_to_point(pa.msm_g1(synth.ints_to_mont([5, po.P - 5]), ctx=ctx)) is None
    with pytest.raises(pa.LengthMismatch):                                            # more scalars than bases
        pa.msm_g1(synth.ints_to_mont([1, 2, 3]), ctx=ctx)
    assert limbs_to_point(pa.msm_g1(synth.ints_to_mont([]).reshape(0, 4), ctx=ctx)) is None


def test_msm_vs_c_oracle(prover):
    """2^14 terms over a table that does not start at G, against the C restatement of the bucket method."""
    import c_oracle as co
    n = 1 << 14
    bases = prover.g1_multiples(G, 0x123456789ABCDEF, n)
    prover.msm_set_bases(bases)
    s = synth.random_elements(4242, n)
    got = prover.msm(to_device(s, prover.device))
    assert np.array_equal(got, co.g1_msm(s, to_host(bases)))


def test_msm_batch(prover, ctx):
    """Several sums over one table at once (two streams inside): each equals the single call, from device and host scalars."""
    n = 1 << 13
    bases = prover.g1_multiples(G, 3, n)
    prover.msm_set_bases(bases)
    cols = [synth.random_elements(900 + k, n) for k in range(5)]
    dev = [to_device(c, prover.device) for c in cols]
    single = np.stack([prover.msm(d) for d in dev])
    assert np.array_equal(prover.msm_batch(dev), single)
    assert np.array_equal(prover.msm_batch(dev[:1]), single[:1])
    assert prover.msm_batch([]).shape == (0, 12)
    pa.msm_set_bases(to_host(bases), ctx=ctx)
    assert np.array_equal(pa.msm_g1_batch(cols, ctx=ctx), single)


@pytest.mark.parametrize("log_n", [10, 14, 17])
def test_msm_crowded_buckets(prover, log_n):
    """Scalars whose digits crowd into a few buckets (all equal, all small, mostly zero): every bucket gets lanes in
    proportion to its run, and a run of more than 64 lanes' worth is added up by a whole workgroup."""
    import time
    n = 1 << log_n
    prover.msm_set_bases(prover.g1_multiples(G, 1, n))
    rng = np.random.default_rng(log_n)
    cases = {
        "all ones": [1] * n,
        "all r - 1": [g1.R_ORDER - 1] * n,
        "below 2^10": [int(v) for v in rng.integers(0, 1 << 10, n)],
        "one in sixteen": [int(v) if k % 16 == 0 else 0 for k, v in enumerate(rng.integers(1, 1 << 62, n))],
        "two values": [0x1234567 if k % 3 else (1 << 200) + 5 for k in range(n)],
    }
    for name, ints in cases.items():
        s = to_device(synth.ints_to_mont(ints), prover.device)
        t0 = time.perf_counter()
        got = prover.msm(s)
        dt = time.perf_counter() - t0
        total = sum(v * (i + 1) for i, v in enumerate(ints)) % g1.R_ORDER
        assert limbs_to_point(got) == (g1.mul(total, g1.G1) if total else None), name
        assert dt < 0.5, (name, dt)           # one lane per bucket would take seconds at the larger sizes
