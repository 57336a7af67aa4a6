"""CPU checks of the coset transforms and the quotient step: definition (Horner at the coset points),
the two oracles against each other, and the property that makes the quotient meaningful."""
import random

import numpy as np
import pytest

import c_oracle as co
import plonk_oracle as po
from plonk_amd import synth

G7 = synth.ints_to_mont([po.GENERATOR])[0]


def test_coset_transform_is_evaluation_on_the_coset():
    rng = random.Random(3)
    N = 32
    f = [rng.randrange(po.P) for _ in range(20)]
    for shift in (po.GENERATOR, 5, pow(3, 77, po.P)):
        ev = po.coset_ntt(f, N, shift)
        w = po.root_of_unity(N)
        for i in (0, 1, 7, N - 1):
            x = shift * pow(w, i, po.P) % po.P
            assert ev[i] == sum(c * pow(x, j, po.P) for j, c in enumerate(f)) % po.P
        assert po.coset_intt(ev, N, shift) == f + [0] * (N - len(f))
        sh = synth.ints_to_mont([shift])[0]
        assert synth.mont_to_ints(co.coset_ntt(synth.ints_to_mont(f), N, sh)) == ev
        assert synth.mont_to_ints(co.coset_intt(synth.ints_to_mont(ev), sh)) == f + [0] * (N - len(f))


@pytest.mark.parametrize("n,N", [(4, 16), (8, 8 * 2), (16, 16 * 16), (64, 256)])
def test_quotient_recovers_the_cofactor(n, N):
    """numerators f_i * Z_H on the coset -> quotient = sum alpha^i f_i, exactly, for deg f_i < N - n."""
    rng = random.Random(n * N)
    alpha = rng.randrange(po.P)
    zh = po.vanishing_on_coset(n, N)
    fs = [[rng.randrange(po.P) for _ in range(N - n)] for _ in range(3)]
    nums = []
    for f in fs:
        ev = po.coset_ntt(f, N)
        nums.append([ev[j] * zh[j % len(zh)] % po.P for j in range(N)])
    want = [sum(pow(alpha, i, po.P) * (f[j] if j < len(f) else 0) for i, f in enumerate(fs)) % po.P for j in range(N)]
    assert po.quotient_polynomial(nums, alpha, n, N) == want
    got = co.quotient([synth.ints_to_mont(x) for x in nums], synth.ints_to_mont([alpha])[0], n, G7)
    assert synth.mont_to_ints(got) == want


def test_gate_quotient_of_a_satisfying_circuit():
    """A trace that satisfies its gates: the gate numerator vanishes on the trace domain, so dividing it by
    Z_H on the coset gives a polynomial t of degree < 3n with t * Z_H = numerator."""
    n, N = 8, 32
    rng = random.Random(8)
    a = [rng.randrange(po.P) for _ in range(n)]
    b = [rng.randrange(po.P) for _ in range(n)]
    c = [rng.randrange(po.P) for _ in range(n)]
    kinds = [rng.choice(["Add", "Mult", "Sub", "Not", "If"]) for _ in range(n)]
    a = [x if k != "If" else rng.randrange(2) for x, k in zip(a, kinds)]
    out = {"Add": lambda x, y, z: x + y, "Sub": lambda x, y, z: x - y, "Mult": lambda x, y, z: x * y,
           "Not": lambda x, y, z: 1 - x, "If": lambda x, y, z: x * y + (1 - x) * z}
    o = [out[k](x, y, z) % po.P for k, x, y, z in zip(kinds, a, b, c)]
    sels = {k: [1 if kk == k else 0 for kk in kinds] for k in set(kinds)}
    table = po.GateEvaluationTable(a, b, c, o, sels)
    gp = po.gate_evaluation_table_to_gate_polynomial(table, n)
    assert all(v == 0 for v in po.build_gate_constraint_evaluation_polynomial(gp, n))      # vanishes on the trace domain
    cos = lambda coef: po.coset_ntt(coef, N)
    w = [cos(x) for x in (gp.input1, gp.input2, gp.input3, gp.output)]
    s = {k: cos(v) for k, v in gp.selectors.items()}
    num = [sum(s[k][j] * po.gate_expression(k, w[0][j], w[1][j], w[2][j], w[3][j]) for k in s) % po.P for j in range(N)]
    t = po.quotient_polynomial([num], 1, n, N)
    assert all(v == 0 for v in t[3 * n:]) and any(t[:3 * n])
    zh = po.vanishing_on_coset(n, N)
    tev = po.coset_ntt(t, N)
    assert [tev[j] * zh[j % 4] % po.P for j in range(N)] == num


def test_vanishing_polynomial_is_zero_on_the_reference_domain():
    """On the plain subgroup (shift 1) - the domain the reference evaluates its numerators on - Z_H is zero at
    every (N/n)-th point, so no quotient exists there: both oracles refuse."""
    assert po.vanishing_on_coset(4, 16, 1)[0] == 0
    with pytest.raises(ZeroDivisionError):
        po.quotient_polynomial([[1] * 16], 1, 4, 16, shift=1)
    with pytest.raises(co.OracleError) as ei:
        co.quotient([synth.ints_to_mont([1] * 16)], synth.ints_to_mont([1])[0], 4, synth.ints_to_mont([1])[0])
    assert ei.value.code == -4
