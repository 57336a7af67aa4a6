"""GPU parity tests: every C-ABI entry point against the CPU oracle and the committed vectors,
bit for bit (integer field arithmetic: the tolerance is zero)."""
import ctypes as C

import numpy as np
import pytest

import c_oracle as co
import plonk_amd as pa
from helpers import case_inputs, decode, encode_like
from plonk_amd import _lib, synth

pytestmark = pytest.mark.gpu

D = pa.Radix2EvaluationDomain


def same(a, b):
    return a.shape == b.shape and np.array_equal(a, b)


def test_extension_is_loaded(ctx):
    """The product path is the HIP library, in-tree."""
    assert _lib.load()._name.endswith("plonk_amd/libplonk_hip.so")
    assert any("libplonk_hip.so" in l for l in open("/proc/self/maps"))


def test_golden_vectors(ctx, vectors):
    for case in vectors["cases"]:
        n, table, perm, beta, gamma = case_inputs(case)
        out = case["outputs"]
        gp = pa.gate_evaluation_table_to_gate_polynomial(table, D(n), ctx=ctx)
        for k, w in enumerate(gp.wires()):
            assert decode(case, out["wire_coeffs"][k]) == encode_like(case, w), (case["name"], "wire", k)
        for k, v in gp.selectors.items():
            assert decode(case, out["sel_coeffs"][k]) == encode_like(case, v), (case["name"], "sel", k)
        z = pa.build_grand_product_evaluation_polynomial(perm, table, beta, gamma, ctx=ctx)
        assert decode(case, out["z"]) == encode_like(case, z), case["name"]
        pp = pa.build_permutation_polynomials(perm, table, beta, gamma, D(n), ctx=ctx)
        assert decode(case, out["z_coeffs"]) == encode_like(case, pp.z)
        assert int(out["z_last"], 16) == synth.mont_to_ints(pp.z_last)[0]
        for k, s in enumerate(pp.sigmas()):
            assert decode(case, out["sigma_coeffs"][k]) == encode_like(case, s)
        for N in case["domains"]:
            g = pa.build_gate_constraint_evaluation_polynomial(gp, D(N), ctx=ctx)
            assert decode(case, out["gate_constraints"][str(N)]) == encode_like(case, g), (case["name"], "gate", N)
            e = pa.build_permutation_constraint_evaluation_polynomial(pp, gp, beta, gamma, D(N), ctx=ctx)
            assert decode(case, out["perm_constraints"][str(N)]) == encode_like(case, e), (case["name"], "perm", N)
            ev = pa.evaluate_over_domain(gp.input1, D(N), ctx=ctx)
            assert decode(case, out["wire0_evals"][str(N)]) == encode_like(case, ev)
        z2, g2, e2 = pa.prove_round23(table, perm, beta, gamma, D(n), D(case["domains"][-1]), ctx=ctx)
        assert decode(case, out["z"]) == encode_like(case, z2)
        assert decode(case, out["gate_constraints"][str(case["domains"][-1])]) == encode_like(case, g2)
        assert decode(case, out["perm_constraints"][str(case["domains"][-1])]) == encode_like(case, e2)


# one-, two- and three-pass transform plans: 2^0..2^11 | 2^12..2^16 | 2^17..
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 10, 11, 12, 13, 16, 17, 18])
def test_transforms_vs_oracle(ctx, log_n):
    n = 1 << log_n
    v = synth.random_elements(1000 + log_n, n)
    assert same(pa.evaluations_to_dense_polynomial(v, D(n), ctx=ctx), co.intt(v, n))
    assert same(pa.evaluate_over_domain(v, D(n), ctx=ctx), co.ntt(v, n))
    assert same(pa.evaluate_over_domain(pa.evaluations_to_dense_polynomial(v, D(n), ctx=ctx), D(n), ctx=ctx), v)
    # zero-padded evaluation on larger domains, incl. a ragged coefficient count
    for m, N in ((n, 4 * n), (max(1, n - 3), 2 * n), (max(1, n // 2), n)):
        assert same(pa.evaluate_over_domain(v[:m], D(N), ctx=ctx), co.ntt(v[:m], N)), (m, N)


@pytest.mark.parametrize("log_n", [2, 6, 10, 12, 16])
def test_each_function_vs_oracle(ctx, log_n):
    """The log_n = 16 configuration: NTT + grand product + gate constraints vs CPU."""
    n = 1 << log_n
    c = synth.make_circuit(n, circuit_id=log_n)
    gp = pa.gate_evaluation_table_to_gate_polynomial(c.table, D(n), ctx=ctx)
    wc, sc = co.gate_polys(c.table.wires(), c.table.selectors, n)
    assert all(same(a, b) for a, b in zip(gp.wires(), wc))
    assert all(same(gp.selectors[k], sc[k]) for k in sc) and set(sc) == set(gp.selectors)
    z = pa.build_grand_product_evaluation_polynomial(c.perm, c.table, c.beta, c.gamma, ctx=ctx)
    assert same(z, co.grand_product(c.table.wires(), c.perm.columns(), c.beta, c.gamma))
    pp = pa.build_permutation_polynomials(c.perm, c.table, c.beta, c.gamma, D(n), ctx=ctx)
    so, zc, zl = co.perm_polys(c.table.wires(), c.perm.columns(), c.beta, c.gamma, n)
    assert same(pp.z, zc) and same(pp.z_last, zl) and all(same(a, b) for a, b in zip(pp.sigmas(), so))
    for N in (n, 4 * n):
        assert same(pa.build_gate_constraint_evaluation_polynomial(gp, D(N), ctx=ctx), co.gate_constraints(wc, sc, N))
        for shift in (1, 4):
            assert same(pa.build_permutation_constraint_evaluation_polynomial(pp, gp, c.beta, c.gamma, D(N), z_shift=shift, ctx=ctx),
                        co.perm_constraints(zc, so, wc, N, c.beta, c.gamma, z_shift=shift))
    z2, g2, e2 = pa.prove_round23(c.table, c.perm, c.beta, c.gamma, D(n), D(4 * n), ctx=ctx)
    zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), 4 * n, c.beta, c.gamma)
    assert same(z2, zr) and same(g2, gr) and same(e2, er)


def test_subset_of_selectors(ctx):
    n = 256
    c = synth.make_circuit(n, kinds=("Add", "Mult", "If"))
    gp = pa.gate_evaluation_table_to_gate_polynomial(c.table, D(n), ctx=ctx)
    assert set(gp.selectors) == {"Add", "Mult", "If"}
    wc, sc = co.gate_polys(c.table.wires(), c.table.selectors, n)
    assert same(pa.build_gate_constraint_evaluation_polynomial(gp, D(4 * n), ctx=ctx), co.gate_constraints(wc, sc, 4 * n))
    # no selectors at all: the constraint vector is zero
    c0 = synth.make_circuit(n, kinds=())
    gp0 = pa.gate_evaluation_table_to_gate_polynomial(c0.table, D(n), ctx=ctx)
    assert not pa.build_gate_constraint_evaluation_polynomial(gp0, D(n), ctx=ctx).any()
    # Int/Bool selectors contribute nothing (src/execution_trace.rs:396-397)
    ci = synth.make_circuit(n, kinds=("Int", "Bool"))
    gpi = pa.gate_evaluation_table_to_gate_polynomial(ci.table, D(n), ctx=ctx)
    assert not pa.build_gate_constraint_evaluation_polynomial(gpi, D(2 * n), ctx=ctx).any()


def test_build_and_convert(ctx):
    v = synth.random_elements(3, 64)
    assert same(pa.build_and_convert(D(64), lambda: v, ctx=ctx), co.intt(v, 64))
    with pytest.raises(pa.LengthMismatch):
        pa.build_and_convert(D(64), lambda: v[:63], ctx=ctx)
    with pytest.raises(pa.LengthMismatch):
        pa.build_and_convert(D(32), lambda: v, ctx=ctx)   # builder returns more than the domain holds


def test_special_values(ctx):
    """zero, one and p-1 columns; transforms of constants and of a delta."""
    n = 64
    zero = np.zeros((n, 4), dtype=np.uint64)
    one = np.tile(synth.ONE_MONT, (n, 1))
    pm1 = synth.ints_to_mont([synth.P_INT - 1] * n)
    assert not pa.evaluate_over_domain(zero, D(n), ctx=ctx).any()
    c = pa.evaluations_to_dense_polynomial(one, D(n), ctx=ctx)   # constant 1 -> coefficient vector (1, 0, ...)
    assert synth.mont_to_ints(c) == [1] + [0] * (n - 1) and pa.degree(c) == 0
    assert same(pa.evaluate_over_domain(pm1, D(4 * n), ctx=ctx), co.ntt(pm1, 4 * n))
    delta = zero.copy(); delta[0] = synth.ONE_MONT
    assert same(pa.evaluate_over_domain(delta[:1], D(n), ctx=ctx), one)
    assert pa.degree(zero) == -1


def test_linearity_of_the_transform(ctx):
    n = 1 << 12
    a, b = synth.random_elements(21, n), synth.random_elements(22, n)
    s = synth.ints_to_mont([(x + y) % synth.P_INT for x, y in zip(synth.mont_to_ints(a[:64]), synth.mont_to_ints(b[:64]))])
    fa, fb = pa.evaluate_over_domain(a[:64], D(n), ctx=ctx), pa.evaluate_over_domain(b[:64], D(n), ctx=ctx)
    fs = pa.evaluate_over_domain(s, D(n), ctx=ctx)
    got = synth.mont_to_ints(fs[:128])
    want = [(x + y) % synth.P_INT for x, y in zip(synth.mont_to_ints(fa[:128]), synth.mont_to_ints(fb[:128]))]
    assert got == want


@pytest.mark.parametrize("log_n", [2, 3, 6, 11, 12, 17])
def test_extend4_equals_padded_transform(prover, log_n):
    """plk_dev_extend4 (three coset transforms + the input column) against interpolation followed by the
    zero-padded 4n-point transform, for field columns and for an index column."""
    from plonk_amd.device import to_device, to_host
    n = 1 << log_n
    c = synth.make_circuit(n, circuit_id=log_n)
    cols = [to_device(c.table.input1, prover.device), to_device(c.table.output, prover.device), to_device(c.perm.sigma2, prover.device)]
    kinds = [_lib.INPUT_FIELD, _lib.INPUT_FIELD, _lib.INPUT_U64]
    got = prover.extend4(cols, n, kinds=kinds)
    want = prover.ntt_batch(prover.ntt_batch(cols, n, inverse=True, kinds=kinds), 4 * n)
    for g, w in zip(got, want):
        assert np.array_equal(to_host(g), to_host(w))
    ref = co.ntt(co.intt(c.table.input1, n), 4 * n)
    assert np.array_equal(to_host(got[0]), ref)


def test_pointwise_kernels_at_extreme_values(prover):
    """The constraint kernels keep sums and differences lazy (values of several p between products): columns made of
    0, 1, p - 1, p - 2, (p - 1) / 2 and random elements, every combination the period lets meet, against big integers."""
    import plonk_oracle as po
    from plonk_amd.device import to_device, to_host
    p, N = po.P, 1 << 10
    special = [0, 1, p - 1, p - 2, (p - 1) // 2, 2, p - 3]
    rng = np.random.default_rng(5)

    def column(period, phase):
        vals = []
        for i in range(N):
            k = (i // period + phase) % (len(special) + 1)
            vals.append(special[k] if k < len(special) else int(rng.integers(0, 1 << 62)) ** 4 % p)
        return vals

    wires = [column(1, 0), column(8, 3), column(64, 5), column(3, 1)]
    sels = [column(5, k) for k in range(9)]
    sigma = [column(7, k + 2) for k in range(4)]
    z = column(11, 4)
    beta, gamma = p - 1, p - 2
    dev = lambda ints: to_device(synth.ints_to_mont(ints), prover.device)
    kinds = pa.AST_KINDS
    got = synth.mont_to_ints(to_host(prover.gate_constraint_eval([dev(w) for w in wires], {k: dev(s) for k, s in zip(kinds, sels)}, N)))
    a, b, c, o = wires
    for i in range(N):
        terms = [0, 0, o[i] - (a[i] + b[i]), o[i] - (a[i] - b[i]), o[i] - a[i] * b[i], o[i] * b[i] - a[i], o[i] * (a[i] - b[i]),
                 o[i] + a[i] - 1, o[i] - (a[i] * b[i] + (1 - a[i]) * c[i])]
        assert got[i] == sum(s[i] * t for s, t in zip(sels, terms)) % p, ("gate", i)
    bm, gm = synth.ints_to_mont([beta])[0], synth.ints_to_mont([gamma])[0]
    for shift in (1, 4):
        got = synth.mont_to_ints(to_host(prover.perm_constraint_eval(dev(z), [dev(s) for s in sigma], [dev(w) for w in wires], N, bm, gm,
                                                                     z_shift=shift)))
        for i in range(N):
            lhs = rhs = 1
            for k in range(4):
                lhs = lhs * (wires[k][i] + beta * i + gamma) % p
                rhs = rhs * (wires[k][i] + beta * sigma[k][i] + gamma) % p
            assert got[i] == (z[(i + shift) % N] * lhs - z[i] * rhs) % p, ("perm", shift, i)


def _extreme_circuit(log_n, kinds):
    n = 1 << log_n
    p = synth.P_INT
    special = [0, 1, p - 1, p - 2, (p - 1) // 2, 2, p - 3]
    rng = np.random.default_rng(100 + log_n)

    def column(period, phase):
        vals = []
        for i in range(n):
            k = (i // period + phase) % (len(special) + 2)
            vals.append(special[k] if k < len(special) else int(rng.integers(0, 1 << 62)) ** 5 % p)
        return synth.ints_to_mont(vals)

    c = synth.make_circuit(n, circuit_id=40 + log_n, kinds=kinds)
    c.table.input1, c.table.input2, c.table.input3, c.table.output = column(1, 0), column(2, 3), column(5, 5), column(3, 1)
    for j, k in enumerate(sorted(c.table.selectors)):
        if j % 2 == 0:
            c.table.selectors[k] = column(4, j)
    return c


EXTREME_CASES = [(3, pa.AST_KINDS), (8, pa.AST_KINDS), (8, ("Add", "Div", "If")), (12, ("Sub", "Mult", "Eq", "Not"))]


@pytest.mark.parametrize("log_n,kinds", EXTREME_CASES)
def test_round23_at_extreme_values(ctx, log_n, kinds):
    """Rounds 2+3 on the 4n-point domain (the coset-major kernels: gate sum regrouped by wire, sigma columns carried
    times beta through their transforms) with wire and selector columns made of 0, 1, 2, p - 1, p - 2, p - 3,
    (p - 1) / 2 and random elements, selectors that are not 0/1 included, against the CPU oracle."""
    n = 1 << log_n
    c = _extreme_circuit(log_n, kinds)
    z, g, e = pa.prove_round23(c.table, c.perm, c.beta, c.gamma, D(n), D(4 * n), ctx=ctx)
    zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), 4 * n, c.beta, c.gamma)
    assert same(z, zr) and same(g, gr) and same(e, er)


@pytest.mark.parametrize("log_n,kinds", EXTREME_CASES)
def test_round23_at_extreme_values_one_kernel(monkeypatch, log_n, kinds):
    """The same circuits through the device-resident call with both constraint vectors taken from the one kernel (what
    the library does from 2^16 rows on, asked for here at every size), several z_shift."""
    from plonk_amd.device import DeviceProver, to_host
    monkeypatch.setenv("PLONK_HIP_FUSED_MIN_LOG", "0")
    prover = DeviceProver(0)
    n = 1 << log_n
    c = _extreme_circuit(log_n, kinds)
    dc = prover.upload(c)
    for z_shift in (1, 4, 4 * n - 1):
        z, g, e = prover.round23(dc, 4 * n, z_shift=z_shift)
        prover.status()
        zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), 4 * n, c.beta, c.gamma, z_shift=z_shift)
        assert np.array_equal(to_host(z), zr) and np.array_equal(to_host(g), gr) and np.array_equal(to_host(e), er), z_shift
