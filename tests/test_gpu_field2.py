"""GPU parity tests of the second field instantiation (field 1, the scalar field of Tweedledee): the same entry
points, the same kernels compiled over another modulus, against the C restatement compiled over that modulus
(oracle/liboracle_tweedle.so) and against big integers.  Bar: bit-exact.  The field's generator and root of unity are
self-derived (tools/derive_field_constants.py): agreement with arkworks for this field is unverifiable."""
import numpy as np
import pytest
import torch

import c_oracle as co
import plonk_amd as pa
import plonk_oracle as po
from plonk_amd import _lib, synth
from plonk_amd.device import DeviceProver, to_device, to_host

pytestmark = pytest.mark.gpu

D = pa.Radix2EvaluationDomain
F1 = synth.field(1)


@pytest.fixture(scope="module")
def ctx1():
    c = pa.Context(0, field="tweedle_fr")
    yield c
    c.close()


@pytest.fixture(scope="module")
def prover1():
    return DeviceProver(0, field=1)


@pytest.fixture(autouse=True)
def tweedle_oracle():
    with co.use_field(1), po.use_field(1):
        yield


def same(a, b):
    return a.shape == b.shape and np.array_equal(a, b)


def test_context_reports_its_field(ctx1, ctx):
    assert ctx1.lib.plk_ctx_field(ctx1.handle) == 1 and ctx.lib.plk_ctx_field(ctx.handle) == 0


# one-, two- and three-pass transform plans
@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 10, 11, 12, 16, 17, 18])
def test_transforms_vs_oracle(ctx1, log_n):
    n = 1 << log_n
    v = F1.random_elements(2000 + log_n, n)
    assert same(pa.evaluations_to_dense_polynomial(v, D(n), ctx=ctx1), co.intt(v, n))
    assert same(pa.evaluate_over_domain(v, D(n), ctx=ctx1), co.ntt(v, n))
    for m, N in ((n, 4 * n), (max(1, n - 3), 2 * n)):
        assert same(pa.evaluate_over_domain(v[:m], D(N), ctx=ctx1), co.ntt(v[:m], N)), (m, N)


def test_known_answers_in_big_integers(ctx1):
    p = F1.P_INT
    w4 = po.root_of_unity(4)
    got = F1.mont_to_ints(pa.evaluate_over_domain(F1.ints_to_mont([1, 2, 3, 4]), D(4), ctx=ctx1))
    assert got == [10, (1 + 2 * w4 + 3 * w4 ** 2 + 4 * w4 ** 3) % p, p - 2, (1 + 2 * w4 ** 3 + 3 * w4 ** 2 + 4 * w4) % p]
    # extreme values: every element p - 1, through a two-pass transform and back
    n = 1 << 13
    x = F1.ints_to_mont([p - 1] * n)
    X = pa.evaluate_over_domain(x, D(n), ctx=ctx1)
    assert F1.mont_to_ints(X[:2]) == [(p - 1) * n % p, 0]
    assert same(pa.evaluations_to_dense_polynomial(X, D(n), ctx=ctx1), x)
    # the largest domain the field allows is 2^34: 2^35 is refused by its own code
    small = np.zeros((4, 4), dtype=np.uint64)
    assert ctx1.lib.plk_intt(ctx1.handle, small.ctypes.data_as(_lib.u64p), 1 << 35, 1 << 35, small.ctypes.data_as(_lib.u64p)) == _lib.PLK_ERR_TWO_ADICITY


@pytest.mark.parametrize("log_n", [2, 6, 10, 12, 16])
def test_each_function_vs_oracle(ctx1, log_n):
    n = 1 << log_n
    c = F1.make_circuit(n, circuit_id=log_n)
    gp = pa.gate_evaluation_table_to_gate_polynomial(c.table, D(n), ctx=ctx1)
    wc, sc = co.gate_polys(c.table.wires(), c.table.selectors, n)
    assert all(same(a, b) for a, b in zip(gp.wires(), wc))
    assert all(same(gp.selectors[k], sc[k]) for k in sc) and set(sc) == set(gp.selectors)
    z = pa.build_grand_product_evaluation_polynomial(c.perm, c.table, c.beta, c.gamma, ctx=ctx1)
    assert same(z, co.grand_product(c.table.wires(), c.perm.columns(), c.beta, c.gamma))
    pp = pa.build_permutation_polynomials(c.perm, c.table, c.beta, c.gamma, D(n), ctx=ctx1)
    so, zc, zl = co.perm_polys(c.table.wires(), c.perm.columns(), c.beta, c.gamma, n)
    assert same(pp.z, zc) and same(pp.z_last, zl) and all(same(a, b) for a, b in zip(pp.sigmas(), so))
    for N in (n, 4 * n):
        assert same(pa.build_gate_constraint_evaluation_polynomial(gp, D(N), ctx=ctx1), co.gate_constraints(wc, sc, N))
        for shift in (1, 4):
            assert same(pa.build_permutation_constraint_evaluation_polynomial(pp, gp, c.beta, c.gamma, D(N), z_shift=shift, ctx=ctx1),
                        co.perm_constraints(zc, so, wc, N, c.beta, c.gamma, z_shift=shift))
    z2, g2, e2 = pa.prove_round23(c.table, c.perm, c.beta, c.gamma, D(n), D(4 * n), ctx=ctx1)
    zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), 4 * n, c.beta, c.gamma)
    assert same(z2, zr) and same(g2, gr) and same(e2, er)


def test_extreme_values_in_the_constraint_kernels(ctx1):
    """Columns of 0, 1, p - 1 and random elements: the lazy limb arithmetic at its bounds over this modulus."""
    p, n = F1.P_INT, 256
    rng = np.random.default_rng(9)
    pool = [0, 1, p - 1, p - 2, 2, (p - 1) // 2]
    cols = [F1.ints_to_mont([pool[i] for i in rng.integers(0, len(pool), n)]) for _ in range(4)]
    c = F1.make_circuit(n, circuit_id=77)
    table = pa.GateEvaluationTable(*cols, selectors=c.table.selectors)
    # (beta = -1, gamma = -3: a denominator term is v - j - 3, never zero for the values of the pool and j < 256)
    for beta, gamma in ((c.beta, c.gamma), (F1.ints_to_mont([p - 1])[0], F1.ints_to_mont([p - 3])[0])):
        z2, g2, e2 = pa.prove_round23(table, c.perm, beta, gamma, D(n), D(4 * n), ctx=ctx1)
        zr, gr, er = co.round23(cols, c.table.selectors, c.perm.columns(), 4 * n, beta, gamma)
        assert same(z2, zr) and same(g2, gr) and same(e2, er)


def test_errors_keep_their_rows(ctx1):
    n = 1 << 9
    c = F1.make_circuit(n, circuit_id=5)
    beta, gamma = F1.mont_to_ints(c.beta)[0], F1.mont_to_ints(c.gamma)[0]
    w = [x.copy() for x in c.table.wires()]
    w[2][17] = F1.ints_to_mont([(-(beta * 17 + gamma)) % F1.P_INT])[0]
    table = pa.GateEvaluationTable(*w, selectors=c.table.selectors)
    with pytest.raises(pa.ZeroDenominator) as ei:
        pa.build_grand_product_evaluation_polynomial(c.perm, table, c.beta, c.gamma, ctx=ctx1)
    assert ei.value.row == 17
    with pytest.raises(pa.PlonkError) as ei:       # beta not reduced modulo THIS field's p (it is below the other field's)
        pa.build_grand_product_evaluation_polynomial(c.perm, c.table, np.array([0, 0, 0, 0x5000000000000000], dtype=np.uint64), c.gamma, ctx=ctx1)
    assert ei.value.code == _lib.PLK_ERR_ARG


def test_round23_2_18_device_resident_full_vectors(prover1):
    n, N = 1 << 18, 1 << 20
    c = F1.make_circuit(n, circuit_id=21)
    dc = prover1.upload(c)
    z, g, e = prover1.round23(dc, N)
    prover1.status()
    co.set_num_threads(min(co.num_threads(), 16))
    zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma)
    assert np.array_equal(to_host(z), zr) and np.array_equal(to_host(g), gr) and np.array_equal(to_host(e), er)


def test_large_transform_closed_form(prover1):
    """2^22 points (three passes): a sparse input's transform in closed form, and a random round trip."""
    N = 1 << 22
    w = po.root_of_unity(N)
    idx, coef = [0, 1, 54321, N - 1], [3, F1.P_INT - 1, 0x1234567890ABCDEF, 2 ** 200 + 9]
    x = torch.zeros((N, 4), dtype=torch.int64, device=prover1.device)
    x[idx] = to_device(F1.ints_to_mont(coef), prover1.device)
    X = prover1.ntt_batch([x], N)[0]
    ks = [0, 1, N - 1, N // 2, 777777]
    assert F1.mont_to_ints(to_host(X[ks])) == [sum(c * pow(w, j * k, F1.P_INT) for c, j in zip(coef, idx)) % F1.P_INT for k in ks]
    assert torch.equal(prover1.ntt_batch([X], N, inverse=True)[0], x)
    y = to_device(F1.random_elements(5, 1 << 20), prover1.device)
    assert torch.equal(prover1.ntt_batch(prover1.ntt_batch([y], 1 << 20, inverse=True), 1 << 20)[0], y)


def test_coset_transforms_and_quotient(ctx1):
    n, N = 1 << 8, 1 << 10
    g = F1.ints_to_mont([5])[0]
    f = F1.random_elements(31, n)
    ev = pa.coset_fft(f, D(N), ctx=ctx1)                      # shift defaults to this field's generator
    assert same(ev, co.coset_ntt(f, N, g))
    assert same(pa.coset_ifft(ev, D(N), ctx=ctx1)[:n], f)
    nums = [F1.random_elements(40 + i, N) for i in range(2)]
    alpha = F1.random_elements(50, 1)[0]
    assert same(pa.quotient_polynomial(nums, alpha, D(n), D(N), ctx=ctx1), co.quotient(nums, alpha, n, g))


def test_tables_before_the_path(ctx1):
    """sigma table (no field arithmetic) and gate-table fill (the field's one) on a trace."""
    n = 1 << 10
    tr = synth.make_trace(n, circuit_id=2)
    vals = F1.random_elements(61, tr.node_values.shape[0])
    dv = F1.ints_to_mont([0])[0]
    table = pa.build_gate_evaluation_table(tr.ops, tr.node_ids, vals, dv, ctx=ctx1)
    wires, sels = co.gate_table_fill(tr.ops, tr.node_ids, vals, dv)
    assert all(same(a, b) for a, b in zip(table.wires(), wires)) and all(same(table.selectors[k], sels[k]) for k in table.selectors) and len(table.selectors) > 0
    perm = pa.build_permutation_table(tr.node_ids, ctx=ctx1)
    assert all(np.array_equal(a, b) for a, b in zip(perm.columns(), co.sigma_table(tr.node_ids)))


def test_commitments_are_refused(ctx1):
    with pytest.raises(pa.PlonkError) as ei:
        pa.msm_g1(F1.random_elements(1, 8), ctx=ctx1)
    assert ei.value.code == _lib.PLK_ERR_ARG and "BLS12-381" in str(ei.value)


def test_two_fields_in_one_process_interleaved(ctx1, ctx):
    """Contexts of both fields used alternately: the instantiations share nothing."""
    n = 1 << 12
    v = F1.random_elements(71, n)            # below both moduli? no: reduce for the first field separately
    v0 = synth.random_elements(71, n)
    for _ in range(2):
        a = pa.evaluate_over_domain(v, D(n), ctx=ctx1)
        with co.use_field(0):
            assert same(pa.evaluate_over_domain(v0, D(n), ctx=ctx), co.ntt(v0, n))
        assert same(a, co.ntt(v, n))


def test_sharded_transform_over_the_second_field(prover1):
    from plonk_amd.node import Node
    N = 1 << 16
    x = to_device(F1.random_elements(81, N), prover1.device)
    direct = prover1.ntt_batch([x], N)[0]
    torch.cuda.synchronize()
    assert np.array_equal(to_host(direct), co.ntt(to_host(x), N))
    node = Node([0, 0, 0, 0], exchange="copy", field=1)
    try:
        m = N // 4
        out = node.ntt([x[g * m:(g + 1) * m].clone() for g in range(4)], N)
        node.sync()
        assert torch.equal(torch.cat(out), direct)
    finally:
        node.close()
