"""GPU tests of the failure behaviour: every reference panic has its error code / exception."""
import ctypes as C

import numpy as np
import pytest

import plonk_amd as pa
import plonk_oracle as po
from plonk_amd import _lib, synth

pytestmark = pytest.mark.gpu
D = pa.Radix2EvaluationDomain


def test_length_assert(ctx):
    v = synth.random_elements(1, 16)
    with pytest.raises(pa.LengthMismatch):           # src/polynomial_utils.rs:22-26
        pa.evaluations_to_dense_polynomial(v[:15], D(16), ctx=ctx)
    with pytest.raises(pa.LengthMismatch):
        pa.evaluations_to_dense_polynomial(v, D(8), ctx=ctx)
    with pytest.raises(pa.LengthMismatch):           # more coefficients than points
        pa.evaluate_over_domain(v, D(8), ctx=ctx)
    c = synth.make_circuit(16)
    with pytest.raises(pa.LengthMismatch):           # table rows != domain size
        pa.gate_evaluation_table_to_gate_polynomial(c.table, D(32), ctx=ctx)
    with pytest.raises(pa.LengthMismatch):
        pa.build_permutation_polynomials(c.perm, c.table, c.beta, c.gamma, D(32), ctx=ctx)


def test_domain_checks(ctx):
    v = synth.random_elements(1, 12)
    with pytest.raises(pa.PlonkError) as ei:
        pa.evaluations_to_dense_polynomial(v, D(12), ctx=ctx)
    assert ei.value.code == _lib.PLK_ERR_NOT_POW2
    with pytest.raises(pa.PlonkError) as ei:
        pa.evaluate_over_domain(v, D(1 << 33), ctx=ctx)
    assert ei.value.code == _lib.PLK_ERR_TWO_ADICITY
    with pytest.raises(pa.PlonkError) as ei:
        pa.evaluate_over_domain(v, D(0), ctx=ctx)
    assert ei.value.code == _lib.PLK_ERR_NOT_POW2


def test_zero_denominator_is_reported_with_its_row(ctx):
    n = 4096
    c = synth.make_circuit(n)
    b, g = synth.mont_to_ints(c.beta)[0], synth.mont_to_ints(c.gamma)[0]
    for row, col in ((0, 0), (1234, 2), (n - 1, 3)):
        wires = [w.copy() for w in c.table.wires()]
        wires[col][row] = synth.ints_to_mont([(-(b * row + g)) % po.P])[0]
        t = pa.GateEvaluationTable(*wires, selectors=c.table.selectors)
        with pytest.raises(pa.ZeroDenominator) as ei:   # `.inverse().unwrap()`, src/permutation_polynomial.rs:258
            pa.build_grand_product_evaluation_polynomial(c.perm, t, c.beta, c.gamma, ctx=ctx)
        assert ei.value.row == row
    # the context stays usable and the next call is clean
    z = pa.build_grand_product_evaluation_polynomial(c.perm, c.table, c.beta, c.gamma, ctx=ctx)
    assert synth.mont_to_ints(z[:1]) == [1]


def test_sigma_out_of_range(ctx):
    n = 2048
    c = synth.make_circuit(n)
    sig = [s.copy() for s in c.perm.columns()]
    sig[1][77] = n            # first bad row
    sig[3][900] = 2**63
    p = pa.PermutationTable(*sig)
    with pytest.raises(pa.SigmaOutOfRange) as ei:       # slice index panic, src/permutation_polynomial.rs:235-238
        pa.build_grand_product_evaluation_polynomial(p, c.table, c.beta, c.gamma, ctx=ctx)
    assert ei.value.row == 77
    with pytest.raises(pa.SigmaOutOfRange):
        pa.prove_round23(c.table, p, c.beta, c.gamma, D(n), D(4 * n), ctx=ctx)


def test_bad_arguments(ctx):
    lib = ctx.lib
    out = (C.c_uint64 * 4)()
    assert lib.plk_intt(ctx.handle, None, 1, 1, out) == _lib.PLK_ERR_ARG
    assert lib.plk_intt(None, out, 1, 1, out) == _lib.PLK_ERR_ARG
    c = synth.make_circuit(8)
    bad_beta = np.array([2**64 - 1] * 4, dtype=np.uint64)  # not reduced mod p
    with pytest.raises(pa.PlonkError) as ei:
        pa.build_grand_product_evaluation_polynomial(c.perm, c.table, bad_beta, c.gamma, ctx=ctx)
    assert ei.value.code == _lib.PLK_ERR_ARG
    with pytest.raises(KeyError):
        pa.gate_evaluation_table_to_gate_polynomial(
            pa.GateEvaluationTable(*c.table.wires(), selectors={"Xor": c.table.input1}), D(8), ctx=ctx)


def test_empty_and_single_row(ctx):
    # n = 0: z = [1] (the loop of src/permutation_polynomial.rs:227 does not run)
    empty = np.zeros((0, 4), dtype=np.uint64)
    t0 = pa.GateEvaluationTable(empty, empty, empty, empty)
    p0 = pa.PermutationTable(*[np.zeros((0,), dtype=np.uint64)] * 4)
    one = synth.ONE_MONT
    z = pa.build_grand_product_evaluation_polynomial(p0, t0, one, one, ctx=ctx)
    assert z.shape == (1, 4) and synth.mont_to_ints(z) == [1]
    # n = 1
    c = synth.make_circuit(1)
    import c_oracle as co
    z1 = pa.build_grand_product_evaluation_polynomial(c.perm, c.table, c.beta, c.gamma, ctx=ctx)
    assert np.array_equal(z1, co.grand_product(c.table.wires(), c.perm.columns(), c.beta, c.gamma))
    # zero coefficients evaluate to zero
    assert not pa.evaluate_over_domain(empty, D(8), ctx=ctx).any()


def test_bad_arguments_of_the_newer_entry_points(ctx, prover):
    """null tables, ranges outside the circuit, a finish that does not match its local call, chunk layouts
    that do not tile the domain: all PLK_ERR_ARG, and the context keeps working afterwards."""
    import torch
    from plonk_amd.device import to_device
    lib, h = ctx.lib, ctx.handle
    assert lib.plk_sigma_table(h, None, 4, None) == _lib.PLK_ERR_ARG
    assert lib.plk_gate_table_fill(h, None, None, 4, None, 0, None, None, None) == _lib.PLK_ERR_ARG
    n = 64
    c = synth.make_circuit(n)
    wires = [to_device(w, prover.device) for w in c.table.wires()]
    sigma = [to_device(s, prover.device) for s in c.perm.columns()]
    with pytest.raises(pa.PlonkError) as ei:                       # rows reach past the circuit
        prover.grand_product_local(wires, sigma, n, 60, 8, c.beta, c.gamma)
    assert ei.value.code == _lib.PLK_ERR_ARG
    totals = prover.grand_product_local(wires, sigma, n, 0, 32, c.beta, c.gamma)
    with pytest.raises(pa.PlonkError) as ei:                       # finish with other rows than the local call
        prover.grand_product_finish(16, torch.stack([totals, totals]), 0)
    assert ei.value.code == _lib.PLK_ERR_ARG
    with pytest.raises(pa.PlonkError) as ei:                       # `mine` outside the range table
        prover.grand_product_finish(32, torch.stack([totals, totals]), 2)
    assert ei.value.code == _lib.PLK_ERR_ARG
    with pytest.raises(pa.PlonkError) as ei:                       # 3 parts do not divide 4n points
        prover.extend4_chunked(wires[:1], n, [_lib.INPUT_FIELD], 3, 1, 1)
    assert ei.value.code == _lib.PLK_ERR_ARG
    with pytest.raises(pa.PlonkError) as ei:                       # extension needs n >= 4
        prover.extend4([wires[0][:2].contiguous()], 2)
    assert ei.value.code == _lib.PLK_ERR_TWO_ADICITY
    # a non-reduced default value is rejected
    t = synth.make_trace(8)
    bad = np.full((4,), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    with pytest.raises(pa.PlonkError) as ei:
        pa.build_gate_evaluation_table(t.ops, t.node_ids, t.node_values, bad, ctx=ctx)
    assert ei.value.code == _lib.PLK_ERR_ARG
    # still healthy
    z = prover.grand_product_cols(wires, sigma, n, c.beta, c.gamma)
    prover.status()
    import c_oracle as co
    from plonk_amd.device import to_host
    assert np.array_equal(to_host(z), co.grand_product(c.table.wires(), c.perm.columns(), c.beta, c.gamma))


def test_sigma_table_random_shapes(ctx):
    """many small random shapes: row counts around the chunk and scan-tile sizes, id ranges from 1 to 2^64."""
    import c_oracle as co
    rng = np.random.default_rng(2024)
    for trial in range(40):
        n = int(rng.integers(1, 3000))
        bits = int(rng.integers(1, 65))
        pool = max(1, int(rng.integers(1, 4 * n)))
        ids_pool = rng.integers(0, 2 ** 63, size=pool, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=pool, dtype=np.uint64)
        ids_pool >>= np.uint64(64 - bits)
        ids_pool[ids_pool == np.uint64(po.NO_WIRE)] = np.uint64(0)
        cols = [ids_pool[rng.integers(0, pool, size=n)] for _ in range(4)]
        for w in range(3):
            cols[w][rng.random(n) < 0.3] = np.uint64(po.NO_WIRE)
        got = pa.build_permutation_table(cols, ctx=ctx)
        want = co.sigma_table(cols)
        assert all(np.array_equal(g, w) for g, w in zip(got.columns(), want)), (trial, n, bits, pool)
