"""GPU tests of rounds 2+3 for small circuits (one block of the grand product up to many, one tile of the transforms
up to several), called repeatedly on the same buffers with changing challenges, across streams, and with failures."""
import numpy as np
import pytest
import torch

import c_oracle as co
import plonk_amd as pa
from plonk_amd import synth
from plonk_amd.device import to_host, _host_fe

pytestmark = pytest.mark.gpu


def _oracle(c, N, beta, gamma, z_shift=1):
    return co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, beta, gamma, z_shift=z_shift)


def _same(outs, refs):
    return all(np.array_equal(to_host(o), r) for o, r in zip(outs, refs))


@pytest.mark.parametrize("log_n", [2, 5, 8, 9, 12, 15, 16])
def test_small_sizes_with_fresh_challenges(prover, log_n):
    """One row per thread in the grand product up to 2^15 rows, four beyond; fresh challenges on every call."""
    n, N = 1 << log_n, 4 << log_n
    c = synth.make_circuit(n, circuit_id=40 + log_n)
    dc = prover.upload(c)
    out = (prover.empty(n + 1), prover.empty(N), prover.empty(N))
    for i in range(3):
        other = synth.make_circuit(4, circuit_id=900 + 7 * log_n + i)      # a source of fresh challenges
        dc.beta, dc.gamma = _host_fe(other.beta), _host_fe(other.gamma)
        for t in out:
            t.zero_()
        prover.round23(dc, N, out=out)
        prover.status()
        assert _same(out, _oracle(c, N, other.beta, other.gamma)), f"call {i}"


def test_on_two_streams_and_between_other_calls(prover):
    """Small calls on a side stream interleaved with a larger call and a stand-alone grand product on the default one:
    the calls share the context's workspace and side stream, the library must keep them apart."""
    n, N = 1 << 10, 1 << 12
    c, c2 = synth.make_circuit(n, circuit_id=61), synth.make_circuit(32 * n, circuit_id=62)
    dc, dc2 = prover.upload(c), prover.upload(c2)
    ref = _oracle(c, N, c.beta, c.gamma)
    ref2 = _oracle(c2, 32 * N, c2.beta, c2.gamma)
    out = (prover.empty(n + 1), prover.empty(N), prover.empty(N))
    out2 = (prover.empty(32 * n + 1), prover.empty(32 * N), prover.empty(32 * N))
    side = torch.cuda.Stream(prover.device)
    for _ in range(3):
        for t in out + out2:
            t.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            prover.round23(dc, N, out=out)
        prover.round23(dc2, 32 * N, out=out2)
        z = prover.grand_product(dc)
        with torch.cuda.stream(side):
            prover.round23(dc, N, out=out)
        torch.cuda.synchronize()
        prover.status()
        assert _same(out, ref) and _same(out2, ref2)
        assert np.array_equal(to_host(z), ref[0])


def test_without_z_and_gate_outputs_and_with_a_wider_shift(prover):
    """z stays in the workspace (no z_out), no gate vector, z_shift = 4."""
    n, N = 1 << 8, 1 << 10
    c = synth.make_circuit(n, circuit_id=65)
    dc = prover.upload(c)
    ref = _oracle(c, N, c.beta, c.gamma, z_shift=4)
    perm = prover.empty(N)
    wires, sels, sigma = dc.tables()
    for i in range(2):
        perm.zero_()
        prover.ctx.check(prover.lib.plk_dev_round23(prover.ctx.handle, prover._stream(), wires, sels, sigma, n, N, dc.beta.ctypes.data,
                                                    dc.gamma.ctypes.data, 4, None, None, perm.data_ptr()))
        prover.status()
        assert np.array_equal(to_host(perm), ref[2]), f"call {i}"


@pytest.mark.parametrize("log_n,row", [(6, 11), (10, 1000), (12, 0)])
def test_failures_are_reported_with_their_row(prover, log_n, row):
    """A zero denominator planted by choosing gamma = -(w + beta * j) for one row, and an index out of range."""
    n, N = 1 << log_n, 4 << log_n
    c = synth.make_circuit(n, circuit_id=66 + log_n)
    dc = prover.upload(c)
    out = (prover.empty(n + 1), prover.empty(N), prover.empty(N))
    prover.round23(dc, N, out=out)
    prover.status()
    w = synth.mont_to_ints(c.table.wires()[0][row:row + 1])[0]
    beta = synth.mont_to_ints(c.beta)[0]
    good_gamma = dc.gamma
    dc.gamma = _host_fe(synth.ints_to_mont([-(w + beta * row)])[0])
    prover.round23(dc, N, out=out)
    with pytest.raises(pa.ZeroDenominator) as ei:
        prover.status()
    assert ei.value.row == row
    dc.gamma = good_gamma
    prover.round23(dc, N, out=out)          # the context stays usable
    prover.status()
    assert _same(out, _oracle(c, N, c.beta, c.gamma))
    sig = dc.sigma[2].clone()
    dc.sigma[2][row] = n + 5
    prover.round23(dc, N, out=out)
    with pytest.raises(pa.PlonkError) as ei:
        prover.status()
    assert ei.value.row == row
    dc.sigma[2].copy_(sig)
    prover.round23(dc, N, out=out)
    prover.status()
    assert _same(out, _oracle(c, N, c.beta, c.gamma))


def test_randomised_sweep(prover):
    """Thirty random cases of the device-resident call: sizes 2^2 .. 2^13, any subset of the selector kinds, the
    4n-point domain mostly (the coset-major kernels) and n, 2n, 8n now and then (the natural-order ones), any z_shift."""
    from plonk_amd.api import AST_KINDS
    rng = np.random.default_rng(2024)
    for case in range(30):
        log_n = int(rng.integers(2, 14))
        n = 1 << log_n
        N = n * (4 if rng.random() < 0.75 else int(rng.choice([1, 2, 8])))
        kinds = tuple(k for k in AST_KINDS if rng.random() < 0.6)
        z_shift = int(rng.choice([1, 1, 4, 3, N - 1, int(rng.integers(0, N))]))
        c = synth.make_circuit(n, circuit_id=int(rng.integers(0, 1 << 20)), kinds=kinds)
        outs = prover.round23(prover.upload(c), N, z_shift=z_shift)
        prover.status()
        assert _same(outs, _oracle(c, N, c.beta, c.gamma, z_shift=z_shift)), (case, log_n, N // n, kinds, z_shift)


def test_rebinding_a_column_is_seen_by_round23(prover):
    """The driver caches its pointer tables per uploaded circuit; a column bound to another tensor afterwards must reach
    every entry point, the cached ones included."""
    import torch
    from plonk_amd import synth
    n, N = 1 << 8, 1 << 10
    a, b = synth.make_circuit(n, circuit_id=31), synth.make_circuit(n, circuit_id=32)
    da, db = prover.upload(a), prover.upload(b)
    za = prover.round23(da, N)
    want = [t.clone() for t in prover.round23(db, N)]
    # turn `da` into circuit b by rebinding (not overwriting) its columns
    da.wires[:] = db.wires
    da.sigma[:] = db.sigma
    for k in list(da.selectors):
        da.selectors[k] = db.selectors[k]
    da.beta, da.gamma = db.beta, db.gamma
    got = prover.round23(da, N)
    prover.status()
    assert all(torch.equal(x, y) for x, y in zip(got, want))
    assert not torch.equal(za[0], want[0])
