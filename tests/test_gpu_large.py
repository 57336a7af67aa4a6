"""GPU tests at the benchmark's full sizes (n = 2^20, N = 2^22) through the device entry points:
size-independent properties plus spot checks against big-integer arithmetic."""
import numpy as np
import pytest
import torch

import c_oracle as co
import plonk_oracle as po
from helpers import horner
from plonk_amd import synth
from plonk_amd.device import to_device, to_host

pytestmark = pytest.mark.gpu

LOG_N = 20
N_ROWS, N_EVAL = 1 << LOG_N, 1 << (LOG_N + 2)


@pytest.fixture(scope="module")
def big(prover):
    c = synth.make_circuit(N_ROWS)
    return c, prover.upload(c)


def test_roundtrip_and_spot_values_2_20(prover, big):
    c, dc = big
    coef = prover.ntt_batch(dc.wires[:2], N_ROWS, inverse=True)
    back = prover.ntt_batch(coef, N_ROWS)
    torch.cuda.synchronize()
    assert torch.equal(back[0], dc.wires[0]) and torch.equal(back[1], dc.wires[1])   # NTT(iNTT(x)) = x
    ev = prover.ntt_batch(coef[:1], N_EVAL)[0]
    # every 4th point of the 4n domain is a point of the n domain: the extension restricted to it is the input
    assert torch.equal(ev[::4], dc.wires[0])
    # evaluations at a few points of the 4n domain by Horner in big integers (a 2^20-term sum each)
    coef_ints = synth.mont_to_ints(to_host(coef[0]))
    w = po.root_of_unity(N_EVAL)
    ev_h = to_host(ev)
    for i in (1, 2**21 + 3, N_EVAL - 1):
        assert synth.mont_to_ints(ev_h[i:i + 1])[0] == horner(coef_ints, pow(w, i, po.P))


def test_grand_product_2_20_vs_serial_recurrence(prover, big):
    c, dc = big
    z = prover.grand_product(dc)
    prover.status()
    z_ref = co.grand_product(c.table.wires(), c.perm.columns(), c.beta, c.gamma)   # the serial loop, one inversion per row
    assert np.array_equal(to_host(z), z_ref)


def test_round23_2_20(prover, big):
    """The n = 2^20, N = 2^22 configuration.  Checks: all three outputs IN FULL against the C restatement of the
    reference (Z: the serial recurrence of src/permutation_polynomial.rs:217-261; the gate and permutation vectors:
    the loops of src/execution_trace.rs:357-431 and src/permutation_polynomial.rs:312-376 over all 2^22 points); the
    two constraint vectors at random points recomputed in big integers from the evaluated columns' definition (Horner
    on the coefficient vectors); the restriction of the gate-constraint vector to the trace domain against the direct
    row-wise formula."""
    c, dc = big
    z, g, e = prover.round23(dc, N_EVAL)
    prover.status()
    z_h = to_host(z)
    co.set_num_threads(min(co.num_threads(), 16))
    z_ref, g_ref, e_ref = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N_EVAL, c.beta, c.gamma)
    assert np.array_equal(z_h, z_ref)
    assert np.array_equal(to_host(g), g_ref)
    assert np.array_equal(to_host(e), e_ref)
    del z_ref, g_ref, e_ref

    # coefficient vectors through the library's own iNTT (parity of the iNTT itself is covered above and in test_gpu_parity)
    cols = dc.wires + [dc.selectors[k] for k in po.AST_KINDS] + [z[:N_ROWS].contiguous()]
    coefs = prover.ntt_batch(cols, N_ROWS, inverse=True)
    sig_coefs = prover.ntt_batch(dc.sigma, N_ROWS, inverse=True, kinds=[1] * 4)
    torch.cuda.synchronize()
    rng = np.random.default_rng(5)
    pts = [0, 1, N_EVAL - 1] + [int(x) for x in rng.integers(0, N_EVAL, 5)]
    beta, gamma = synth.mont_to_ints(c.beta)[0], synth.mont_to_ints(c.gamma)[0]
    w = po.root_of_unity(N_EVAL)
    coef_ints = [synth.mont_to_ints(to_host(t)) for t in coefs]
    sig_ints = [synth.mont_to_ints(to_host(t)) for t in sig_coefs]
    g_h, e_h = to_host(g), to_host(e)
    for i in pts:
        x, xn = pow(w, i, po.P), pow(w, (i + 1) % N_EVAL, po.P)
        a, b, cc, o = (horner(coef_ints[k], x) for k in range(4))
        sel = {k: horner(coef_ints[4 + j], x) for j, k in enumerate(po.AST_KINDS)}
        want_g = sum(sel[k] * po.gate_expression(k, a, b, cc, o) for k in po.AST_KINDS) % po.P
        assert synth.mont_to_ints(g_h[i:i + 1])[0] == want_g, i
        zc, zn = horner(coef_ints[13], x), horner(coef_ints[13], xn)
        lhs = rhs = 1
        for k, wv in enumerate((a, b, cc, o)):
            lhs = lhs * (wv + beta * i + gamma) % po.P
            rhs = rhs * (wv + beta * horner(sig_ints[k], x) + gamma) % po.P
        assert synth.mont_to_ints(e_h[i:i + 1])[0] == (zn * lhs - zc * rhs) % po.P, i

    # on the trace domain (every 4th point) the gate vector equals the row-wise expression on the raw table
    rows = [0, 1, 77777, N_ROWS - 1]
    wires_i = [synth.mont_to_ints(w_[rows]) for w_ in c.table.wires()]
    for r_i, r in enumerate(rows):
        a, b, cc, o = (wires_i[k][r_i] for k in range(4))
        kind = next(k for k in po.AST_KINDS if c.table.selectors[k][r].any())
        assert synth.mont_to_ints(g_h[4 * r:4 * r + 1])[0] == po.gate_expression(kind, a, b, cc, o)


def test_batch_of_2_18_circuits_through_one_context(prover):
    """The shape of the proof-throughput mode (64 circuits of n = 2^18 dealt to 8 GPUs = 8 per GPU): eight DIFFERENT
    circuits back to back through one context and one shared set of output buffers, nothing synchronised between
    them but the copy of the results; every output compared in full with the C restatement."""
    n, N = 1 << 18, 1 << 20
    circuits = [synth.make_circuit(n, circuit_id=100 + i) for i in range(8)]
    dcs = [prover.upload(c) for c in circuits]
    out = (prover.empty(n + 1), prover.empty(N), prover.empty(N))
    got = []
    for dc in dcs:
        prover.round23(dc, N, out=out)
        got.append([t.clone() for t in out])          # stream-ordered copies; the next call overwrites `out`
    prover.status()
    co.set_num_threads(min(co.num_threads(), 16))
    for c, (z, g, e) in zip(circuits, got):
        zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma)
        assert np.array_equal(to_host(z), zr) and np.array_equal(to_host(g), gr) and np.array_equal(to_host(e), er)


def test_device_constraint_kernels_match_host_entry_points(prover):
    """plk_dev_* element-wise entry points on columns that already sit in HBM."""
    n, N = 1 << 12, 1 << 14
    c = synth.make_circuit(n, circuit_id=3)
    dc = prover.upload(c)
    z = prover.grand_product(dc)
    coefs = prover.ntt_batch(dc.wires + [dc.selectors[k] for k in po.AST_KINDS] + [z[:n].contiguous()], n, inverse=True)
    sig_coefs = prover.ntt_batch(dc.sigma, n, inverse=True, kinds=[1] * 4)
    evals = prover.ntt_batch(coefs + sig_coefs, N)
    w_ev, s_ev, z_ev, g_ev = evals[:4], dict(zip(po.AST_KINDS, evals[4:13])), evals[13], evals[14:18]
    g = prover.gate_constraint_eval(w_ev, s_ev, N)
    e = prover.perm_constraint_eval(z_ev, g_ev, w_ev, N, c.beta, c.gamma)
    prover.status()
    zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma)
    assert np.array_equal(to_host(g), gr) and np.array_equal(to_host(e), er) and np.array_equal(to_host(z), zr)


@pytest.mark.parametrize("log_n", [22, 26])
def test_big_transforms_closed_form(prover, log_n):
    """Sizes past the oracle's reach (2^26 also takes the two-level twiddle path of the first pass):
    a sparse input has the closed-form transform X[k] = sum_i c_i w^(j_i k); random data round-trips."""
    N = 1 << log_n
    w = po.root_of_unity(N)
    idx = [0, 1, 12345, N // 2 + 3, N - 1]
    coef = [3, po.P - 1, 0x1234567890ABCDEF, 7, 2**200 + 9]
    x = torch.zeros((N, 4), dtype=torch.int64, device=prover.device)
    x[idx] = to_device(synth.ints_to_mont(coef), prover.device)
    X = prover.ntt_batch([x], N)[0]
    rng = np.random.default_rng(log_n)
    ks = [0, 1, N - 1, N // 2] + [int(k) for k in rng.integers(0, N, size=24)]
    got = synth.mont_to_ints(to_host(X[ks]))
    for k, g in zip(ks, got):
        assert g == sum(c * pow(w, j * k, po.P) for c, j in zip(coef, idx)) % po.P, k
    back = prover.ntt_batch([X], N, inverse=True)[0]
    assert torch.equal(back, x)
    del X, back
    # random residues below p (top limb below the modulus' top limb)
    y = torch.randint(-2**63, 2**63 - 1, (N, 4), dtype=torch.int64, device=prover.device)
    y[:, 3] = torch.randint(0, 0x7000000000000000, (N,), dtype=torch.int64, device=prover.device)
    Y = prover.ntt_batch([y], N, inverse=True)[0]
    assert torch.equal(prover.ntt_batch([Y], N)[0], y)
    # linearity ties the random vector to the closed form: NTT(iNTT(y) + x) = y + NTT(x) at the sampled points
    s = prover.ntt_batch([field_add(prover, Y, x)], N)[0]
    want = [(a + b) % po.P for a, b in zip(synth.mont_to_ints(to_host(y[ks])), got)]
    assert synth.mont_to_ints(to_host(s[ks])) == want


def field_add(prover, a, b):
    """a + b for a sparse b: the sum is formed on the host for the few non-zero rows of b."""
    out = a.clone()
    nz = torch.nonzero(b.any(dim=1)).flatten()
    if nz.numel():
        av = synth.mont_to_ints(to_host(a[nz]))
        bv = synth.mont_to_ints(to_host(b[nz]))
        out[nz] = to_device(synth.ints_to_mont([(p + q) % po.P for p, q in zip(av, bv)]), prover.device)
    return out
