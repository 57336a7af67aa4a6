"""GPU parity of the coset transforms and the quotient step against the C oracle, and the property test at
full size: the gate numerator of a satisfying circuit divides exactly by the vanishing polynomial."""
import numpy as np
import pytest
import torch

import c_oracle as co
import plonk_amd as pa
import plonk_oracle as po
from plonk_amd import _lib, synth
from plonk_amd.device import to_device, to_host

pytestmark = pytest.mark.gpu
D = pa.Radix2EvaluationDomain
G7 = synth.ints_to_mont([po.GENERATOR])[0]


@pytest.mark.parametrize("log_n,log_N", [(0, 0), (0, 3), (3, 3), (5, 7), (10, 12), (11, 11), (12, 14), (16, 18)])
def test_coset_transforms_vs_oracle(ctx, log_n, log_N):
    n_in, N = 1 << log_n, 1 << log_N
    f = synth.random_elements(50 + log_N, n_in)
    for shift in (None, synth.ints_to_mont([pow(3, 1000003, po.P)])[0]):
        sh = G7 if shift is None else shift
        ev = pa.coset_fft(f, D(N), shift=shift, ctx=ctx)
        assert np.array_equal(ev, co.coset_ntt(f, N, sh))
        back = pa.coset_ifft(ev, D(N), shift=shift, ctx=ctx)
        assert np.array_equal(back[:n_in], f) and not back[n_in:].any()
        assert np.array_equal(back, co.coset_intt(ev, sh))


@pytest.mark.parametrize("n,N,count", [(4, 16, 1), (256, 1024, 2), (1024, 1024 * 16, 3), (4096, 4096 * 2, 8)])
def test_quotient_vs_oracle(ctx, n, N, count):
    nums = [synth.random_elements(7 * i + N, N) for i in range(count)]
    alpha = synth.random_elements(99, 1)[0]
    got = pa.quotient_polynomial(nums, alpha, D(n), D(N), ctx=ctx)
    assert np.array_equal(got, co.quotient(nums, alpha, n, G7))


def test_quotient_failures(ctx):
    N = 16
    nums = [synth.random_elements(1, N)]
    one = synth.ints_to_mont([1])[0]
    with pytest.raises(pa.ZeroDenominator):            # the reference's own domain: Z_H vanishes on it
        pa.quotient_polynomial(nums, one, D(4), D(N), shift=one, ctx=ctx)
    with pytest.raises(pa.PlonkError) as ei:           # more than 8 numerators
        pa.quotient_polynomial(nums * 9, one, D(4), D(N), ctx=ctx)
    assert ei.value.code == _lib.PLK_ERR_ARG
    with pytest.raises(pa.PlonkError) as ei:           # N / n above 16
        pa.quotient_polynomial([synth.random_elements(1, 64)], one, D(2), D(64), ctx=ctx)
    assert ei.value.code == _lib.PLK_ERR_ARG
    with pytest.raises(pa.LengthMismatch):             # coset_ifft wants as many evaluations as points
        pa.coset_ifft(nums[0][:8], D(N), ctx=ctx)
    with pytest.raises(pa.PlonkError) as ei:           # zero shift
        pa.coset_fft(nums[0], D(N), shift=np.zeros(4, dtype=np.uint64), ctx=ctx)
    assert ei.value.code == _lib.PLK_ERR_ARG
    got = pa.quotient_polynomial(nums, one, D(4), D(N), ctx=ctx)      # still healthy
    assert np.array_equal(got, co.quotient(nums, one, 4, G7))


@pytest.mark.parametrize("log_n", [10, 18])
def test_gate_quotient_of_a_satisfying_circuit_on_device(prover, log_n):
    """rows are Mult or Int gates with output = a * b: the gate numerator vanishes on the trace domain.  On the
    coset 7 * <omega_4n>: numerator -> quotient t; then t has degree < 3n and t * Z_H is the numerator again."""
    n, N = 1 << log_n, 4 << log_n
    d = prover.device
    c = synth.make_circuit(n, circuit_id=3, kinds=("Int", "Mult"))
    a, b, cc = (to_device(x, d) for x in (c.table.input1, c.table.input2, c.table.input3))
    o = prover.mul(a, b)
    sels = {k: to_device(v, d) for k, v in c.table.selectors.items()}
    cols = [a, b, cc, o, sels["Int"], sels["Mult"]]
    coef = prover.ntt_batch(cols, n, inverse=True)
    on_trace = prover.gate_constraint_eval(prover.ntt_batch(coef[:4], n), {"Int": cols[4], "Mult": cols[5]}, n)
    assert not on_trace.any().item()                                         # the trace satisfies its gates
    ev = prover.coset_ntt(coef, N, G7)
    num = prover.gate_constraint_eval(ev[:4], {"Int": ev[4], "Mult": ev[5]}, N)
    assert num.any().item()
    one = synth.ints_to_mont([1])[0]
    t = prover.quotient([num], one, n, G7)
    torch.cuda.synchronize()
    assert not t[3 * n:].any().item() and t[:3 * n].any().item()             # degree < 3n
    zh = synth.ints_to_mont(po.vanishing_on_coset(n, N))                      # 4 values, period 4
    zh_full = to_device(np.tile(zh, (N // 4, 1)), d)
    back = prover.mul(prover.coset_ntt([t], N, G7)[0], zh_full)
    assert torch.equal(back, num)
    if log_n == 10:                                                          # small enough for the oracle end to end
        assert np.array_equal(to_host(t), co.quotient([to_host(num)], one, n, G7))
