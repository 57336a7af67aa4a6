"""plk_dev_grand_product_noalloc (the entry point that allocates nothing: its per-row terms and block products
live in a fixed array inside the library's code object) against plk_dev_grand_product and the serial CPU loop.

Bar: bit-exact (prime-field integers).  Two calls in a row share the array, the second on a smaller circuit, so
stale contents of the first call would show; sizes beyond the array are refused."""
import numpy as np
import pytest
import torch

import c_oracle as co
from plonk_amd import _lib, synth
import plonk_amd as pa
import plonk_oracle as po
from plonk_amd.device import to_host

pytestmark = pytest.mark.gpu


def _both(prover, n, circuit_id):
    c = synth.make_circuit(n, circuit_id=circuit_id)
    dc = prover.upload(c)
    want = prover.grand_product(dc)
    prover.status()
    got = prover.empty(n + 1)
    got.fill_(-1)                                   # poison: every element must be written by the call
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    prover.grand_product_noalloc(dc, got)
    prover.status()
    free_after = torch.cuda.mem_get_info()[0]
    assert free_after >= free_before                # nothing was taken from the device heap
    assert torch.equal(got, want)
    z_ref = co.grand_product(c.table.wires(), c.perm.columns(), c.beta, c.gamma)
    assert np.array_equal(to_host(got), z_ref)


def test_noalloc_two_calls_second_smaller(prover):
    _both(prover, _lib.NOALLOC_MAX_ROWS, circuit_id=11)      # the largest size the array holds
    _both(prover, 1 << 10, circuit_id=12)                    # then a smaller one over the same array
    _both(prover, 1000, circuit_id=13)                       # ragged: not a multiple of the block size
    _both(prover, 1, circuit_id=14)


def test_noalloc_rejects_sizes_beyond_the_array(prover):
    n = 2 * _lib.NOALLOC_MAX_ROWS
    c = synth.make_circuit(n, circuit_id=15)
    dc = prover.upload(c)
    out = prover.empty(n + 1)
    with pytest.raises(pa.PlonkError) as ei:
        prover.grand_product_noalloc(dc, out)
    assert ei.value.code == _lib.PLK_ERR_ARG and "at most" in str(ei.value)
    # the refusal leaves the context usable
    _both(prover, 1 << 8, circuit_id=16)


def test_noalloc_reports_zero_denominator(prover):
    """Data-dependent failures come through plk_dev_status as with plk_dev_grand_product."""
    n = 1 << 9
    c = synth.make_circuit(n, circuit_id=17)
    dc = prover.upload(c)
    # row 5: v_1[5] = -(beta * 5 + gamma) makes the first denominator factor zero
    beta, gamma = synth.mont_to_ints(c.beta)[0], synth.mont_to_ints(c.gamma)[0]
    bad = (-(beta * 5 + gamma)) % po.P
    dc.wires[0][5] = torch.from_numpy(synth.ints_to_mont([bad]).view(np.int64)[0].copy()).to(prover.device)
    out = prover.empty(n + 1)
    prover.grand_product_noalloc(dc, out)
    with pytest.raises(pa.ZeroDenominator) as ei:
        prover.status()
    assert ei.value.row == 5


def test_noalloc_under_graph_capture(prover):
    """The call recorded into a HIP graph and replayed: nothing in it allocates, synchronises or reads host memory at
    replay time, and a replay on changed column contents gives that data's grand product."""
    n = 1 << 14
    c = synth.make_circuit(n, circuit_id=18)
    dc = prover.upload(c)
    got = prover.empty(n + 1)
    got.fill_(-1)
    prover.grand_product_noalloc(dc, got)              # once outside the capture (kernels loaded, result known)
    prover.status()
    want = got.clone()
    got.fill_(-1)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        prover.grand_product_noalloc(dc, got)
    torch.cuda.synchronize()
    assert int(got[0, 0]) == -1                        # captured, not run
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # other data in the same buffers, same graph
    c2 = synth.make_circuit(n, circuit_id=19)
    for dst, src in zip(dc.wires, c2.table.wires()):
        dst.copy_(torch.from_numpy(src.view(np.int64)))
    for dst, src in zip(dc.sigma, c2.perm.columns()):
        dst.copy_(torch.from_numpy(src.view(np.int64)))
    graph.replay()
    torch.cuda.synchronize()
    z_ref = co.grand_product(c2.table.wires(), c2.perm.columns(), c.beta, c.gamma)   # (beta, gamma are baked into the graph)
    assert np.array_equal(to_host(got), z_ref)
    # the context is in order after the capture: an ordinary call on the ordinary stream
    _both(prover, 1 << 9, circuit_id=20)
