"""GPU tests of how calls on one context are ordered across streams, and of the page-locked host path."""
import numpy as np
import pytest
import torch

import c_oracle as co
import plonk_amd as pa
from plonk_amd import _lib, synth
from plonk_amd.device import to_host

pytestmark = pytest.mark.gpu
D = pa.Radix2EvaluationDomain


def test_calls_on_two_streams_share_the_workspace_safely(prover):
    """Two rounds 2+3 enqueued back to back on two different streams without any synchronisation between them:
    the context's workspace is shared, so the library must order the second call behind the first."""
    n, N = 1 << 14, 1 << 16
    ca, cb = synth.make_circuit(n, circuit_id=21), synth.make_circuit(n, circuit_id=22)
    da, db = prover.upload(ca), prover.upload(cb)
    ref_a = [t.clone() for t in prover.round23(da, N)]
    ref_b = [t.clone() for t in prover.round23(db, N)]
    prover.status()
    out_a = (prover.empty(n + 1), prover.empty(N), prover.empty(N))
    out_b = (prover.empty(n + 1), prover.empty(N), prover.empty(N))
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(prover.device), torch.cuda.Stream(prover.device)
    for _ in range(3):
        with torch.cuda.stream(s1):
            prover.round23(da, N, out=out_a)
        with torch.cuda.stream(s2):
            prover.round23(db, N, out=out_b)
            z_b = prover.grand_product(db)          # a third call, same stream
        with torch.cuda.stream(s1):
            z_a = prover.grand_product(da)          # back on the first stream
    torch.cuda.synchronize()
    prover.status()
    assert all(torch.equal(x, y) for x, y in zip(out_a, ref_a))
    assert all(torch.equal(x, y) for x, y in zip(out_b, ref_b))
    assert torch.equal(z_a, ref_a[0]) and torch.equal(z_b, ref_b[0])
    assert np.array_equal(to_host(z_a), co.grand_product(ca.table.wires(), ca.perm.columns(), ca.beta, ca.gamma))


def test_round23_on_page_locked_arrays_equals_pageable(ctx):
    """plk_round23 through plk_host_alloc memory (DMA) and through ordinary numpy arrays (threaded staging)."""
    n = 1 << 16          # 2 MiB per column x 13 + sigma: beyond the 8 MiB threshold of the threaded copy in total per call
    c = synth.make_circuit(n, circuit_id=23)
    z, g, p = pa.prove_round23(c.table, c.perm, c.beta, c.gamma, D(n), D(4 * n), ctx=ctx)

    def pin(a):
        b = ctx.pinned_empty(a.shape, a.dtype)
        b[...] = a
        return b

    pt = pa.GateEvaluationTable(*[pin(w) for w in c.table.wires()], selectors={k: pin(v) for k, v in c.table.selectors.items()})
    pp = pa.PermutationTable(*[pin(s) for s in c.perm.columns()])
    outs = (ctx.pinned_empty((n + 1, 4)), ctx.pinned_empty((4 * n, 4)), ctx.pinned_empty((4 * n, 4)))
    z2, g2, p2 = pa.prove_round23(pt, pp, c.beta, c.gamma, D(n), D(4 * n), ctx=ctx, out=outs)
    assert np.array_equal(z2, z) and np.array_equal(g2, g) and np.array_equal(p2, p)
    zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), 4 * n, c.beta, c.gamma)
    assert np.array_equal(z2, zr) and np.array_equal(g2, gr) and np.array_equal(p2, er)


def test_pinned_buffer_may_outlive_its_context():
    """A page-locked array released after plk_ctx_destroy (the advisor's use-after-free scenario)."""
    import gc
    c2 = pa.Context(0)
    buf = c2.pinned_empty((1024, 4))
    buf[...] = 7
    c2.close()
    assert int(buf[5, 1]) == 7
    del buf
    gc.collect()


def test_build_and_convert_rejects_a_bogus_domain_before_sizing_anything(ctx):
    def builder():
        raise AssertionError("the builder must not run for an invalid domain")
    with pytest.raises(pa.PlonkError) as ei:
        pa.build_and_convert(D(12), builder, ctx=ctx)
    assert ei.value.code == _lib.PLK_ERR_NOT_POW2
    # sizes no host could hold, straight through the C ABI (the Python mirror would size its result first)
    called = []
    fn = _lib.BUILDER_FN(lambda user, dst, cap: called.append(cap) or 0)
    small = np.zeros((4, 4), dtype=np.uint64)
    for size, code in ((1 << 40, _lib.PLK_ERR_TWO_ADICITY), ((1 << 62) + 1, _lib.PLK_ERR_NOT_POW2)):
        rc = ctx.lib.plk_build_and_convert(ctx.handle, size, fn, None, small.ctypes.data_as(_lib.u64p))
        assert rc == code and not called


def test_a_stream_may_be_destroyed_between_calls(prover):
    """A caller with one stream per request: create a stream, enqueue a call, wait for it, destroy the stream - and the
    next call, on another stream, must not touch the dead handle (the library records its ordering event on a call's
    own stream at the end of that call, never on the previous call's stream)."""
    import ctypes as C
    from plonk_amd.device import _table, _ptr
    hip = C.CDLL("libamdhip64.so")        # already mapped by torch / the library: the same runtime
    n = 1 << 12
    c = synth.make_circuit(n, circuit_id=24)
    dc = prover.upload(c)
    want = prover.grand_product(dc)
    prover.status()
    torch.cuda.synchronize()
    b, g = dc.beta.ctypes.data_as(_lib.u64p), dc.gamma.ctypes.data_as(_lib.u64p)
    for _ in range(3):
        st = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(st)) == 0
        z = prover.empty(n + 1)
        torch.cuda.synchronize()            # (the tensor's allocation is not stream-ordered against a raw stream)
        rc = prover.lib.plk_dev_grand_product(prover.ctx.handle, st, _table(dc.wires), _table(dc.sigma), n, b, g, _ptr(z))
        prover.ctx.check(rc)
        assert hip.hipStreamSynchronize(st) == 0
        assert hip.hipStreamDestroy(st) == 0
        assert torch.equal(z, want)
        # the next call comes on torch's stream: the destroyed handle must not be used again
        z2 = prover.grand_product(dc)
        prover.status()
        assert torch.equal(z2, want)
