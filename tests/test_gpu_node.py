"""GPU tests of the native node layer (plk_node_*: G ranks in one process behind the C ABI).  On this one-GPU box
the G ranks of the copy exchange share GPU 0 - the index bookkeeping, the event ordering between the ranks' streams
and the host threads are the ones a node of G GPUs runs - and RCCL runs in a one-rank node."""
import ctypes as C

import numpy as np
import pytest
import torch

import c_oracle as co
import plonk_amd as pa
from plonk_amd import _lib, synth
from plonk_amd.device import to_device, to_host
from plonk_amd.node import Node

pytestmark = pytest.mark.gpu


def _shards(x: torch.Tensor, G: int):
    m = x.shape[0] // G
    return [x[g * m:(g + 1) * m].clone() for g in range(G)]


@pytest.mark.parametrize("G", [1, 2, 4, 8])
def test_node_ntt_copy_exchange_matches_direct_transform(prover, G):
    """The sharded transform at 2^20 points over G ranks against the library's direct transform (itself checked
    against the oracle), both directions, both output layouts, repeated calls reusing the node's buffers; and a
    smaller size against the CPU oracle directly."""
    node = Node([0] * G, exchange="copy")
    try:
        for log_n in (20, 12):
            N = 1 << log_n
            x = to_device(synth.random_elements(700 + log_n + G, N), prover.device)
            direct = prover.ntt_batch([x], N)[0]
            torch.cuda.synchronize()
            if log_n == 12:
                assert np.array_equal(to_host(direct), co.ntt(to_host(x), N))
            for rep in range(2):
                out = node.ntt(_shards(x, G), N)
                node.sync()
                assert torch.equal(torch.cat(out), direct), (G, log_n, rep)
            back = node.ntt(out, N, inverse=True)
            node.sync()
            assert torch.equal(torch.cat(back), x)
            N1 = 1 << ((log_n + 1) // 2)
            N2 = N // N1
            blk = node.ntt(_shards(x, G), N, natural_output=False)
            node.sync()
            got = torch.cat([b.view(N1 // G, N2, 4) for b in blk])              # [k1][k2] = X[k1 + N1 k2]
            assert torch.equal(got, direct.view(N2, N1, 4).permute(1, 0, 2))
    finally:
        node.close()


def test_node_ntt_2_24_two_ranks(prover):
    """The configured size of the sharded transform, 2^24 points, over two ranks sharing the GPU."""
    N = 1 << 24
    x = to_device(synth.random_elements(77, N), prover.device)
    direct = prover.ntt_batch([x], N)[0]
    torch.cuda.synchronize()
    node = Node([0, 0], exchange="copy")
    try:
        out = node.ntt(_shards(x, 2), N)
        node.sync()
        assert torch.equal(torch.cat(out), direct)
        del direct
        back = node.ntt(out, N, inverse=True)
        node.sync()
        assert torch.equal(torch.cat(back), x)
    finally:
        node.close()


def test_node_ntt_host_arrays():
    """plk_node_ntt_host: host array in, host array out, four ranks on GPU 0, against the CPU oracle."""
    node = Node([0] * 4, exchange="copy")
    try:
        for log_n in (10, 16):
            x = synth.random_elements(900 + log_n, 1 << log_n)
            y = node.ntt_host(x)
            assert np.array_equal(y, co.ntt(x, 1 << log_n))
            assert np.array_equal(node.ntt_host(y, inverse=True), x)
    finally:
        node.close()


def test_node_rccl_one_rank(prover):
    """RCCL itself (ncclCommInitAll, ncclAllToAll on the rank's stream between the library's kernels), bound at run time,
    in a node of one rank: what one GPU allows."""
    node = Node([0], exchange="rccl")
    try:
        for log_n in (8, 16, 22):
            N = 1 << log_n
            x = to_device(synth.random_elements(800 + log_n, N), prover.device)
            direct = prover.ntt_batch([x], N)[0]
            out = node.ntt([x], N)
            node.sync()
            assert torch.equal(out[0], direct), log_n
            back = node.ntt(out, N, inverse=True)
            node.sync()
            assert torch.equal(back[0], x)
            N1 = 1 << ((log_n + 1) // 2)
            blk = node.ntt([x], N, natural_output=False)       # two exchanges: [k1][k2] = X[k1 + N1 k2]
            node.sync()
            assert torch.equal(blk[0].view(N1, N // N1, 4), direct.view(N // N1, N1, 4).permute(1, 0, 2))
    finally:
        node.close()


def test_node_rccl_refuses_two_ranks_on_one_gpu():
    with pytest.raises(pa.PlonkError) as ei:
        Node([0, 0], exchange="rccl")
    assert ei.value.code == _lib.PLK_ERR_ARG


def test_node_argument_errors():
    node = Node([0, 0], exchange="copy")
    try:
        x = torch.zeros((24, 4), dtype=torch.int64, device="cuda:0")
        with pytest.raises(pa.PlonkError) as ei:
            node.ntt([x, x.clone()], 48)
        assert ei.value.code == _lib.PLK_ERR_NOT_POW2
        y = torch.zeros((1, 4), dtype=torch.int64, device="cuda:0")
        with pytest.raises(pa.PlonkError) as ei:
            node.ntt([y, y.clone()], 2)                     # the 2 x 1 view has a side the ranks do not divide
        assert ei.value.code == _lib.PLK_ERR_ARG
        with pytest.raises(pa.PlonkError) as ei:
            node.ntt([x, x], 32, out=[x, x])                # output aliases input
        assert ei.value.code == _lib.PLK_ERR_ARG
    finally:
        node.close()


def test_node_round23_batch(prover):
    """Batched proofs (replicas only): six circuits dealt to two ranks, every output in full against the CPU
    restatement; then a batch holding one circuit with a zero denominator, which must be named."""
    n, N = 1 << 12, 1 << 14
    node = Node([0, 0], exchange="copy")
    try:
        cs = [synth.make_circuit(n, circuit_id=200 + i) for i in range(6)]
        dcs = [prover.upload(c) for c in cs]
        outs = [(prover.empty(n + 1), prover.empty(N), prover.empty(N)) for _ in cs]
        node.round23_batch(dcs, N, outs)
        for c, (z, g, e) in zip(cs, outs):
            zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma)
            assert np.array_equal(to_host(z), zr) and np.array_equal(to_host(g), gr) and np.array_equal(to_host(e), er)
        import plonk_oracle as po
        beta, gamma = synth.mont_to_ints(cs[3].beta)[0], synth.mont_to_ints(cs[3].gamma)[0]
        bad = (-(beta * 9 + gamma)) % po.P
        dcs[3].wires[1][9] = torch.from_numpy(synth.ints_to_mont([bad]).view(np.int64)[0].copy()).to(prover.device)
        with pytest.raises(pa.PlonkError) as ei:
            node.round23_batch(dcs, N, outs)
        assert ei.value.code == _lib.PLK_ERR_ZERO_DENOMINATOR and "circuit 3" in str(ei.value)
        # the node is usable afterwards
        node.round23_batch(dcs[:2], N, outs[:2])
        zr = co.grand_product(cs[0].table.wires(), cs[0].perm.columns(), cs[0].beta, cs[0].gamma)
        assert np.array_equal(to_host(outs[0][0]), zr)
    finally:
        node.close()


def test_node_round23_rccl_one_rank(prover):
    """The one-proof mode with RCCL's all-to-all as its exchange, in the one-rank node a single GPU allows."""
    n, N = 1 << 12, 1 << 14
    node = Node([0], exchange="rccl")
    try:
        c = synth.make_circuit(n, circuit_id=401)
        dc = prover.upload(c)
        outs = [(prover.empty(n + 1), prover.empty(N), prover.empty(N))]
        for z_shift in (1, 4):
            node.round23([dc], N, outs, z_shift=z_shift)
            zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma, z_shift=z_shift)
            assert np.array_equal(to_host(outs[0][0]), zr) and np.array_equal(to_host(outs[0][1]), gr) and np.array_equal(to_host(outs[0][2]), er)
    finally:
        node.close()


def test_node_round23_host_arrays():
    """plk_node_round23_host: host arrays in and out over four ranks, against the CPU restatement; twice (the node keeps
    the columns' buffers), the second circuit smaller and without some selectors; then its argument errors."""
    node = Node([0] * 4, exchange="copy")
    try:
        for n, kinds, z_shift in ((1 << 12, None, 1), (1 << 8, ("Add", "Mult", "If"), 4)):
            from plonk_amd.api import AST_KINDS
            c = synth.make_circuit(n, circuit_id=500 + n, kinds=AST_KINDS if kinds is None else kinds)
            z, g, e = node.round23_host(c, 4 * n, z_shift=z_shift)
            zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), 4 * n, c.beta, c.gamma, z_shift=z_shift)
            assert np.array_equal(z, zr) and np.array_equal(g, gr) and np.array_equal(e, er)
        with pytest.raises(pa.PlonkError) as ei:
            node.round23_host(c, 2 * n)
        assert ei.value.code == _lib.PLK_ERR_ARG
    finally:
        node.close()


def test_node_round23_fewest_rows_and_columns():
    """Four rows over four ranks (one point of the trace domain each, the halo as long as the range), and eight ranks
    with nine columns to deal (no selectors: most ranks hold one column and an unused slot)."""
    from plonk_amd.device import DeviceProver
    prover = DeviceProver(0)
    for G, n, kinds, z_shift in ((4, 4, None, 4), (4, 4, (), 1), (8, 64, (), 4), (8, 8, ("Add",), 3)):
        from plonk_amd.api import AST_KINDS
        N = 4 * n
        node = Node([0] * G, exchange="copy")
        try:
            c = synth.make_circuit(n, circuit_id=77 + G + n, kinds=AST_KINDS if kinds is None else kinds)
            copies = [prover.upload(c) for _ in range(G)]
            outs = [(prover.empty(n + 1), prover.empty(N // G), prover.empty(N // G)) for _ in range(G)]
            node.round23(copies, N, outs, z_shift=z_shift)
            zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma, z_shift=z_shift)
            assert all(np.array_equal(to_host(o[0]), zr) for o in outs)
            assert np.array_equal(np.concatenate([to_host(o[1]) for o in outs]), gr), (G, n, kinds)
            assert np.array_equal(np.concatenate([to_host(o[2]) for o in outs]), er), (G, n, kinds)
        finally:
            node.close()


@pytest.mark.parametrize("G,kinds", [(1, None), (2, None), (4, ("Add", "Mul", "Const")), (8, None), (2, ())])
def test_node_round23_one_proof_over_the_ranks(prover, G, kinds):
    """plk_node_round23: the columns dealt to G ranks, one exchange, the constraints on N / G points per rank - the
    ranks' pieces joined are the vectors of the CPU restatement, Z whole on every rank; several z_shift, and a second
    call on the same node (its buffers reused)."""
    from plonk_amd.api import AST_KINDS
    n, N = 1 << 10, 1 << 12
    node = Node([0] * G, exchange="copy")
    try:
        for rep, z_shift in enumerate((1, 4, N // G)):
            c = synth.make_circuit(n, circuit_id=300 + G + rep, kinds=AST_KINDS if kinds is None else kinds)
            copies = [prover.upload(c) for _ in range(G)]
            outs = [(prover.empty(n + 1), prover.empty(N // G), prover.empty(N // G)) for _ in range(G)]
            node.round23(copies, N, outs, z_shift=z_shift)
            zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma, z_shift=z_shift)
            for z, _, _ in outs:
                assert np.array_equal(to_host(z), zr)
            assert np.array_equal(np.concatenate([to_host(o[1]) for o in outs]), gr), (G, z_shift)
            assert np.array_equal(np.concatenate([to_host(o[2]) for o in outs]), er), (G, z_shift)
    finally:
        node.close()


def test_node_round23_errors_and_large(prover):
    """Argument errors of plk_node_round23; a zero denominator named with its rank; and 2^16 rows over four ranks against
    the single-GPU call (itself checked against the oracle)."""
    node = Node([0] * 4, exchange="copy")
    try:
        n, N = 1 << 16, 1 << 18
        c = synth.make_circuit(n, circuit_id=321)
        copies = [prover.upload(c) for _ in range(4)]
        outs = [(prover.empty(n + 1), prover.empty(N // 4), prover.empty(N // 4)) for _ in range(4)]
        with pytest.raises(pa.PlonkError) as ei:
            node.round23(copies, 2 * n, outs)                        # only the 4n domain
        assert ei.value.code == _lib.PLK_ERR_ARG
        with pytest.raises(pa.PlonkError) as ei:
            node.round23(copies, N, outs, z_shift=N // 4 + 1)        # halo longer than a rank's range
        assert ei.value.code == _lib.PLK_ERR_ARG
        node.round23(copies, N, outs, z_shift=4)
        z, g, e = prover.round23(copies[0], N, z_shift=4)
        torch.cuda.synchronize()
        assert torch.equal(outs[2][0], z)
        assert torch.equal(torch.cat([o[1] for o in outs]), g) and torch.equal(torch.cat([o[2] for o in outs]), e)
        import plonk_oracle as po
        beta, gamma = synth.mont_to_ints(c.beta)[0], synth.mont_to_ints(c.gamma)[0]
        bad = synth.ints_to_mont([(-(beta * 9 + gamma)) % po.P]).view(np.int64)[0].copy()
        for dc in copies:
            dc.wires[1][9] = torch.from_numpy(bad).to(prover.device)
        with pytest.raises(pa.PlonkError) as ei:
            node.round23(copies, N, outs)
        assert ei.value.code == _lib.PLK_ERR_ZERO_DENOMINATOR and "rank" in str(ei.value)
    finally:
        node.close()


def test_device_memory_helpers(ctx):
    """plk_dev_alloc / upload / download / free: what a host without a HIP binding uses around the plk_dev_* calls."""
    lib = ctx.lib
    x = synth.random_elements(5, 1 << 10)
    p = C.c_void_p()
    ctx.check(lib.plk_dev_alloc(ctx.handle, x.nbytes, C.byref(p)))
    st = lib.plk_ctx_stream(ctx.handle)
    ctx.check(lib.plk_dev_upload(ctx.handle, st, p, x.ctypes.data, x.nbytes))
    tab = (C.c_void_p * 1)(p.value)
    ctx.check(lib.plk_dev_ntt_batch(ctx.handle, st, C.cast(tab, _lib.u64pp), None, 1, 1 << 10, 1 << 10, 1, C.cast(tab, _lib.u64pp)))
    y = np.empty_like(x)
    ctx.check(lib.plk_dev_download(ctx.handle, st, y.ctypes.data, p, y.nbytes))
    lib.plk_dev_free(ctx.handle, p)
    assert np.array_equal(y, co.intt(x, 1 << 10))
    # 16 MiB of ordinary (pageable) memory: the threaded staging copy, both directions
    big = synth.random_elements(6, 1 << 19)
    ctx.check(lib.plk_dev_alloc(ctx.handle, big.nbytes, C.byref(p)))
    ctx.check(lib.plk_dev_upload(ctx.handle, st, p, big.ctypes.data, big.nbytes))
    back = np.zeros_like(big)
    ctx.check(lib.plk_dev_download(ctx.handle, st, back.ctypes.data, p, back.nbytes))
    lib.plk_dev_free(ctx.handle, p)
    assert np.array_equal(back, big)
    assert lib.plk_dev_upload(ctx.handle, st, None, big.ctypes.data, 64) == _lib.PLK_ERR_ARG
