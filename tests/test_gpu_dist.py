"""GPU tests of the multi-GPU building blocks on one GPU (world size 1: the exchanges are the
identity, every local computation is the HIP library's)."""
import numpy as np
import pytest
import torch

import c_oracle as co
import plonk_oracle as po
from plonk_amd import dist as pdist
from plonk_amd import synth
from plonk_amd.device import to_device, to_host

pytestmark = pytest.mark.gpu


def test_strided_rows_mul_and_twiddles(prover):
    rows, length, N = 37, 64, 256
    mat = synth.random_elements(91, rows * length).reshape(rows, length, 4)
    got = to_host(prover.ntt_rows(to_device(mat, prover.device), N))
    for r in (0, 1, 17, 36):
        assert np.array_equal(got[r], co.ntt(mat[r], N))
    inv = to_host(prover.ntt_rows(to_device(mat, prover.device), length, inverse=True))
    assert np.array_equal(inv[5], co.intt(mat[5], length))
    a, b = synth.random_elements(1, 500), synth.random_elements(2, 500)
    prod = to_host(prover.mul(to_device(a, prover.device), to_device(b, prover.device)))
    assert all(np.array_equal(prod[i], co.fe_mul(a[i], b[i])) for i in (0, 3, 499))
    tw = synth.mont_to_ints(to_host(prover.twiddle_block(1 << 20, rows=3, cols=5, row0=1000, col0=7)).reshape(-1, 4))
    w = po.root_of_unity(1 << 20)
    assert tw == [pow(w, (1000 + r) * (7 + c), po.P) for r in range(3) for c in range(5)]
    twi = synth.mont_to_ints(to_host(prover.twiddle_block(1 << 10, 2, 2, 3, 0, inverse=True, scale_by_inverse_n=True)).reshape(-1, 4))
    wi, ninv = pow(po.root_of_unity(1 << 10), -1, po.P), pow(1 << 10, -1, po.P)
    assert twi == [pow(wi, (3 + r) * c, po.P) * ninv % po.P for r in range(2) for c in range(2)]


@pytest.mark.parametrize("log_n", [8, 13, 16])
def test_four_step_matches_direct_transform(prover, log_n):
    N = 1 << log_n
    x = synth.random_elements(300 + log_n, N)
    xd = to_device(x, prover.device)
    got = pdist.four_step_ntt(xd, N, prover)
    assert np.array_equal(to_host(got), co.ntt(x, N))
    back = pdist.four_step_ntt(got, N, prover, inverse=True)
    assert torch.equal(back, xd)


def test_four_step_2_24_roundtrip_and_agreement(prover):
    """The 2^24-point transform on one GPU: the four-step result equals the library's direct
    three-pass transform (itself checked against the oracle) and inverts back to the input."""
    N = 1 << 24
    x = to_device(synth.random_elements(77, N), prover.device)
    direct = prover.ntt_batch([x], N)[0]
    four = pdist.four_step_ntt(x, N, prover)
    assert torch.equal(four, direct)
    assert torch.equal(pdist.four_step_ntt(four, N, prover, inverse=True), x)


def test_poly_sharded_round23_single_rank(prover):
    n, N = 1 << 12, 1 << 14
    c = synth.make_circuit(n, circuit_id=5)
    dc = prover.upload(c)
    sels = [dc.selectors.get(k) for k in po.AST_KINDS]
    z, g, e = pdist.round23_poly_sharded(prover, dc.wires, sels, dc.sigma, c.beta, c.gamma, n, N)
    prover.status()
    zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma)
    assert np.array_equal(to_host(z), zr) and np.array_equal(to_host(g), gr) and np.array_equal(to_host(e), er)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ranks_share_the_gpu(world):
    """several ranks on GPU 0 (exchanges over gloo): the sharded transform and the one-proof-over-G-ranks
    path with the HIP engine, against the CPU oracle; with two ranks also the sharded transform at its full
    configured size, 2^24 points, against the direct transform."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29620 + world), os.path.join(root, "tests", "dist_gpu_worker.py")]
    r = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="2"), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(" ok") == world


def test_rccl_runs_in_a_one_rank_group():
    """RCCL itself (backend "nccl"), which needs one GPU per rank, in a group of one rank with the collectives forced:
    all-to-all, all-gather, all-reduce and barrier between the library's kernels on device tensors."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "dist_rccl_worker.py")], env=dict(os.environ, OMP_NUM_THREADS="4"),
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rccl one-rank ok" in r.stdout


def test_four_step_fused_phases_all_layouts(prover):
    """The fused phases at world size 1 for sizes on both sides of the single-pass / two-pass boundary (a side above
    2^7 points takes two passes), both directions and both output layouts, against the direct transform."""
    for log_n in (4, 7, 14, 15, 16, 19, 22):
        N = 1 << log_n
        x = to_device(synth.random_elements(500 + log_n, N), prover.device)
        direct = prover.ntt_batch([x], N)[0]
        assert torch.equal(pdist.four_step_ntt(x, N, prover), direct), log_n
        N1, N2 = pdist.split_domain(N)
        blk = pdist.four_step_ntt(x, N, prover, natural_output=False)
        assert torch.equal(blk, direct.view(N2, N1, 4).permute(1, 0, 2)), log_n
        assert torch.equal(pdist.four_step_ntt(direct, N, prover, inverse=True), x), log_n
        if log_n <= 16:
            assert np.array_equal(to_host(direct), co.ntt(to_host(x), N)), log_n


@pytest.mark.parametrize("n,ranges", [(1 << 14, 4), (5000, 3), (1025, 2), (7, 7)])
def test_grand_product_by_row_ranges(prover, n, ranges):
    """plk_dev_grand_product_local / _finish on one GPU, range after range, against the serial recurrence."""
    c = synth.make_circuit(1 << 14, circuit_id=11)
    w_h = [w[:n] for w in c.table.wires()]
    sg_h = [np.minimum(s[:n], np.uint64(n - 1)) for s in c.perm.columns()]
    wires = [to_device(w, prover.device) for w in w_h]
    sigma = [to_device(s, prover.device) for s in sg_h]
    per = (n + ranges - 1) // ranges
    bounds = [(min(g * per, n), min(per, n - min(g * per, n))) for g in range(ranges)]
    totals = torch.stack([prover.grand_product_local(wires, sigma, n, f, r, c.beta, c.gamma) for f, r in bounds])
    want = co.grand_product(w_h, sg_h, c.beta, c.gamma)
    for g, (f, r) in enumerate(bounds):
        prover.grand_product_local(wires, sigma, n, f, r, c.beta, c.gamma)
        z_blk = prover.grand_product_finish(r, totals, g)
        prover.status()
        assert np.array_equal(to_host(z_blk), want[f + 1:f + 1 + r]), (g, f, r)


def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` started alone (no launcher around it) starts two ranks itself as a child process and prints
    ONE line that says so, with the sharded transform's figure in it; here the two ranks share GPU 0 and exchange
    over gloo, because RCCL wants one GPU per rank."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PLONK_BENCH_SHARE_GPU="1", PLONK_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "16"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    line = lines[0]
    assert line["n_gpus"] == 2 and line["config"]["circuits_per_step"] == 2
    assert line["four_step"]["round_trip_bit_exact"] is True and line["four_step"]["ms"] > 0
    # and the same transform through the native node layer (a child process of rank 0: two ranks, copy exchange here)
    assert line["node_ntt"]["round_trip_bit_exact"] is True and line["node_ntt"]["ms"] > 0
    # and one proof of 2^20 rows dealt over the node's two ranks, compared in full with the single-GPU call
    assert line["node_round23"]["matches_single_gpu"] is True and line["node_round23"]["ms"] > 0
