"""Worker of tests/test_gpu_dist.py::test_two_ranks_share_the_gpu: two (or more) ranks on GPU 0,
exchanges over gloo (RCCL refuses several ranks on one device), every local computation in
libplonk_hip.so.  Checks plonk_amd.dist against the CPU oracle.  Exit code != 0 on any mismatch."""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import numpy as np
import torch
import torch.distributed as dist

import c_oracle as co
from plonk_amd import dist as pdist
from plonk_amd import synth
from plonk_amd.device import DeviceProver, to_device, to_host


def main():
    faulthandler.dump_traceback_later(240, exit=True)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    eng = DeviceProver(0)
    d = eng.device

    for log_n in (10, 17):
        N = 1 << log_n
        x = synth.random_elements(777 + log_n, N)
        lo, hi = rank * N // world, (rank + 1) * N // world
        for inverse in (False, True):
            want = co.intt(x, N) if inverse else co.ntt(x, N)
            got = pdist.four_step_ntt(to_device(x[lo:hi], d), N, eng, inverse=inverse)
            assert np.array_equal(to_host(got), want[lo:hi]), (log_n, inverse)

    # a size past the CPU oracle's comfort: the sharded transform against the library's direct transform of the whole
    # vector (itself checked against the oracle up to 2^18 and by closed forms beyond), both output layouts
    N = 1 << 20
    x = synth.random_elements(778, N)
    lo, hi = rank * N // world, (rank + 1) * N // world
    direct = eng.ntt_batch([to_device(x, d)], N)[0]
    shard = to_device(x[lo:hi], d)
    got = pdist.four_step_ntt(shard, N, eng)
    assert torch.equal(got, direct[lo:hi]), "four-step at 2^20, natural order"
    blk = pdist.four_step_ntt(shard, N, eng, natural_output=False)            # [my k1][k2] = X[k1 + N1 k2]
    N1, N2 = pdist.split_domain(N)
    r1 = N1 // world
    assert torch.equal(blk, direct.view(N2, N1, 4)[:, rank * r1:(rank + 1) * r1].permute(1, 0, 2)), "four-step at 2^20, transposed output"
    assert torch.equal(pdist.four_step_ntt(got, N, eng, inverse=True), shard), "four-step at 2^20, inverse"
    del direct, blk, got

    if world == 2:
        # the full size of the sharded-transform configuration, 2^24 points: every rank builds the same input on the
        # device, transforms its shard through the exchanges and compares with its range of the direct transform
        N = 1 << 24
        g = torch.Generator(device=d)
        g.manual_seed(2024)
        x = torch.randint(-(1 << 63), (1 << 63) - 1, (N, 4), dtype=torch.int64, device=d, generator=g)
        x[:, 3] &= (1 << 62) - 1                                             # below 2^254: reduced elements
        lo, hi = rank * N // world, (rank + 1) * N // world
        direct = eng.ntt_batch([x], N)[0][lo:hi].clone()
        shard = x[lo:hi].clone()
        del x
        got = pdist.four_step_ntt(shard, N, eng)
        assert torch.equal(got, direct), "four-step at 2^24, natural order"
        assert torch.equal(pdist.four_step_ntt(got, N, eng, inverse=True), shard), "four-step at 2^24, inverse"
        del direct, got, shard

    for rows_n in (1 << 12, 3001):               # even and uneven row blocks
        cc = synth.make_circuit(1 << 12, circuit_id=5)
        w_h = [w[:rows_n] for w in cc.table.wires()]
        sg_h = [np.minimum(s[:rows_n], np.uint64(rows_n - 1)) for s in cc.perm.columns()]
        z = pdist.grand_product_sharded(eng, [to_device(w, d) for w in w_h], [to_device(s, d) for s in sg_h], rows_n, cc.beta, cc.gamma)
        eng.status()
        assert np.array_equal(to_host(z), co.grand_product(w_h, sg_h, cc.beta, cc.gamma)), ("sharded grand product", rows_n)

    # one multi-scalar multiplication with its terms dealt to the ranks (each rank's table holds its shard of the bases)
    import g1_oracle as g1
    gen = np.array(g1.fq_to_mont_limbs(g1.GX) + g1.fq_to_mont_limbs(g1.GY), dtype=np.uint64)
    terms = 1 << 12
    lo, hi = rank * terms // world, (rank + 1) * terms // world
    shard = eng.g1_multiples(gen, 1 + lo, hi - lo)              # (lo + 1 + i) G
    eng.msm_set_bases(shard)
    scal = synth.random_elements(41, terms)
    got = pdist.msm_sharded(eng, to_device(scal[lo:hi], d))
    total = sum(v * (i + 1) for i, v in enumerate(synth.mont_to_ints(scal))) % g1.R_ORDER
    want = g1.mul(total, g1.G1)
    assert [int(x) for x in got] == g1.fq_to_mont_limbs(want[0]) + g1.fq_to_mont_limbs(want[1]), "sharded msm"

    n, N = 1 << 10, 1 << 12
    c = synth.make_circuit(n, circuit_id=3)
    dc = eng.upload(c)
    sels = [dc.selectors.get(k) for k in co.AST_KINDS]
    zr, gr, pr = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma)
    lo, hi = rank * N // world, (rank + 1) * N // world
    for _ in range(3):                           # repeated calls: workspace and exchange reuse
        z, gate, perm = pdist.round23_poly_sharded(eng, dc.wires, sels, dc.sigma, c.beta, c.gamma, n, N)
        eng.status()
        assert np.array_equal(to_host(z), zr)
        assert np.array_equal(to_host(gate), gr[lo:hi]), "gate chunk"
        assert np.array_equal(to_host(perm), pr[lo:hi]), "perm chunk"

    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok")


if __name__ == "__main__":
    main()
