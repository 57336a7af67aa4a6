"""Worker of tests/test_dist_gloo.py: runs under torch.distributed.run with the gloo backend and
checks plonk_amd.dist against the single-process oracle.  Exit code != 0 on any mismatch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import numpy as np
import torch
import torch.distributed as dist

import c_oracle as co
from oracle_engine import OracleEngine, n2t, t2n
from plonk_amd import dist as pdist
from plonk_amd import synth


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    eng = OracleEngine()

    # batched-proof mode: a partition of the circuit ids, no overlap, no gap
    mine = pdist.shard_circuits(11)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    assert sorted(sum(gathered, [])) == list(range(11)), gathered

    # four-step transform, natural order in and out, both directions, two sizes (square and non-square view)
    for log_n in (6, 9):
        N = 1 << log_n
        x = synth.random_elements(4242 + log_n, N)
        shard = n2t(x[rank * N // world:(rank + 1) * N // world])
        for inverse in (False, True):
            want = co.intt(x, N) if inverse else co.ntt(x, N)
            got = pdist.four_step_ntt(shard, N, eng, inverse=inverse)
            assert np.array_equal(t2n(got), want[rank * N // world:(rank + 1) * N // world]), (log_n, inverse, "natural")
        # without the last exchange: block [my k1][k2] = X[k1 + N1 k2]
        N1, N2 = pdist.split_domain(N)
        blk = t2n(pdist.four_step_ntt(shard, N, eng, natural_output=False))
        full = co.ntt(x, N)
        r1 = N1 // world
        for i in range(r1):
            for k2 in (0, 1, N2 - 1):
                assert np.array_equal(blk[i, k2], full[(rank * r1 + i) + N1 * k2])

    # one multi-scalar multiplication with its terms dealt to the ranks
    import g1_oracle as g1
    gen = np.array(g1.fq_to_mont_limbs(g1.GX) + g1.fq_to_mont_limbs(g1.GY), dtype=np.uint64)
    terms = 8 * world
    bases = np.stack([co.g1_msm(synth.ints_to_mont([3 * k + 1]), gen[None, :]) for k in range(terms)])
    scal = synth.random_elements(31, terms)
    lo, hi = rank * terms // world, (rank + 1) * terms // world
    eng.msm_set_bases(bases[lo:hi])
    got = pdist.msm_sharded(eng, n2t(scal[lo:hi]))
    assert np.array_equal(got, co.g1_msm(scal, bases)), "sharded msm"

    # one proof over all ranks
    n, N = 32, 128
    c = synth.make_circuit(n, circuit_id=7)
    wires = [n2t(w) for w in c.table.wires()]
    sels = [n2t(c.table.selectors[k]) if k in c.table.selectors else None for k in co.AST_KINDS]
    sigma = [n2t(s) for s in c.perm.columns()]
    for shift in (1, 4):
        z, gate, perm = pdist.round23_poly_sharded(eng, wires, sels, sigma, c.beta, c.gamma, n, N, z_shift=shift)
        zr, gr, pr = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma, z_shift=shift)
        lo, hi = rank * N // world, (rank + 1) * N // world
        assert np.array_equal(t2n(z), zr)
        assert np.array_equal(t2n(gate), gr[lo:hi]), "gate chunk"
        assert np.array_equal(t2n(perm), pr[lo:hi]), ("perm chunk", shift)
    # the grand product alone, rows split over the ranks: even and uneven blocks, and fewer rows than ranks
    for rows_n in (32, 27, 1):
        cc = synth.make_circuit(32, circuit_id=9)
        w_t = [n2t(w[:rows_n]) for w in cc.table.wires()]
        sg = [np.minimum(s[:rows_n], np.uint64(rows_n - 1)) for s in cc.perm.columns()]
        z = pdist.grand_product_sharded(eng, w_t, [n2t(s) for s in sg], rows_n, cc.beta, cc.gamma)
        want = co.grand_product([w[:rows_n] for w in cc.table.wires()], sg, cc.beta, cc.gamma)
        assert np.array_equal(t2n(z), want), ("sharded grand product", rows_n)
    # absent selector columns
    c2 = synth.make_circuit(n, circuit_id=8, kinds=("Add", "Mult", "Eq"))
    sels2 = [n2t(c2.table.selectors[k]) if k in c2.table.selectors else None for k in co.AST_KINDS]
    z, gate, perm = pdist.round23_poly_sharded(eng, [n2t(w) for w in c2.table.wires()], sels2, [n2t(s) for s in c2.perm.columns()],
                                               c2.beta, c2.gamma, n, N)
    zr, gr, pr = co.round23(c2.table.wires(), c2.table.selectors, c2.perm.columns(), N, c2.beta, c2.gamma)
    lo, hi = rank * N // world, (rank + 1) * N // world
    assert np.array_equal(t2n(gate), gr[lo:hi]) and np.array_equal(t2n(perm), pr[lo:hi])

    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok")


if __name__ == "__main__":
    main()
