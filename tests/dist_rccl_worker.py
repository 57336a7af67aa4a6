"""Worker of tests/test_gpu_dist.py::test_rccl_runs_in_a_one_rank_group: a one-rank RCCL ("nccl") process group on
GPU 0 with the collectives forced (PLONK_DIST_FORCE_COLLECTIVES=1), so that RCCL's all-to-all and all-gather really
run between the library's kernels, on device tensors, in stream order and without host synchronisation."""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
os.environ["PLONK_DIST_FORCE_COLLECTIVES"] = "1"

import numpy as np
import torch
import torch.distributed as dist

import c_oracle as co
from plonk_amd import dist as pdist
from plonk_amd import synth
from plonk_amd.device import DeviceProver, to_device, to_host


def main():
    faulthandler.dump_traceback_later(200, exit=True)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{os.environ.get('PLONK_TEST_PORT', '29671')}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    assert pdist._FORCE and dist.get_backend() == "nccl"
    eng = DeviceProver(0)
    d = eng.device
    N = 1 << 20
    x = to_device(synth.random_elements(901, N), d)
    direct = eng.ntt_batch([x], N)[0]
    outs = []
    for _ in range(4):                                   # back to back: kernels and collectives alternate on the GPU
        y = pdist.four_step_ntt(x, N, eng)
        outs.append(pdist.four_step_ntt(y, N, eng, inverse=True))
    torch.cuda.synchronize()
    assert torch.equal(y, direct), "four-step through RCCL differs from the direct transform"
    assert all(torch.equal(o, x) for o in outs), "round trip through RCCL"
    # the sharded grand product's all-gathers and the one-proof-over-the-group path
    n, NN = 1 << 12, 1 << 14
    c = synth.make_circuit(n, circuit_id=8)
    dc = eng.upload(c)
    sels = [dc.selectors.get(k) for k in co.AST_KINDS]
    zr, gr, pr = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), NN, c.beta, c.gamma)
    for _ in range(3):
        z, gate, perm = pdist.round23_poly_sharded(eng, dc.wires, sels, dc.sigma, c.beta, c.gamma, n, NN)
    eng.status()
    assert np.array_equal(to_host(z), zr) and np.array_equal(to_host(gate), gr) and np.array_equal(to_host(perm), pr)
    t = torch.ones(4, device=d)
    dist.all_reduce(t)
    dist.barrier()
    dist.destroy_process_group()
    print("rccl one-rank ok")


if __name__ == "__main__":
    main()
