"""CPU tests of the multi-process paths (plonk_amd.dist) over gloo: the exchanges and index
bookkeeping run for real between processes; the local arithmetic is supplied by the oracle."""
import os
import subprocess
import sys

import pytest

from plonk_amd import dist as pdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,port", [(2, 29731), (4, 29732), (8, 29733)])
def test_dist_paths_over_gloo(world, port):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(" ok") == world


def test_single_process_helpers():
    assert pdist.split_domain(1 << 24) == (1 << 12, 1 << 12)
    assert pdist.split_domain(1 << 9) == (32, 16)
    assert pdist.shard_circuits(5) == [0, 1, 2, 3, 4]
    with pytest.raises(ValueError):
        pdist.split_domain(12)
