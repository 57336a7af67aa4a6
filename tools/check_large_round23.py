"""One-off check of the device-resident rounds 2+3 at sizes above the headline's (three-pass transform plans) against
the CPU oracle, full vectors.  usage: python tools/check_large_round23.py [log_n ...]   (a minute of CPU time per 2^22 rows)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import c_oracle as co
from plonk_amd import synth
from plonk_amd.device import DeviceProver, to_host

prover = DeviceProver(0)
bad = 0
for log_n in [int(a) for a in sys.argv[1:]] or [21]:
    n, N = 1 << log_n, 4 << log_n
    c = synth.make_circuit(n, circuit_id=log_n)
    t0 = time.time()
    z, g, p = prover.round23(prover.upload(c), N)
    prover.status()
    z, g, p = to_host(z), to_host(g), to_host(p)
    t1 = time.time()
    zr, gr, pr = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma)
    ok = np.array_equal(z, zr) and np.array_equal(g, gr) and np.array_equal(p, pr)
    bad += not ok
    print(f"n=2^{log_n}: {'ok' if ok else 'MISMATCH'}  (device incl. transfers {t1 - t0:.1f} s, oracle {time.time() - t1:.1f} s)", flush=True)
sys.exit(1 if bad else 0)
