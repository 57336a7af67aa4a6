"""Randomised sweep of the device-resident rounds 2+3 against the CPU oracle: sizes 2^2 .. 2^15, any subset of the
selector kinds, any z_shift, challenges and columns fresh per case.  usage: python tools/fuzz_round23.py [cases] [seed] [field]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import c_oracle as co
from plonk_amd import synth
from plonk_amd.api import AST_KINDS
from plonk_amd.device import DeviceProver, to_host

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
field = int(sys.argv[3]) if len(sys.argv) > 3 else 0
prover = DeviceProver(0, field=field)
F = synth.field(field)
bad = 0
for case in range(cases):
    log_n = int(rng.integers(2, 16))
    n = 1 << log_n
    blow = 4 if rng.random() < 0.8 else int(rng.choice([1, 2, 8]))      # mostly the coset form, sometimes the plain one
    N = blow * n
    kinds = tuple(k for k in AST_KINDS if rng.random() < 0.6)
    z_shift = int(rng.choice([1, 1, 4, 3, N - 1, int(rng.integers(0, N))]))
    c = F.make_circuit(n, circuit_id=int(rng.integers(0, 1 << 20)), kinds=kinds)
    dc = prover.upload(c)
    z, g, p = prover.round23(dc, N, z_shift=z_shift)
    prover.status()
    with co.use_field(field):
        zr, gr, pr = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma, z_shift=z_shift)
    ok = np.array_equal(to_host(z), zr) and np.array_equal(to_host(g), gr) and np.array_equal(to_host(p), pr)
    bad += not ok
    print(f"case {case}: n=2^{log_n} N={blow}n kinds={len(kinds)} z_shift={z_shift} {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
