"""A short randomised sweep of the device-resident rounds 2+3 against the CPU oracle, both fields: sizes 2^2 .. 2^13, any
subset of the selector kinds, mostly the 4n domain (the coset form) and sometimes n, 2n or 8n (the plain form), any
z_shift.  tools/fuzz_round23.py is the same sweep at any length."""
import numpy as np
import pytest

import c_oracle as co
from plonk_amd import synth
from plonk_amd.api import AST_KINDS
from plonk_amd.device import DeviceProver, to_host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("field,seed,fused_from", [(0, 11, None), (1, 12, None), (0, 13, "0"), (1, 14, "0")])
def test_random_circuits_match_the_oracle(field, seed, fused_from, monkeypatch):
    """`fused_from` = "0": the context takes both constraint vectors from the one kernel at every size (the library
    does so from 2^16 rows on), so that the small random cases cover that kernel as well."""
    if fused_from is not None:
        monkeypatch.setenv("PLONK_HIP_FUSED_MIN_LOG", fused_from)
    rng = np.random.default_rng(seed)
    prover = DeviceProver(0, field=field)
    F = synth.field(field)
    for case in range(16):
        log_n = int(rng.integers(2, 14))
        n = 1 << log_n
        blow = 4 if rng.random() < 0.75 else int(rng.choice([1, 2, 8]))
        N = blow * n
        kinds = tuple(k for k in AST_KINDS if rng.random() < 0.6)
        z_shift = int(rng.choice([1, 4, 3, N - 1, int(rng.integers(0, N))]))
        c = F.make_circuit(n, circuit_id=int(rng.integers(0, 1 << 20)), kinds=kinds)
        dc = prover.upload(c)
        z, g, p = prover.round23(dc, N, z_shift=z_shift)
        prover.status()
        with co.use_field(field):
            zr, gr, pr = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma, z_shift=z_shift)
        what = (field, case, log_n, blow, kinds, z_shift)
        assert np.array_equal(to_host(z), zr), what
        assert np.array_equal(to_host(g), gr), what
        assert np.array_equal(to_host(p), pr), what
