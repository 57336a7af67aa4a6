"""Wall time of the host entry point plk_round23 (caller-owned host memory in and out) at n = 2^log_n."""
import sys
import time

sys.path.insert(0, ".")
import plonk_amd as pa
from plonk_amd import synth

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
c = synth.make_circuit(n)
ctx = pa.Context(0)
D = pa.Radix2EvaluationDomain
for it in range(4):
    t0 = time.perf_counter()
    z, g, p = pa.prove_round23(c.table, c.perm, c.beta, c.gamma, D(n), D(4 * n), ctx=ctx)
    print(f"call {it}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
mb = (13 * 32 * n + 4 * 8 * n + 2 * 32 * 4 * n + 32 * (n + 1)) / 1e6
print(f"{mb:.0f} MB cross PCIe per call")
