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
# the same call on page-locked columns (outputs still pageable: the mirror allocates them with numpy)
from plonk_amd.api import GateEvaluationTable, PermutationTable


def pin(a):
    b = ctx.pinned_empty(a.shape, a.dtype)
    b[...] = a
    return b


pt = GateEvaluationTable(*[pin(w) for w in c.table.wires()], selectors={k: pin(v) for k, v in c.table.selectors.items()})
pp = PermutationTable(*[pin(s) for s in c.perm.columns()])
for it in range(3):
    t0 = time.perf_counter()
    z2, g2, p2 = pa.prove_round23(pt, pp, c.beta, c.gamma, D(n), D(4 * n), ctx=ctx)
    print(f"pinned inputs, call {it}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
outs = (ctx.pinned_empty((n + 1, 4)), ctx.pinned_empty((4 * n, 4)), ctx.pinned_empty((4 * n, 4)))
for it in range(3):
    t0 = time.perf_counter()
    z2, g2, p2 = pa.prove_round23(pt, pp, c.beta, c.gamma, D(n), D(4 * n), ctx=ctx, out=outs)
    print(f"pinned inputs and outputs, call {it}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
assert (z2 == z).all() and (g2 == g).all() and (p2 == p).all()
mb = (13 * 32 * n + 4 * 8 * n + 2 * 32 * 4 * n + 32 * (n + 1)) / 1e6
print(f"{mb:.0f} MB cross PCIe per call")
