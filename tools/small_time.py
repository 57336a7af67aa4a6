"""Rounds 2+3 of a small circuit (plk_dev_round23, n = 2^log_n): time per call when calls are issued back to back
(throughput), when each call is waited for (latency), and the host time spent issuing one call.
PLONK_HIP_GRAPHS=0 turns the recorded-launch path off for comparison."""
import sys
import time

sys.path.insert(0, ".")
import torch

from plonk_amd import synth
from plonk_amd.device import DeviceProver

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = 1 << log_n
p = DeviceProver(0)
dc = p.upload(synth.make_circuit(n))
out = (p.empty(n + 1), p.empty(4 * n), p.empty(4 * n))
for _ in range(10):
    p.round23(dc, 4 * n, out=out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    p.round23(dc, 4 * n, out=out)
issue = (time.perf_counter() - t) / reps
torch.cuda.synchronize()
through = (time.perf_counter() - t) / reps
t = time.perf_counter()
for _ in range(reps):
    p.round23(dc, 4 * n, out=out)
    torch.cuda.synchronize()
lat = (time.perf_counter() - t) / reps
p.status()
print(f"n=2^{log_n}: issue {issue * 1e3:.4f} ms  back-to-back {through * 1e3:.4f} ms  waited-for {lat * 1e3:.4f} ms per call")

# the same with the argument tables built once: the library's own share of the issue time
from plonk_amd import _lib
from plonk_amd.device import _table, _ptr, AST_KINDS
args = (p.ctx.handle, p._stream(), _table(dc.wires), _table([dc.selectors.get(k) for k in AST_KINDS]), _table(dc.sigma), n, 4 * n,
        dc.beta.ctypes.data_as(_lib.u64p), dc.gamma.ctypes.data_as(_lib.u64p), 1, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]))
fn = p.lib.plk_dev_round23
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    fn(*args)
issue = (time.perf_counter() - t) / reps
torch.cuda.synchronize()
through = (time.perf_counter() - t) / reps
print(f"          C ABI alone: issue {issue * 1e3:.4f} ms  back-to-back {through * 1e3:.4f} ms")
z = p.empty(n + 1)
gargs = (p.ctx.handle, p._stream(), _table(dc.wires), _table(dc.sigma), n, dc.beta.ctypes.data_as(_lib.u64p), dc.gamma.ctypes.data_as(_lib.u64p), _ptr(z))
fn = p.lib.plk_dev_grand_product
fn(*gargs)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    fn(*gargs)
issue = (time.perf_counter() - t) / reps
torch.cuda.synchronize()
through = (time.perf_counter() - t) / reps
print(f"          grand product alone: issue {issue * 1e3:.4f} ms  back-to-back {through * 1e3:.4f} ms")
