"""Per-kernel time of the stand-alone grand product (plk_dev_grand_product) at n = 2^log_n."""
import sys

sys.path.insert(0, ".")
import torch

from plonk_amd import synth
from plonk_amd.device import DeviceProver

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
p = DeviceProver(0)
dc = p.upload(synth.make_circuit(1 << log_n))
z = p.empty((1 << log_n) + 1)
for _ in range(3):
    p.grand_product(dc, out=z)
torch.cuda.synchronize()
p.profile_begin()
t0 = torch.cuda.Event(enable_timing=True)
t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(20):
    p.grand_product(dc, out=z)
t1.record()
torch.cuda.synchronize()
print(f"{t0.elapsed_time(t1) / 20:.4f} ms per call", {r["label"]: round(r["total_ms"] / r["launches"], 4) for r in p.profile_end()})
p.status()
