"""Times one 2^20-term multi-scalar multiplication for scalars that crowd into few buckets (run on a GPU from the repository root)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import g1_oracle as g1
from plonk_amd import synth
from plonk_amd.device import DeviceProver, to_device
p = DeviceProver(0)
gen = np.array(g1.fq_to_mont_limbs(g1.GX) + g1.fq_to_mont_limbs(g1.GY), dtype=np.uint64)
n = 1 << 20
p.msm_set_bases(p.g1_multiples(gen, 1, n))
one = synth.ints_to_mont([1])[0]
rng = np.random.default_rng(1)
cases = {"random": synth.random_elements(3, n), "all ones": np.tile(one, (n, 1)),
         "below 2^16": synth.ints_to_mont([int(v) for v in rng.integers(0, 1 << 16, n)]),
         "two values": synth.ints_to_mont([0x1234567 if k % 3 else (1 << 200) + 5 for k in range(n)])}
for name, s in cases.items():
    d = to_device(s, p.device)
    p.msm(d); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): p.msm(d)
    print("2^20 terms, %s: %.2f ms" % (name, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
