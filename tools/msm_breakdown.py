"""Print the per-kernel times of a `bench.py --workload msm` JSON line, launches of one kernel added up."""
import json
import sys

d = json.loads(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_msm.json").read().strip().splitlines()[-1])
agg = {}
for k in d["kernels"]:
    name = k["label"].split("/")[0]
    agg[name] = agg.get(name, 0) + k["ms_per_step"]
for name, v in sorted(agg.items(), key=lambda x: -x[1]):
    print(f"{name:32s} {v:8.3f}")
print("kernels", round(sum(agg.values()), 3), "step", d["value"], d.get("cpu_baseline"), d.get("matches_cpu"))
