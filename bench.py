#!/usr/bin/env python3
"""Benchmark of the PLONK polynomial hot path on MI355X.

One step = prover rounds 2+3 for one synthetic circuit of n = 2^20 rows (BASELINE.json config 3):
grand product Z, interpolation of the 18 columns (4 wires, 9 selectors, 4 sigma, Z), their
low-degree extension to the N = 4n domain (wires evaluated once), gate-constraint and
permutation-constraint evaluation.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W

With N > 1 every rank (one process per GPU) proves its own circuit per step (batched-proof mode, no
data-path collective): weak scaling.  `value` is the time per circuit over the whole job,
ms_per_step / N.  The ranks come either from an outer `python -m torch.distributed.run ... bench.py
--gpus N` (WORLD_SIZE is then set) or, when bench.py is started alone with --gpus N > 1, from this
script itself: before anything touches the GPU it starts that same launcher as a child process,
relays the child's output and leaves with its exit code.  A run whose world size differs from
--gpus ends with a non-zero status instead of a line that claims another N.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
NTT_ROOFLINE_GELEM = HBM_PEAK_GBS / 64.0   # 64 algorithmic bytes per element per transform
# The roof that binds the transforms in practice: 256-bit modular products per second.  Measured on MI355X with
# nothing else in the kernel (tools/ubench_shoup.hip, the library's mont29_x2 at the transform kernels' occupancy of
# 4 waves per SIMD; profiles/r02_ubench_products.txt): 150 G products/s.  The bound from multiply-add issue alone
# (153 v_mad_u64_u32 per product at the 31.4 T lane-ops/s of tools/ubench_valu.hip) would be 205 G products/s.
VALU_PEAK_GPRODUCTS = 150.0
VALU_MAD_ISSUE_BOUND_GPRODUCTS = 205.0
# The same routine alone at 8 waves per SIMD (163.9; its precomputed-quotient form 178.6).  The in-tile steps do not
# turn that occupancy into throughput: one stage per LDS round trip at 64 registers reaches 155.5 G products/s against
# 154.9 for the library's radix-4 steps at 4 waves (tools/ubench_occupancy.hip, profiles/r03_ubench_occupancy.txt).
VALU_PEAK_8_WAVES_GPRODUCTS = 163.9
VALU_PEAK_8_WAVES_SHOUP_GPRODUCTS = 178.6


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=None,
                    help="log2 of the number of rows (default: 20; 24 for --workload ntt and 18 for --workload batch, the sizes "
                         "of the sharded-transform and batched-proof configurations)")
    ap.add_argument("--blowup", type=int, default=4, help="evaluation domain = blowup * n")
    ap.add_argument("--workload", default="round23", choices=["round23", "ntt", "node-ntt", "node-round23", "round23-sharded", "batch", "tables", "quotient", "msm"],
                    help="round23: the headline metric (default). ntt: ONE four-step transform of 2^log-n points sharded over "
                         "the GPUs (config 4 with --log-n 24), one process per GPU over torch.distributed. node-ntt: the same transform "
                         "through the library's native node layer (plk_node_ntt: ONE process, --gpus ranks, RCCL all-to-all). round23-sharded: ONE proof over all GPUs. "
                         "batch: --circuits independent circuits dealt round-robin (config 5 with --log-n 18 --circuits 64)")
    ap.add_argument("--circuits", type=int, default=64)
    ap.add_argument("--field", type=int, default=0, choices=[0, 1],
                    help="0: BLS12-381 Fr (the headline). 1: the second instantiation, Tweedle Fr (round23 workload only; not the headline, "
                         "no CPU comparison unless --cpu-sample-log-n is reachable)")
    ap.add_argument("--cpu-sample-log-n", type=int, default=20,
                    help="rows of the CPU-baseline run (log2); below --log-n the time is scaled by the row count")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-four-step", action="store_true", help="multi-GPU runs: leave the sharded 2^24-point transform out")
    ap.add_argument("--four-step-limit", type=int, default=120, help="seconds the sharded transform may take before the line goes out without it")
    ap.add_argument("--no-profile", action="store_true", help="time the steps without the per-launch events (no roofline object)")
    ap.add_argument("--no-lean-variant", action="store_true",
                    help="skip the extra figure without the Int/Bool selector columns (a profiler run wants one launch shape per kernel)")
    args = ap.parse_args()
    if args.log_n is None:
        args.log_n = {"ntt": 24, "node-ntt": 24, "batch": 18}.get(args.workload, 20)
    return args


def cpu_baseline(log_n_full: int, blowup: int, sample_log_n: int, field: int = 0):
    """The C restatement of the reference algorithm (oracle/plonk_oracle.c) timed on this host on circuit 0 of
    2^sample_log_n rows (the whole workload when that is log_n_full, else scaled linearly in the number of rows).
    Returns (the bench line's object, the circuit, the oracle's (z, gate, perm)): the caller checks the GPU's
    outputs for that circuit against them."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle as co
    from plonk_amd import synth
    ns = 1 << sample_log_n
    c = synth.field(field).make_circuit(ns, circuit_id=0)
    with co.use_field(field):
        threads = min(co.num_threads(), 16)
        co.set_num_threads(threads)
        t0 = time.perf_counter()
        ref = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), blowup * ns, c.beta, c.gamma)
        dt = time.perf_counter() - t0
    scale = float(1 << (log_n_full - sample_log_n))
    scaled = f"; scaled x{int(scale)} by row count (ignores the log factor of the transforms)" if scale != 1 else "; the full workload, unscaled"
    return {
        "value": round(dt * 1e3 * scale, 1), "unit": "ms", "cores": threads, "kind": "port",
        "sample": f"one round 2+3 at n=2^{sample_log_n}, N=2^{sample_log_n + blowup.bit_length() - 1} took {dt:.2f} s "
                  f"(transforms on {threads} OpenMP threads, grand-product and constraint loops single-threaded as in the "
                  f"reference)" + scaled,
    }, c, ref


def pmc_record(label: str):
    """Counter figures of one kernel label from the newest committed rocprofv3 --pmc summary, if any:
    (record, file name); record has hbm_bytes_per_launch and the VALU issue figures."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json")), reverse=True):
        try:
            with open(path) as f:
                data = json.load(f)
            if label in data:
                return data[label], os.path.basename(path)
        except Exception:
            continue
    return {}, None


def timed(fn, steps, warmup, dist, device):
    """warm up, then time `steps` calls between barriers; max over ranks; seconds."""
    import torch

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        fn()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def other_workloads(args, prover, dist, rank, world):
    """The multi-GPU shapes that are not the headline metric (run explicitly with --workload)."""
    from plonk_amd import dist as pdist
    from plonk_amd import synth
    from plonk_amd.api import AST_KINDS
    from plonk_amd.device import to_device
    n = 1 << args.log_n
    base = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dtype": "u32", "data": "synthetic", "vs_baseline": None}
    if args.workload == "ntt":
        N = n
        x = to_device(synth.random_elements(99 + rank, N // world), prover.device)
        dt = timed(lambda: pdist.four_step_ntt(x, N, prover), args.steps, args.warmup, dist, prover.device)
        ms = dt * 1e3 / args.steps
        gelem = N / (ms * 1e-3) / 1e9
        # without the last exchange: the result stays as [my k1][k2] blocks (what an order-agnostic consumer takes)
        ms_t = timed(lambda: pdist.four_step_ntt(x, N, prover, natural_output=False), args.steps, 1, dist, prover.device) * 1e3 / args.steps
        direct_ms = None
        if world == 1:
            direct_ms = timed(lambda: prover.ntt_batch([x], N), args.steps, 1, dist, prover.device) * 1e3 / args.steps
        line = dict(base, metric=f"four-step NTT of 2^{args.log_n} points, natural order in and out", value=round(gelem, 3),
                    unit="Gelem/s", ms_per_step=round(ms, 4), higher_is_better=True, scaling="strong",
                    config={"workload": f"one transform of N=2^{args.log_n} sharded contiguously over {world} GPU(s), three "
                                        f"all-to-all exchanges of {(world - 1) * N // (world * world) * 32 / 2**20:.1f} MiB per GPU each"},
                    frac_of_hbm_roofline=round(gelem / NTT_ROOFLINE_GELEM, 4),
                    transposed_output={"ms_per_step": round(ms_t, 4), "gelem_per_s": round(N / (ms_t * 1e-3) / 1e9, 3),
                                       "note": "natural_output=False: two exchanges, result as [my k1][k2] = X[k1 + N1 k2]"},
                    direct_transform_ms=round(direct_ms, 4) if direct_ms is not None else None)
    elif args.workload == "tables":
        # the two table-building steps before the path, from one flattened trace per GPU
        import torch
        tr = synth.make_trace(n, circuit_id=rank)
        ids = [to_device(c, prover.device) for c in tr.node_ids]
        ops = torch.from_numpy(tr.ops).to(prover.device)
        vals = to_device(tr.node_values, prover.device)
        dv = synth.ints_to_mont([0])[0]

        def step():
            prover.gate_table_fill(ops, ids, vals, dv)
            prover.sigma_table(ids)
        step()
        torch.cuda.synchronize()
        prover.profile_begin()
        dt = timed(step, args.steps, 0, dist, prover.device)
        prof = prover.profile_end()
        ms = dt * 1e3 / args.steps
        fill = [r for r in prof if r["label"] == "gate_table_fill_kernel"][0]
        fill_ms = fill["total_ms"] / fill["launches"]
        fill_gbs = fill["alg_bytes"] / fill["launches"] / (fill_ms * 1e-3) / 1e9
        sig_ms = sum(r["total_ms"] for r in prof if r["label"].startswith("sig_")) / args.steps
        line = dict(base, metric=f"gate table + sigma table of n=2^{args.log_n} rows", value=round(ms, 4), unit="ms",
                    ms_per_step=round(ms, 4), higher_is_better=False, scaling="weak", dtype="u64", warmup=1,
                    config={"workload": f"flattened trace of 2^{args.log_n} rows per GPU -> 4 wire + 9 selector columns (gather by "
                                        f"node id) and 4 sigma columns (stable radix sort of 2^{args.log_n + 2} (node id, position) "
                                        f"pairs + cycle links)"},
                    roofline={"bound": "hbm", "kernel": "gate_table_fill_kernel", "achieved": round(fill_gbs, 1), "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": round(fill_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                              "avg_launch_ms": round(fill_ms, 4),
                              "note": "577 algorithmic bytes per row (1 + 32 + 128 in, 13 x 32 out)"},
                    sigma_table={"ms": round(sig_ms, 4), "alg_GBps": round(64.0 * n / (sig_ms * 1e-3) / 1e9, 1),
                                 "note": "64 algorithmic bytes per row; a radix sort moves 24 bytes per pair and real pass on top"},
                    kernels=[{"label": r["label"], "launches": r["launches"], "ms_per_step": round(r["total_ms"] / args.steps, 4)}
                             for r in prof])
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import c_oracle as co
            t0 = time.perf_counter()
            co.gate_table_fill(tr.ops, tr.node_ids, tr.node_values, dv)
            co.sigma_table(tr.node_ids)
            line["cpu_baseline"] = {"value": round((time.perf_counter() - t0) * 1e3, 2), "unit": "ms", "cores": 1, "kind": "port",
                                    "sample": f"the same 2^{args.log_n} rows: C restatement, sort-based grouping (the reference "
                                              f"groups with HashMaps)"}
    elif args.workload == "quotient":
        # the step after the path: two numerator vectors on the coset 7 * <omega_N> -> quotient coefficients
        N = args.blowup * n
        nums = [to_device(synth.random_elements(300 + 7 * rank + i, N), prover.device) for i in range(2)]
        alpha, g7 = synth.random_elements(301, 1)[0], synth.ints_to_mont([7])[0]
        out = prover.quotient(nums, alpha, n, g7)
        prover.profile_begin()
        dt = timed(lambda: prover.quotient(nums, alpha, n, g7), args.steps, args.warmup, dist, prover.device)
        prof = prover.profile_end()
        ms = dt * 1e3 / args.steps
        alg = 32.0 * 3 * N                  # two numerators in, N coefficients out
        line = dict(base, metric=f"quotient completion at n=2^{args.log_n}, N={args.blowup}n", value=round(ms, 4), unit="ms",
                    ms_per_step=round(ms, 4), higher_is_better=False, scaling="weak", dtype="u32",
                    config={"workload": f"(num_0 + alpha num_1) / (x^n - 1) on the coset 7<omega_N>, then coset_ifft of N=2^"
                                        f"{args.log_n + args.blowup.bit_length() - 1} points, per GPU"},
                    roofline={"bound": "hbm", "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                              "note": "whole step; 96 algorithmic bytes per point"},
                    kernels=[{"label": r["label"], "launches": r["launches"],
                              "ms_per_step": round(r["total_ms"] / (args.steps + args.warmup), 4)} for r in prof])
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import c_oracle as co
            from plonk_amd.device import to_host
            host = [to_host(x) for x in nums]
            co.set_num_threads(min(co.num_threads(), 16))
            t0 = time.perf_counter()
            ref = co.quotient(host, alpha, n, g7)
            line["cpu_baseline"] = {"value": round((time.perf_counter() - t0) * 1e3, 1), "unit": "ms", "cores": min(co.num_threads(), 16),
                                    "kind": "port", "sample": "the same workload, C restatement, transforms on OpenMP threads"}
            line["matches_cpu"] = bool((to_host(out) == ref).all())
    elif args.workload == "msm":
        # commitment of one polynomial: sum_i coeff_i * base_i in BLS12-381 G1, bases (i + 1) * G resident on the GPU
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import g1_oracle as g1
        gen = np.array(g1.fq_to_mont_limbs(g1.GX) + g1.fq_to_mont_limbs(g1.GY), dtype=np.uint64)
        bases = prover.g1_multiples(gen, 1, n)
        prover.msm_set_bases(bases)
        s_host = synth.random_elements(500 + rank, n)
        s_dev = to_device(s_host, prover.device)
        out = prover.msm(s_dev)
        prover.profile_begin()
        dt = timed(lambda: prover.msm(s_dev), args.steps, args.warmup, dist, prover.device)
        prof = prover.profile_end()
        ms = dt * 1e3 / args.steps
        # four commitments at once (the wire polynomials of one round), overlapped on two streams
        four = [s_dev] + [to_device(synth.random_elements(600 + 10 * rank + k, n), prover.device) for k in range(3)]
        assert (prover.msm_batch(four)[0] == out).all()
        ms_each_of_four = timed(lambda: prover.msm_batch(four), max(2, args.steps // 2), 1, dist, prover.device) * 1e3 / max(2, args.steps // 2) / 4
        alg = (32.0 + 96.0) * n                     # one scalar and one affine base per term
        line = dict(base, metric=f"G1 multi-scalar multiplication of 2^{args.log_n} terms (one KZG commitment)", value=round(ms, 4),
                    unit="ms", ms_per_step=round(ms, 4), higher_is_better=False, scaling="weak", dtype="u32",
                    config={"workload": f"sum of 2^{args.log_n} scalar * point products in BLS12-381 G1 per GPU, bases resident, "
                                        f"result (affine) on the host"},
                    terms_per_s=round(n / (ms * 1e-3), 1), ms_each_in_a_batch_of_4=round(ms_each_of_four, 4),
                    roofline={"bound": "hbm", "achieved": round(alg / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                              "note": "whole step; 128 algorithmic bytes per term; the work is ~180 base-field products per "
                                      "term (one mixed addition per term and window over a precomputed table), i.e. bound "
                                      "by integer multiply-add issue"},
                    kernels=[{"label": r["label"], "launches": r["launches"],
                              "ms_per_step": round(r["total_ms"] / (args.steps + args.warmup), 4)} for r in prof])
        if world == 1 and not args.no_cpu_baseline:
            import c_oracle as co
            from plonk_amd.device import to_host
            co.set_num_threads(min(co.num_threads(), 16))
            b_host = to_host(bases)
            t0 = time.perf_counter()
            ref = co.g1_msm(s_host, b_host)
            line["cpu_baseline"] = {"value": round((time.perf_counter() - t0) * 1e3, 1), "unit": "ms", "cores": min(co.num_threads(), 16),
                                    "kind": "port", "sample": "the same terms, C bucket method, windows on OpenMP threads"}
            line["matches_cpu"] = bool((out == ref).all())
    elif args.workload == "round23-sharded":
        N = args.blowup * n
        c = synth.make_circuit(n, circuit_id=0)
        dc = prover.upload(c)
        sels = [dc.selectors.get(k) for k in AST_KINDS]
        dt = timed(lambda: pdist.round23_poly_sharded(prover, dc.wires, sels, dc.sigma, c.beta, c.gamma, n, N), args.steps,
                   args.warmup, dist, prover.device)
        prover.status()
        ms = dt * 1e3 / args.steps
        line = dict(base, metric=f"prover round 2+3 time at n=2^{args.log_n}, one proof over all GPUs", value=round(ms, 4), unit="ms",
                    ms_per_step=round(ms, 4), higher_is_better=False, scaling="strong",
                    config={"workload": f"one circuit, columns dealt to {world} GPU(s) for interpolation/extension, one all-to-all, "
                                        f"constraints on N/{world} points per GPU"})
    else:  # batch
        N = args.blowup * n
        ids = pdist.shard_circuits(args.circuits)
        circuits = [prover.upload(synth.make_circuit(n, circuit_id=i)) for i in ids]
        out = (prover.empty(n + 1), prover.empty(N), prover.empty(N))

        def step():
            for dc in circuits:
                prover.round23(dc, N, out=out)
        dt = timed(step, args.steps, args.warmup, dist, prover.device)
        prover.status()
        ms = dt * 1e3 / args.steps
        line = dict(base, metric=f"proof throughput, {args.circuits} circuits of n=2^{args.log_n} (rounds 2+3)",
                    value=round(args.circuits / (ms * 1e-3), 2), unit="proofs/s", ms_per_step=round(ms, 4), higher_is_better=True,
                    scaling="strong", config={"workload": f"{args.circuits} independent circuits dealt round-robin over {world} GPU(s), "
                                                          f"no collective"})
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def four_step_extra(prover, dist, world, device, log_N=24, steps=5, warmup=2):
    """Not part of `value`: ONE transform of 2^24 points sharded contiguously over the ranks (four-step form, three
    all-to-all exchanges over RCCL), timed like the main region, plus a forward-then-inverse round trip that must give
    the input back bit for bit - which checks the exchanges and their ordering against the library's kernels on
    whatever links the ranks really have.  Every rank calls this; the dictionary is the same on all of them."""
    import torch
    from plonk_amd import dist as pdist
    N = 1 << log_N
    m = N // world
    g = torch.Generator(device=device)
    g.manual_seed(1234 + (dist.get_rank() if dist is not None else 0))
    # canonical elements: three random low limbs and a top limb below 2^62 (value < 2^254 < p)
    x = torch.randint(-(1 << 63), (1 << 63) - 1, (m, 4), dtype=torch.int64, device=device, generator=g)
    x[:, 3] = x[:, 3] & ((1 << 62) - 1)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_ms(fn):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt * 1e3 / steps

    y = pdist.four_step_ntt(x, N, prover)
    back = pdist.four_step_ntt(y, N, prover, inverse=True)
    ok = torch.tensor([1 if torch.equal(back, x) else 0], dtype=torch.int64, device=device)
    if dist is not None:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    ms = timed_ms(lambda: pdist.four_step_ntt(x, N, prover))
    ms_t = timed_ms(lambda: pdist.four_step_ntt(x, N, prover, natural_output=False))
    return {"what": f"one transform of N=2^{log_N} points sharded contiguously over {world} GPU(s), natural order in and out: "
                    f"three all-to-all exchanges of {(world - 1) * N // (world * world) * 32 / 2**20:.1f} MiB per GPU each",
            "ms": round(ms, 4), "gelem_per_s": round(N / (ms * 1e-3) / 1e9, 3),
            "frac_of_hbm_roofline": round(N / (ms * 1e-3) / 1e9 / NTT_ROOFLINE_GELEM, 4),
            "transposed_output_ms": round(ms_t, 4), "round_trip_bit_exact": bool(int(ok.item())), "steps": steps, "scaling": "strong"}


def node_ntt(args) -> int:
    """--workload node-ntt: ONE transform of 2^log-n points over --gpus ranks held by THIS process (plk_node_ntt: a host
    thread and a stream per rank, fused four-step phases, three all-to-all exchanges over RCCL; with
    PLONK_BENCH_SHARE_GPU=1 the ranks share GPU 0 and exchange by device-to-device copies)."""
    import torch
    from plonk_amd.node import Node
    G, N = args.gpus, 1 << args.log_n
    share = os.environ.get("PLONK_BENCH_SHARE_GPU") == "1"
    devices = [0] * G if share else list(range(G))
    exchange = os.environ.get("PLONK_BENCH_NODE_EXCHANGE", "copy" if share else "rccl")
    node = Node(devices, exchange=exchange)
    m = N // G
    xs = []
    for g in range(G):
        gen = torch.Generator(device=node.device(g))
        gen.manual_seed(4321 + g)
        x = torch.randint(-(1 << 63), (1 << 63) - 1, (m, 4), dtype=torch.int64, device=node.device(g), generator=gen)
        x[:, 3] = x[:, 3] & ((1 << 62) - 1)          # canonical: below 2^254 < p
        xs.append(x)
    ys = node.ntt(xs, N)
    back = node.ntt(ys, N, inverse=True)
    node.sync()
    ok = all(torch.equal(a, b) for a, b in zip(back, xs))
    out = [torch.empty_like(x) for x in xs]

    def timed_ms(natural):
        for _ in range(args.warmup):
            node.ntt(xs, N, natural_output=natural, out=out)
        node.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            node.ntt(xs, N, natural_output=natural, out=out)
        node.sync()
        return (time.perf_counter() - t0) * 1e3 / args.steps

    ms, ms_t = timed_ms(True), timed_ms(False)
    gelem = N / (ms * 1e-3) / 1e9
    line = {"metric": f"four-step NTT of 2^{args.log_n} points, natural order in and out (native node layer)", "value": round(gelem, 3),
            "unit": "Gelem/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"one transform of N=2^{args.log_n} sharded contiguously over {G} rank(s) in one process "
                                   f"(plk_node_ntt, exchange: {exchange}, devices {devices}), three all-to-all exchanges of "
                                   f"{(G - 1) * N // (G * G) * 32 / 2**20:.1f} MiB per rank each"},
            "frac_of_hbm_roofline": round(gelem / NTT_ROOFLINE_GELEM, 4), "transposed_output_ms": round(ms_t, 4),
            "round_trip_bit_exact": bool(ok)}
    node.close()
    print(json.dumps(line), flush=True)
    return 0 if ok else 6


def node_round23(args) -> int:
    """--workload node-round23: rounds 2+3 of ONE circuit of 2^log-n rows over --gpus ranks held by THIS process
    (plk_node_round23: the 18 columns dealt to the ranks, one all-to-all, the constraints on N / G points per rank), its
    pieces compared in full with the single-GPU call on GPU 0."""
    import torch
    from plonk_amd import synth
    from plonk_amd.device import DeviceCircuit, DeviceProver
    from plonk_amd.node import Node
    G, n = args.gpus, 1 << args.log_n
    N = 4 * n
    share = os.environ.get("PLONK_BENCH_SHARE_GPU") == "1"
    devices = [0] * G if share else list(range(G))
    exchange = os.environ.get("PLONK_BENCH_NODE_EXCHANGE", "copy" if share else "rccl")
    prover = DeviceProver(0)
    dc0 = prover.upload(synth.make_circuit(n, circuit_id=0))
    node = Node(devices, exchange=exchange)

    def on(dc, dev):
        return DeviceCircuit(dc.n, [w.to(dev).clone() for w in dc.wires], {k: v.to(dev).clone() for k, v in dc.selectors.items()},
                             [t.to(dev).clone() for t in dc.sigma], dc.beta, dc.gamma)

    copies = [on(dc0, node.device(g)) for g in range(G)]
    outs = [tuple(torch.empty((rows, 4), dtype=torch.int64, device=node.device(g)) for rows in (n + 1, N // G, N // G)) for g in range(G)]
    node.round23(copies, N, outs)
    z, gate, perm = prover.round23(dc0, N)
    prover.status()
    ok = all(torch.equal(o[0].to(prover.device), z) for o in outs) \
        and torch.equal(torch.cat([o[1].to(prover.device) for o in outs]), gate) \
        and torch.equal(torch.cat([o[2].to(prover.device) for o in outs]), perm)
    for _ in range(args.warmup):
        node.round23(copies, N, outs)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        node.round23(copies, N, outs)           # returns when every rank is done
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    slots = -(-18 // G)
    line = {"metric": f"prover round 2+3 time at n=2^{args.log_n} (one proof over the node's ranks)", "value": round(ms, 4), "unit": "ms", "n_gpus": G,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": False, "scaling": "strong",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"rounds 2+3 of ONE circuit, n=2^{args.log_n}, N=4n, its 18 columns dealt to {G} rank(s) in one process "
                                   f"(plk_node_round23, exchange: {exchange}, devices {devices}); one all-to-all of "
                                   f"{(G - 1) * slots * (N // G + 1) * 32 / 2**20:.1f} MiB sent per rank; Z computed on every rank"},
            "matches_single_gpu": bool(ok)}
    node.close()
    print(json.dumps(line), flush=True)
    return 0 if ok else 6


def node_round23_extra(world: int, limit: int):
    """Extra figure of a multi-GPU run, outside `value`: ONE proof over all the GPUs through the native node layer, in a
    child process of rank 0 like `node_ntt_extra`."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                             "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "node-round23", "--gpus", str(world), "--steps", "5", "--warmup", "2"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=limit)
    except subprocess.TimeoutExpired:
        return {"error": f"did not finish within {limit} s"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        return {"error": f"exit status {r.returncode}: " + (r.stderr.strip().splitlines() or ["no output"])[-1][:300]}
    d = json.loads(lines[-1])
    return {"what": d["config"]["workload"], "ms": d["ms_per_step"], "matches_single_gpu": d["matches_single_gpu"]}


def node_ntt_extra(world: int, limit: int):
    """Extra figure of a multi-GPU run, outside `value`: the sharded 2^24-point transform through the library's native
    node layer (one process driving all the GPUs, RCCL bound by the library itself), run as a child process of rank 0
    while the ranks of the main run wait.  Returns the child's line, or an object with an `error`."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                             "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "node-ntt", "--gpus", str(world), "--steps", "5", "--warmup", "2"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=limit)
    except subprocess.TimeoutExpired:
        return {"error": f"did not finish within {limit} s"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"exit status {r.returncode}: " + (r.stderr.strip().splitlines() or ["no output"])[-1][:300]}
    d = json.loads(lines[-1])
    return {"what": d["config"]["workload"], "ms": d["ms_per_step"], "gelem_per_s": d["value"], "frac_of_hbm_roofline": d["frac_of_hbm_roofline"],
            "transposed_output_ms": d["transposed_output_ms"], "round_trip_bit_exact": d["round_trip_bit_exact"]}


def launch_ranks(args) -> int:
    """bench.py started alone with --gpus N > 1: start N ranks of this same command line through
    torch.distributed.run as a CHILD process (this process has not touched the GPU and never will), pass the child's
    stdout and stderr through, and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:               # a port nobody listens on right now
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.workload == "node-ntt":
        sys.exit(node_ntt(args))              # one process holds every rank
    if args.workload == "node-round23":
        sys.exit(node_round23(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    import torch
    from plonk_amd import synth
    from plonk_amd.device import DeviceProver

    if os.environ.get("PLONK_BENCH_WATCHDOG"):   # seconds; dumps every thread's stack and exits if the run is still going
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PLONK_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # the line would claim a number of GPUs that did not run
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr, flush=True)
        sys.exit(2)
    dist = None
    # Rehearsal on a box with fewer GPUs than ranks: PLONK_BENCH_SHARE_GPU=1 maps ranks onto the GPUs
    # present and PLONK_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).
    if os.environ.get("PLONK_BENCH_SHARE_GPU") == "1":
        local_rank %= max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("PLONK_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n, N = 1 << args.log_n, args.blowup << args.log_n

    prover = DeviceProver(local_rank, field=args.field)
    if args.workload != "round23":
        if args.field != 0:
            sys.exit("bench.py: --field applies to the round23 workload only")
        return other_workloads(args, prover, dist, rank, world)
    circuit = synth.field(args.field).make_circuit(n, circuit_id=rank)
    dc = prover.upload(circuit)
    out = (prover.empty(n + 1), prover.empty(N), prover.empty(N))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        prover.round23(dc, N, out=out)
    prover.status()  # also the first correctness gate: no zero denominator, sigma in range

    barrier()
    if not args.no_profile:
        prover.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        prover.round23(dc, N, out=out)
    barrier()
    dt = time.perf_counter() - t0
    prof = prover.profile_end() if not args.no_profile else []
    prover.status()

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=prover.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt * 1e3 / args.steps

    # Not part of `value`: the same call with the Int and Bool selector columns left out (a null selector is an absent
    # one).  The reference transforms them and then multiplies their evaluations by F::zero()
    # (src/execution_trace.rs:396-397), so the three outputs cannot depend on them.
    lean = None
    if world == 1 and not args.no_lean_variant and {"Int", "Bool"} <= set(dc.selectors):
        from plonk_amd.device import DeviceCircuit
        dc2 = DeviceCircuit(dc.n, dc.wires, {k: v for k, v in dc.selectors.items() if k not in ("Int", "Bool")}, dc.sigma, dc.beta, dc.gamma)
        out2 = (prover.empty(n + 1), prover.empty(N), prover.empty(N))
        for _ in range(max(1, args.warmup)):
            prover.round23(dc2, N, out=out2)
        same = all(bool(torch.equal(a, b)) for a, b in zip(out, out2))
        steps2 = max(3, args.steps // 2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps2):
            prover.round23(dc2, N, out=out2)
        torch.cuda.synchronize()
        lean = {"ms_per_step": round((time.perf_counter() - t1) * 1e3 / steps2, 4), "outputs_equal": same,
                "note": "Int and Bool selector columns passed as absent: 16 interpolations + 16 extensions; not the headline value"}

    if rank == 0:
        S = len(circuit.table.selectors)
        ncol = 4 + S + 5
        # dominant kernel label by device time
        dom = max(prof, key=lambda r: r["total_ms"]) if prof else None
        roofline = None
        if dom and dom["total_ms"] > 0:
            per_launch_ms = dom["total_ms"] / dom["launches"]
            per_launch_bytes = dom["alg_bytes"] / dom["launches"]
            achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
            pmc, src = pmc_record(dom["label"])
            roofline = {"bound": "hbm", "kernel": dom["label"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc.get("hbm_bytes_per_launch"),
                        "traffic_source": src, "avg_launch_ms": round(per_launch_ms, 4),
                        "alg_bytes_per_launch": per_launch_bytes,
                        "valu_cycles_per_inst_per_simd": pmc.get("cycles_per_valu_inst_per_simd"),
                        "note": "a transform of N points over B columns is 64*N*B algorithmic bytes (160*n*B for an n -> 4n "
                                "extension); each of its P launches is credited 1/P of that. The nominal bound is HBM, but the "
                                "256-bit modular products keep the vector ALUs issuing back to back (see "
                                "valu_cycles_per_inst_per_simd against the ~4.2 cycles this instruction mix costs): the kernel "
                                "is bound by integer multiply-add issue"}
        roofline_valu = None
        if dom and dom.get("products", 0) > 0:
            gp = dom["products"] / dom["launches"] / (per_launch_ms * 1e-3) / 1e9
            all_products = sum(r.get("products", 0.0) for r in prof) / args.steps
            roofline_valu = {"bound": "valu", "kernel": dom["label"], "achieved": round(gp, 1), "peak": VALU_PEAK_GPRODUCTS,
                             "unit": "G products/s", "frac": round(gp / VALU_PEAK_GPRODUCTS, 4),
                             "products_per_launch": dom["products"] / dom["launches"],
                             "mad_issue_bound": VALU_MAD_ISSUE_BOUND_GPRODUCTS,
                             "frac_of_mad_issue_bound": round(gp / VALU_MAD_ISSUE_BOUND_GPRODUCTS, 4),
                             "peak_8_waves": VALU_PEAK_8_WAVES_GPRODUCTS, "frac_of_8_wave_roof": round(gp / VALU_PEAK_8_WAVES_GPRODUCTS, 4),
                             "peak_8_waves_precomputed_quotient": VALU_PEAK_8_WAVES_SHOUP_GPRODUCTS,
                             "frac_of_8_wave_precomputed_quotient_roof": round(gp / VALU_PEAK_8_WAVES_SHOUP_GPRODUCTS, 4),
                             "step_products": all_products,
                             "step_products_per_s_G": round(all_products / (ms_per_step * 1e-3) / 1e9, 1),
                             "note": "256-bit modular products of the launch (counted from its shape: in-tile butterflies, "
                                     "inter-pass twiddle, coset prescale, u64 conversion) over its measured duration; peak = the "
                                     "same product routine alone in a kernel at the same occupancy, 4 waves per SIMD "
                                     "(tools/ubench_shoup.hip); the 8-wave figures are the routine alone at that occupancy, which "
                                     "the in-tile steps measured at it do not reach (tools/ubench_occupancy.hip: 155.5 against "
                                     "154.9 G products/s). This, not HBM, is the roof the transform kernels sit under"}
        # whole-transform throughput of the extension (all launches of the N-point forward transform)
        lde = [r for r in prof if f"/ntt{args.log_n + args.blowup.bit_length() - 1}" in r["label"] or "/lde" in r["label"]]
        ntt = None
        if lde:
            lde_ms = sum(r["total_ms"] for r in lde) / args.steps
            gelem = ncol * N / (lde_ms * 1e-3) / 1e9
            ntt = {"what": f"{ncol} extensions, {n} coefficients -> {N} points (as {args.blowup - 1} coset transforms of {n} points "
                           f"each; coset 0 is the input table), {len(lde)} launches per step over two streams",
                   "ms_per_batch": round(lde_ms, 4), "gelem_per_s": round(gelem, 3),
                   "frac_of_hbm_roofline": round(gelem / NTT_ROOFLINE_GELEM, 4), "passes": len(lde),
                   # the same time over the points that were really transformed (coset 0 costs nothing)
                   "gelem_per_s_transformed_points": round(gelem * (args.blowup - 1) / args.blowup, 3),
                   "frac_of_hbm_roofline_transformed_points": round(gelem * (args.blowup - 1) / args.blowup / NTT_ROOFLINE_GELEM, 4)}
        alg_bytes_step = (18 * 64 + 18 * 160 + 320 + 448 * args.blowup + 320 * args.blowup) * n if S == 9 else None
        line = {
            "metric": f"prover round 2+3 time at n=2^{args.log_n}", "value": round(ms_per_step / world, 4), "unit": "ms",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"round 2+3, n=2^{args.log_n}, N={args.blowup}n, S={S} selectors, 18 interpolations + "
                                   f"18 extensions (wires evaluated once), one circuit per GPU per step",
                       "field": ("BLS12-381 Fr", "Tweedle Fr (second instantiation: constants self-derived, not the headline field)")[args.field],
                       "circuits_per_step": world},
            "host_call_note": "inputs and outputs resident in HBM; the reference-shaped call on host vectors (plk_round23) moves 772 MB "
                              "over PCIe per call and takes 15.7 ms page-locked / 26 ms pageable at this size",
            "proofs_per_s": round(world / (ms_per_step * 1e-3), 3),
            "step_alg_bytes": alg_bytes_step,
            "step_frac_of_hbm_roofline": round(alg_bytes_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if alg_bytes_step else None,
            "roofline": roofline, "roofline_valu": roofline_valu, "ntt": ntt, "without_int_bool_selectors": lean,
            "kernels": [{"label": r["label"], "launches": r["launches"], "ms_per_step": round(r["total_ms"] / args.steps, 4),
                         "alg_GBps": round(r["alg_bytes"] / max(r["total_ms"], 1e-9) / 1e6, 1),
                         "Gproducts_per_s": round(r.get("products", 0.0) / max(r["total_ms"], 1e-9) / 1e6, 1)} for r in prof],
        }
        if world == 1 and not args.no_cpu_baseline:
            # the CPU restatement on circuit 0, timed; and the GPU's three vectors for that same circuit compared with
            # its results IN FULL (the timed run's own outputs when the sample is the whole workload)
            from plonk_amd.device import to_host
            sample_log_n = min(args.cpu_sample_log_n, args.log_n)
            line["cpu_baseline"], c_ref, ref = cpu_baseline(args.log_n, args.blowup, sample_log_n, args.field)
            if sample_log_n == args.log_n:
                got = out
            else:
                got = prover.round23(prover.upload(c_ref), args.blowup << sample_log_n)
                prover.status()
            import numpy as np
            same = [bool(np.array_equal(to_host(g), r)) for g, r in zip(got, ref)]
            line["matches_cpu"] = all(same)
            line["matches_cpu_detail"] = {"z": same[0], "gate_constraints": same[1], "permutation_constraints": same[2],
                                          "rows": 1 << sample_log_n, "points": args.blowup << sample_log_n,
                                          "what": "every element of the three output vectors of "
                                                  + ("the timed run" if sample_log_n == args.log_n else "a run on the sample circuit")
                                                  + " against the C restatement of the reference"}
    else:
        line = None

    # Extra figure of a multi-GPU run, after the timed region and outside `value`: the sharded 2^24-point transform,
    # the one shape of this path with a real exchange step.  A watchdog bounds it: if the exchanges do not finish in
    # time the line goes out without the figure and the ranks leave, so the headline measurement cannot be lost to it.
    if world > 1 and not args.no_four_step:
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(args.four_step_limit):
                # a stuck exchange: the headline line still goes out, and every rank leaves with a status that says so
                print(f"bench.py: rank {rank} of {world}: the sharded 2^24-point transform (three all-to-all exchanges) did not "
                      f"finish within {args.four_step_limit} s; pending since the timed region ended", file=sys.stderr, flush=True)
                if rank == 0:
                    line["four_step"] = {"error": f"did not finish within {args.four_step_limit} s"}
                    print(json.dumps(line), flush=True)
                os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            extra = four_step_extra(prover, dist, world, prover.device)
        except Exception as e:   # noqa: BLE001 - whatever the exchange raised, the main figure still goes out ...
            extra = {"error": f"{type(e).__name__}: {e}"[:300]}
            print(f"bench.py: rank {rank} of {world}: the sharded transform failed: {extra['error']}", file=sys.stderr, flush=True)
            if rank == 0:
                line["four_step"] = extra
                print(json.dumps(line), flush=True)
            os._exit(5)          # ... and the status says that the run was not clean (the other ranks leave by their watchdogs)
        done.set()
        if rank == 0:
            line["four_step"] = extra
            if extra.get("round_trip_bit_exact") is False:
                print(json.dumps(line), flush=True)
                print("bench.py: the sharded transform's forward-then-inverse round trip did not return its input", file=sys.stderr, flush=True)
                os._exit(6)
    node_failed = False
    # The other ranks wait for rank 0's child processes on the HOST (a key in the rendezvous store): a collective
    # barrier would sit on their GPUs as a spinning kernel beside the children's work.
    store = dist.distributed_c10d._get_default_store() if (dist is not None and world > 1 and not args.no_four_step) else None
    if store is not None and rank != 0:
        import datetime
        try:
            store.wait(["plonk_bench_children_done"], datetime.timedelta(seconds=2 * args.four_step_limit + 60))
        except Exception:   # noqa: BLE001 - rank 0 reports what happened to its children; this rank only stops waiting
            pass
    if world > 1 and not args.no_four_step and rank == 0:
        # the same transform once more through the native node layer (one child process over all the GPUs)
        line["node_ntt"] = node_ntt_extra(world, args.four_step_limit)
        node_failed = "error" in line["node_ntt"] or line["node_ntt"].get("round_trip_bit_exact") is False
        if args.workload == "round23" and args.field == 0:
            line["node_round23"] = node_round23_extra(world, args.four_step_limit)
            node_failed = node_failed or "error" in line["node_round23"] or line["node_round23"].get("matches_single_gpu") is False
        if store is not None:
            store.set("plonk_bench_children_done", "1")
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if node_failed:
        print("bench.py: the native node layer failed: " + json.dumps({k: line.get(k) for k in ("node_ntt", "node_round23")}), file=sys.stderr, flush=True)
        sys.exit(7)
    if line is not None and line.get("matches_cpu") is False:
        print("bench.py: the GPU's outputs differ from the CPU restatement: " + json.dumps(line["matches_cpu_detail"]), file=sys.stderr, flush=True)
        sys.exit(4)


if __name__ == "__main__":
    main()
