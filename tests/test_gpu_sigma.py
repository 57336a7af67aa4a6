"""GPU parity of the sigma-table step (plk_sigma_table / plk_dev_sigma_table) against the CPU
restatements and the committed vectors; index work, so the comparison is exact."""
import json
import os

import numpy as np
import pytest

import c_oracle as co
import plonk_amd as pa
import plonk_oracle as po
from plonk_amd import synth
from plonk_amd.device import to_device

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_WIRE = np.uint64(po.NO_WIRE)


def cols_equal(got, want):
    return all(np.array_equal(np.asarray(g, dtype=np.uint64), np.asarray(w, dtype=np.uint64)) for g, w in zip(got, want))


def test_golden_vectors(ctx):
    with open(os.path.join(ROOT, "tests", "golden", "sigma_vectors.json")) as f:
        doc = json.load(f)
    for case in doc["cases"]:
        if "node_ids" in case:
            ids = [np.array(c, dtype=np.uint64) for c in case["node_ids"]]
        else:
            ids = synth.make_node_ids(case["generator"]["n"], case["generator"]["circuit_id"])
        t = pa.build_permutation_table(ids, ctx=ctx)
        assert cols_equal(t.columns(), case["sigma"]), case["name"]


# below, at and just past one wave's chunk (2048 pairs = 512 rows), odd sizes, and the bench size
@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 511, 512, 513, 2047, 2048, 2049, 100003, 1 << 20])
def test_host_entry_vs_oracle(ctx, n):
    ids = synth.make_node_ids(n, circuit_id=n % 7)
    t = pa.build_permutation_table(ids, ctx=ctx)
    assert cols_equal(t.columns(), co.sigma_table(ids))


def test_empty(ctx):
    t = pa.build_permutation_table([np.empty((0,), dtype=np.uint64)] * 4, ctx=ctx)
    assert all(c.shape == (0,) for c in t.columns())


def test_wide_and_degenerate_ids(ctx):
    n = 5000
    rng = np.random.default_rng(5)
    # ids spread over all 64 bits (every sort pass is a real one), drawn from a small pool so groups form
    pool = rng.integers(0, 2 ** 63, size=97, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=97, dtype=np.uint64)
    pool[0], pool[1], pool[2] = np.uint64(2 ** 64 - 2), np.uint64(0), np.uint64(0xFF)
    wide = [pool[rng.integers(0, 97, size=n)] for _ in range(4)]
    wide[2][rng.integers(0, n, size=n // 3)] = NO_WIRE
    for ids in (wide,
                [np.full((n,), 7, dtype=np.uint64)] * 4,                       # one node on every position: one 4n-cycle
                [np.full((n,), NO_WIRE, dtype=np.uint64)] * 4,                 # nothing used
                [np.arange(w * n, (w + 1) * n, dtype=np.uint64) for w in range(4)]):   # every node once
        t = pa.build_permutation_table(ids, ctx=ctx)
        assert cols_equal(t.columns(), co.sigma_table(ids))


def test_device_entry_and_reuse(prover):
    """device-resident call, twice on one context with a smaller second input (workspace reuse)."""
    for n in (1 << 16, 3000):
        ids = synth.make_node_ids(n, circuit_id=2)
        dev = [to_device(c, prover.device) for c in ids]
        out = prover.sigma_table(dev)
        prover.status()
        assert cols_equal([o.cpu().numpy().view(np.uint64) for o in out], co.sigma_table(ids))


def test_table_feeds_the_grand_product(ctx):
    """the produced table is a valid input of the next step (indices in range, no error)."""
    n = 1 << 12
    c = synth.make_circuit(n)
    t = pa.build_permutation_table(synth.make_node_ids(n), ctx=ctx)
    z = pa.build_grand_product_evaluation_polynomial(t, c.table, c.beta, c.gamma, ctx=ctx)
    want = co.grand_product(c.table.wires(), t.columns(), c.beta, c.gamma)
    assert np.array_equal(z, want)


# ---- gate evaluation table


def test_gate_table_golden(ctx):
    with open(os.path.join(ROOT, "tests", "golden", "sigma_vectors.json")) as f:
        doc = json.load(f)
    for case in doc["gate_table_cases"]:
        kinds = list(case["selectors"])
        t = pa.build_gate_evaluation_table(np.array(case["ops"], dtype=np.uint8), [np.array(c, dtype=np.uint64) for c in case["node_ids"]],
                                           synth.ints_to_mont([int(x, 16) for x in case["node_values"]]),
                                           synth.ints_to_mont([int(case["default_val"], 16)])[0], kinds_present=kinds, ctx=ctx)
        assert [synth.mont_to_ints(c) for c in t.wires()] == [[int(x, 16) for x in c] for c in case["wires"]], case["name"]
        assert {k: synth.mont_to_ints(v) for k, v in t.selectors.items()} == case["selectors"], case["name"]


@pytest.mark.parametrize("n", [1, 3, 255, 256, 257, 100003, 1 << 18])
def test_gate_table_vs_oracle(ctx, n):
    t = synth.make_trace(n, circuit_id=n % 5)
    dv = synth.ints_to_mont([5])[0]
    vals = t.node_values[: max(1, t.node_values.shape[0] - n // 3)]   # the newest nodes have no value: default applies
    got = pa.build_gate_evaluation_table(t.ops, t.node_ids, vals, dv, kinds_present=pa.AST_KINDS, ctx=ctx)
    w, sels = co.gate_table_fill(t.ops, t.node_ids, vals, dv)
    assert cols_equal([c.reshape(-1) for c in got.wires()], [c.reshape(-1) for c in w])
    assert set(got.selectors) == set(sels) == set(pa.AST_KINDS)
    for k in sels:
        assert np.array_equal(got.selectors[k], sels[k]), k


def test_trace_to_constraints_on_device(prover):
    """flattened trace -> gate table + sigma table -> rounds 2+3, all device-resident, against the CPU chain."""
    from plonk_amd.device import DeviceCircuit
    n = 1 << 12
    t = synth.make_trace(n)
    c = synth.make_circuit(n)            # beta, gamma only
    dv = synth.ints_to_mont([0])[0]
    d_ids = [to_device(x, prover.device) for x in t.node_ids]
    import torch
    wires, sels = prover.gate_table_fill(torch.from_numpy(t.ops).to(prover.device), d_ids, to_device(t.node_values, prover.device), dv)
    sigma = prover.sigma_table(d_ids)
    from plonk_amd.device import _host_fe
    dc = DeviceCircuit(n, wires, sels, sigma, _host_fe(c.beta), _host_fe(c.gamma))
    z, g, p = prover.round23(dc, 4 * n)
    prover.status()
    w_ref, s_ref = co.gate_table_fill(t.ops, t.node_ids, t.node_values, dv)
    sig_ref = co.sigma_table(t.node_ids)
    z_ref, g_ref, p_ref = co.round23(w_ref, s_ref, sig_ref, 4 * n, c.beta, c.gamma)
    for got, want in ((z, z_ref), (g, g_ref), (p, p_ref)):
        assert np.array_equal(got.cpu().numpy().view(np.uint64), want)
