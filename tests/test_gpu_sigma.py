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
