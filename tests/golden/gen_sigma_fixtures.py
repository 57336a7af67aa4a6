#!/usr/bin/env python3
"""Generates tests/golden/sigma_vectors.json: permutation (sigma) tables computed by the
reference-shaped restatement in oracle/plonk_oracle.py (operation tables keyed by kind, WirePosition
keys, offset table), together with the flattened node-id columns the C ABI takes.

Run from the repository root:  python3 tests/golden/gen_sigma_fixtures.py
The reference has no fixtures for this step (and cannot be built here); these vectors pin the two
restatements and the HIP path to each other, not to the Rust code.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import plonk_oracle as po  # noqa: E402
from plonk_amd import synth  # noqa: E402


def structured_case(name, trace):
    offsets = po.offset_table_build(trace)
    table = po.build_permutation_table(trace, offsets, po.build_permutation_trace(trace))
    return {
        "name": name,
        "trace": {"single_input": {k: [[list(i), o] for i, o in rows] for k, rows in trace.single_input.items()},
                  "double_input": {k: [[list(i), o] for i, o in rows] for k, rows in trace.double_input.items()},
                  "triple_input": {k: [[list(i), o] for i, o in rows] for k, rows in trace.triple_input.items()}},
        "node_ids": po.flatten_trace(trace),
        "sigma": [table.sigma1, table.sigma2, table.sigma3, table.sigma_output],
    }


def gate_table_case(name, trace, node_values, default_val):
    """build_gate_evaluation_table (completed pattern) on a structured trace + its flattened inputs."""
    offsets = po.offset_table_build(trace)
    t = po.build_gate_evaluation_table(trace, offsets, po.wire_map_from_node_values(trace, node_values), default_val)
    hx = lambda v: [hex(x) for x in v]
    return {"name": name, "ops": po.flatten_ops(trace), "node_ids": po.flatten_trace(trace), "node_values": hx(node_values),
            "default_val": hex(default_val), "wires": [hx(c) for c in (t.input1, t.input2, t.input3, t.output)],
            "selectors": {k: v for k, v in t.selectors.items()}}


def flat_case(name, node_ids, generator=None):
    t = po.sigma_table_from_node_ids(node_ids)
    case = {"name": name, "sigma": [t.sigma1, t.sigma2, t.sigma3, t.sigma_output]}
    if generator:
        case["generator"] = generator
    else:
        case["node_ids"] = node_ids
    return case


def main():
    rng = random.Random(0x51C3A)
    cases = []
    # Add(Int 40, Int 2): one double-input row; the two literals and the sum are three nodes
    cases.append(structured_case("example_add_40_2", po.ExecutionTrace(double_input={"Add": [((1, 2), 0)]})))
    # x = a + b; y = x * x; z = not y; w = if z then x else y  (shared nodes across kinds and wires)
    cases.append(structured_case("shared_nodes", po.ExecutionTrace(
        single_input={"Not": [((11,), 12)]},
        double_input={"Add": [((1, 2), 10)], "Mult": [((10, 10), 11)]},
        triple_input={"If": [((12, 10, 11), 13)]})))
    for t in range(4):
        nn = rng.randint(2, 9)
        rows = lambda ar, cnt: [(tuple(rng.randrange(nn) for _ in range(ar)), rng.randrange(nn)) for _ in range(cnt)]
        kinds2 = list(po.DOUBLE_INPUT)
        rng.shuffle(kinds2)
        cases.append(structured_case(f"random_{t}", po.ExecutionTrace(
            {"Not": rows(1, rng.randint(0, 5))}, {k: rows(2, rng.randint(0, 4)) for k in kinds2[:3]},
            {"If": rows(3, rng.randint(0, 5))})))
    big = (1 << 64) - 2
    cases.append(flat_case("wide_ids", [[big, 5, 1 << 40, po.NO_WIRE, 0xFF], [0xFF00, big, po.NO_WIRE, 1 << 40, 5],
                                        [po.NO_WIRE] * 5, [0xFF, 0xFF00, 1 << 63, big, 0]]))
    cases.append(flat_case("all_unused", [[po.NO_WIRE] * 3 for _ in range(4)]))
    cases.append(flat_case("one_node", [[7] * 6 for _ in range(4)]))
    for n in (8, 16, 1000):
        cols = [[int(x) for x in c] for c in synth.make_node_ids(n)]
        cases.append(flat_case(f"synth_n{n}", cols, generator={"name": "plonk_amd.synth.make_node_ids", "n": n, "circuit_id": 0}))
    gate_cases = []
    # Add(Int 40, Int 2) with its evaluated values 40, 2 -> 42 (src/ast.rs:473-477)
    gate_cases.append(gate_table_case("example_add_40_2", po.ExecutionTrace(double_input={"Add": [((1, 2), 0)]}), [42, 40, 2], 0))
    shared = po.ExecutionTrace(single_input={"Not": [((11,), 12)]}, double_input={"Add": [((1, 2), 10)], "Mult": [((10, 10), 11)]},
                               triple_input={"If": [((12, 10, 11), 13)]})
    gate_cases.append(gate_table_case("shared_nodes", shared, [rng.randrange(po.P) for _ in range(14)], 7))
    gate_cases.append(gate_table_case("nodes_without_value", shared, [rng.randrange(po.P) for _ in range(12)], po.P - 1))
    path = os.path.join(HERE, "sigma_vectors.json")
    with open(path, "w") as f:
        json.dump({"no_wire": po.NO_WIRE, "cases": cases, "gate_table_cases": gate_cases}, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
