"""Shared helpers for the parity tests (oracle-side conversions, fixture decoding)."""
import hashlib

import numpy as np

import plonk_oracle as po
from plonk_amd import synth
from plonk_amd.api import GateEvaluationTable, PermutationTable


def h2i(v):
    return [int(x, 16) for x in v]


def digest_ints(v):
    h = hashlib.sha256()
    for x in v:
        h.update(int(x).to_bytes(32, "little"))
    return h.hexdigest()


def decode(case, v):
    """fixture value -> comparable object (list of ints for hex cases, digest string otherwise)."""
    return h2i(v) if case["encoding"] == "hex" else v


def encode_like(case, limbs):
    """(n,4) uint64 Montgomery limbs -> the fixture's encoding."""
    ints = synth.mont_to_ints(limbs)
    return ints if case["encoding"] == "hex" else digest_ints(ints)


def case_inputs(case):
    """-> (n, GateEvaluationTable, PermutationTable, beta, gamma) in the host interface's types."""
    n = case["n"]
    if case["encoding"] == "hex":
        inp = case["inputs"]
        wires = [synth.ints_to_mont(h2i(c)) for c in inp["wires"]]
        sels = {k: synth.ints_to_mont(h2i(v)) for k, v in inp["selectors"].items()}
        sig = [np.array(c, dtype=np.uint64) for c in inp["sigma"]]
        table, perm = GateEvaluationTable(*wires, selectors=sels), PermutationTable(*sig)
    else:
        c = synth.make_circuit(inp_n := case["inputs"]["n"], case["inputs"]["circuit_id"])
        assert inp_n == n
        table, perm = c.table, c.perm
    beta = synth.ints_to_mont([int(case["beta"], 16)])[0]
    gamma = synth.ints_to_mont([int(case["gamma"], 16)])[0]
    return n, table, perm, beta, gamma


def oracle_table(table):
    return po.GateEvaluationTable(*[synth.mont_to_ints(w) for w in table.wires()],
                                  selectors={k: synth.mont_to_ints(v) for k, v in table.selectors.items()})


def oracle_perm(perm):
    return po.PermutationTable(*[[int(x) for x in s] for s in perm.columns()])


def horner(coeff_ints, x):
    acc = 0
    for c in reversed(coeff_ints):
        acc = (acc * x + c) % po.P
    return acc
