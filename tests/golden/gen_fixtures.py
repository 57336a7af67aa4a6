#!/usr/bin/env python3
"""Generates tests/golden/*.json with the big-integer oracle (oracle/plonk_oracle.py).

The reference (wu-s-john/PLONK) has no tests, fixtures or runnable binary, so these vectors are
produced by the oracle's restatement of the reference algorithm on seeded inputs; they pin the
C oracle and the HIP path to it, not to the Rust crate ("parity unpinned" upstream).

Values are canonical integers as hex strings.  Inputs come from plonk_amd.synth (splitmix64),
plus the reference's only built-in circuit, Add(Int 40, Int 2) (src/ast.rs:471-488).

    python tests/golden/gen_fixtures.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import plonk_oracle as po  # noqa: E402
from plonk_amd import synth  # noqa: E402


def hx(v):
    return [hex(x) for x in v]


def digest(v):
    h = hashlib.sha256()
    for x in v:
        h.update(int(x).to_bytes(32, "little"))
    return h.hexdigest()


def run_case(name, n, table, perm, beta, gamma, domains, full):
    enc = hx if full else digest
    gp = po.gate_evaluation_table_to_gate_polynomial(table, n)
    z = po.build_grand_product_evaluation_polynomial(perm, table, beta, gamma)
    pp = po.build_permutation_polynomials(perm, table, beta, gamma, n)
    out = {
        "wire_coeffs": [enc(c) for c in (gp.input1, gp.input2, gp.input3, gp.output)],
        "sel_coeffs": {k: enc(v) for k, v in gp.selectors.items()},
        "z": enc(z),
        "sigma_coeffs": [enc(c) for c in (pp.sigma1, pp.sigma2, pp.sigma3, pp.sigma_output)],
        "z_coeffs": enc(pp.z),
        "z_last": hex(pp.z_last),
        "gate_constraints": {}, "perm_constraints": {}, "wire0_evals": {},
    }
    for N in domains:
        out["gate_constraints"][str(N)] = enc(po.build_gate_constraint_evaluation_polynomial(gp, N))
        out["perm_constraints"][str(N)] = enc(po.build_permutation_constraint_evaluation_polynomial(pp, gp, beta, gamma, N))
        out["wire0_evals"][str(N)] = enc(po.ntt(gp.input1, N))
    case = {"name": name, "n": n, "domains": list(domains), "encoding": "hex" if full else "sha256-le32",
            "beta": hex(beta), "gamma": hex(gamma), "outputs": out}
    if full:
        case["inputs"] = {
            "wires": [hx(c) for c in (table.input1, table.input2, table.input3, table.output)],
            "selectors": {k: hx(v) for k, v in table.selectors.items()},
            "sigma": [list(map(int, c)) for c in (perm.sigma1, perm.sigma2, perm.sigma3, perm.sigma_output)],
        }
    else:
        case["inputs"] = {"generator": "plonk_amd.synth.make_circuit", "n": n, "circuit_id": 0}
    return case


def synth_case(n, full):
    c = synth.make_circuit(n)
    table = po.GateEvaluationTable(*[synth.mont_to_ints(w) for w in c.table.wires()],
                                   selectors={k: synth.mont_to_ints(v) for k, v in c.table.selectors.items()})
    perm = po.PermutationTable(*[[int(x) for x in s] for s in c.perm.columns()])
    beta, gamma = synth.mont_to_ints(c.beta)[0], synth.mont_to_ints(c.gamma)[0]
    return run_case(f"synth_n{n}", n, table, perm, beta, gamma, (n, 4 * n), full)


def main():
    cases = []
    n, table, perm = po.example_circuit()
    cases.append(run_case("example_add_40_2", n, table, perm, 5, 7, (4, 16), True))
    for n in (4, 8, 16):
        cases.append(synth_case(n, True))
    cases.append(synth_case(1024, False))
    doc = {
        "field": {"name": "BLS12-381 Fr", "modulus": hex(po.P), "generator": po.GENERATOR, "two_adicity": po.TWO_ADICITY,
                  "two_adic_root": hex(po.TWO_ADIC_ROOT), "r_mod_p": hex(po.R_MOD_P), "r2_mod_p": hex(po.R2_MOD_P)},
        "known_answers": {
            "omega_4": hex(po.root_of_unity(4)),
            "ntt_1_2_3_4": hx(po.ntt([1, 2, 3, 4], 4)),
            "intt_of_that": hx(po.intt(po.ntt([1, 2, 3, 4], 4), 4)),
            "mont_limbs_of_1": [hex(l) for l in po.to_mont(1)],
            "mont_limbs_of_p_minus_1": [hex(l) for l in po.to_mont(po.P - 1)],
        },
        "cases": cases,
    }
    path = os.path.join(HERE, "plonk_vectors.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
