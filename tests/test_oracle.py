"""CPU tests: the oracles against the known answers, each other and the committed vectors."""
import random

import numpy as np
import pytest

import c_oracle as co
import plonk_oracle as po
from helpers import case_inputs, decode, encode_like, h2i, oracle_perm, oracle_table
from plonk_amd import synth


def test_field_constants_rederived(vectors):
    f = vectors["field"]
    p = int(f["modulus"], 16)
    assert p == po.P and p.bit_length() == 255
    # two-adicity and root: p - 1 = 2^32 * odd, root = g^((p-1)/2^32) has exact order 2^32
    assert (p - 1) % (1 << 32) == 0 and ((p - 1) >> 32) % 2 == 1
    root = pow(7, (p - 1) >> 32, p)
    assert hex(root) == f["two_adic_root"] == "0x16a2a19edfe81f20d09b681922c813b4b63683508c2280b93829971f439f0d2b"
    assert pow(root, 1 << 31, p) == p - 1
    assert pow(7, (p - 1) // 2, p) == p - 1  # 7 is a non-residue, hence a generator candidate of the 2-Sylow part
    assert int(f["r_mod_p"], 16) == (1 << 256) % p and int(f["r2_mod_p"], 16) == (1 << 512) % p


def test_known_answers(vectors):
    ka = vectors["known_answers"]
    assert hex(po.root_of_unity(4)) == ka["omega_4"] == "0x8d51ccce760304d0ec030002760300000001000000000000"
    assert [hex(x) for x in po.ntt([1, 2, 3, 4], 4)] == ka["ntt_1_2_3_4"]
    assert po.ntt([1, 2, 3, 4], 4)[0] == 10
    assert h2i(ka["intt_of_that"]) == [1, 2, 3, 4]
    assert [hex(l) for l in po.to_mont(1)] == ka["mont_limbs_of_1"]


@pytest.mark.parametrize("n", [1, 2, 4, 8, 32, 128])
def test_fast_transform_is_the_dft(n):
    rng = random.Random(n)
    v = [rng.randrange(po.P) for _ in range(n)]
    assert po.ntt(v, n) == po.dft_naive(v, n)
    assert po.intt(v, n) == po.dft_naive(v, n, inverse=True)
    assert po.intt(po.ntt(v, n), n) == v
    short = v[: max(1, n // 4)]
    assert po.ntt(short, n) == po.dft_naive(short, n)


def test_c_field_ops_match_bigint():
    rng = random.Random(11)
    edge = [0, 1, 2, po.P - 1, po.P - 2, (1 << 255) % po.P, po.R_MOD_P, po.R2_MOD_P]
    vals = edge + [rng.randrange(po.P) for _ in range(200)]
    for a in vals:
        for b in (vals[rng.randrange(len(vals))], vals[rng.randrange(len(vals))]):
            A, B = np.array(po.to_mont(a), dtype=np.uint64), np.array(po.to_mont(b), dtype=np.uint64)
            assert po.from_mont(co.fe_mul(A, B)) == a * b % po.P
            assert po.from_mont(co.fe_add(A, B)) == (a + b) % po.P
            assert po.from_mont(co.fe_sub(A, B)) == (a - b) % po.P
        if a:
            assert po.from_mont(co.fe_inverse(np.array(po.to_mont(a), dtype=np.uint64))) == pow(a, -1, po.P)
    with pytest.raises(ZeroDivisionError):
        co.fe_inverse(np.zeros(4, dtype=np.uint64))
    for x in (0, 1, 5, 2**32, 2**64 - 1):
        assert po.from_mont(co.fe_from_u64(x)) == x


def test_c_transforms_match_bigint():
    rng = random.Random(5)
    for n in (1, 2, 8, 64, 512):
        v = [rng.randrange(po.P) for _ in range(n)]
        m = synth.ints_to_mont(v)
        assert synth.mont_to_ints(co.ntt(m, n)) == po.ntt(v, n)
        assert synth.mont_to_ints(co.intt(m, n)) == po.intt(v, n)
        assert synth.mont_to_ints(co.ntt(m[: max(1, n // 2)], 2 * n)) == po.ntt(v[: max(1, n // 2)], 2 * n)
    with pytest.raises(co.OracleError):
        co.ntt(synth.ints_to_mont([1, 2, 3]), 3)
    with pytest.raises(co.OracleError):
        co.intt(synth.ints_to_mont([1, 2, 3]), 4)


def test_vectors_reproduce_from_both_oracles(vectors):
    """Every committed case: the big-integer oracle regenerates it and the C restatement agrees."""
    for case in vectors["cases"]:
        n, table, perm, beta, gamma = case_inputs(case)
        out = case["outputs"]
        ot, op = oracle_table(table), oracle_perm(perm)
        b, g = int(case["beta"], 16), int(case["gamma"], 16)
        gp = po.gate_evaluation_table_to_gate_polynomial(ot, n)
        wc, sc = co.gate_polys(table.wires(), table.selectors, n)
        for k, ref in enumerate((gp.input1, gp.input2, gp.input3, gp.output)):
            assert decode(case, out["wire_coeffs"][k]) == encode_like(case, synth.ints_to_mont(ref)) == encode_like(case, wc[k])
        for k in table.selectors:
            assert decode(case, out["sel_coeffs"][k]) == encode_like(case, sc[k])
        z = co.grand_product(table.wires(), perm.columns(), beta, gamma)
        assert decode(case, out["z"]) == encode_like(case, z)
        assert encode_like(case, z) == encode_like(case, synth.ints_to_mont(po.build_grand_product_evaluation_polynomial(op, ot, b, g)))
        s_out, z_c, z_last = co.perm_polys(table.wires(), perm.columns(), beta, gamma, n)
        assert decode(case, out["z_coeffs"]) == encode_like(case, z_c)
        assert int(out["z_last"], 16) == synth.mont_to_ints(z_last)[0]
        for k in range(4):
            assert decode(case, out["sigma_coeffs"][k]) == encode_like(case, s_out[k])
        for N in case["domains"]:
            assert decode(case, out["gate_constraints"][str(N)]) == encode_like(case, co.gate_constraints(wc, sc, N))
            assert decode(case, out["perm_constraints"][str(N)]) == encode_like(case, co.perm_constraints(z_c, s_out, wc, N, beta, gamma))
            assert decode(case, out["wire0_evals"][str(N)]) == encode_like(case, co.ntt(wc[0], N))


def test_example_circuit_satisfies_its_gates(vectors):
    """The reference's one built-in circuit, Add(Int 40, Int 2): the gate constraint vector is zero
    on the trace domain."""
    case = next(c for c in vectors["cases"] if c["name"] == "example_add_40_2")
    assert all(int(x, 16) == 0 for x in case["outputs"]["gate_constraints"]["4"])
    assert case["inputs"]["wires"][3][2] == hex(42)


def test_reference_quirks_are_kept():
    """The grand product uses the value at the permuted row and the bare row index, so z[n] != 1 in
    general and the permutation check does not vanish: both must be reproduced, not repaired."""
    c = synth.make_circuit(8)
    z = synth.mont_to_ints(co.grand_product(c.table.wires(), c.perm.columns(), c.beta, c.gamma))
    assert z[0] == 1 and z[-1] != 1
    ot, op = oracle_table(c.table), oracle_perm(c.perm)
    b, g = synth.mont_to_ints(c.beta)[0], synth.mont_to_ints(c.gamma)[0]
    # strict mode reproduces the length-assert panic of build_permutation_polynomials
    with pytest.raises(AssertionError):
        po.build_permutation_polynomials(op, ot, b, g, 8, strict=True)
    # z_shift: the reference shifts by ONE point of the evaluation domain, also on the 4n domain
    pp = po.build_permutation_polynomials(op, ot, b, g, 8)
    gp = po.gate_evaluation_table_to_gate_polynomial(ot, 8)
    e1 = po.build_permutation_constraint_evaluation_polynomial(pp, gp, b, g, 32)
    e4 = po.build_permutation_constraint_evaluation_polynomial(pp, gp, b, g, 32, z_shift=4)
    assert e1 != e4


def test_oracle_error_paths():
    c = synth.make_circuit(8)
    sig = [s.copy() for s in c.perm.columns()]
    sig[2][5] = 8  # out of range
    with pytest.raises(co.OracleError) as ei:
        co.grand_product(c.table.wires(), sig, c.beta, c.gamma)
    assert ei.value.code == -5 and ei.value.row == 5
    # zero denominator: make v1[3] = -(beta*3 + gamma)
    b, g = synth.mont_to_ints(c.beta)[0], synth.mont_to_ints(c.gamma)[0]
    wires = [w.copy() for w in c.table.wires()]
    wires[1][3] = synth.ints_to_mont([(-(b * 3 + g)) % po.P])[0]
    with pytest.raises(co.OracleError) as ei:
        co.grand_product(wires, c.perm.columns(), c.beta, c.gamma)
    assert ei.value.code == -4 and ei.value.row == 3
