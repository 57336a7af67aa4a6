"""CPU tests of the second field instantiation (the scalar field of Tweedledee, field 1): its constants re-derived
from the modulus, the two restatements of the oracle against each other over it, and the constants the library
exports for it.  PARITY UNPINNED twice over: as for the first field nothing the reference holds pins these results,
and here not even the generator is pinned - 5, the smallest quadratic non-residue, is this build's own choice
(tools/derive_field_constants.py); ark-ff's constants for the field were not available."""
import ctypes as C
import importlib.util
import os

import numpy as np

import c_oracle as co
import plonk_oracle as po
from plonk_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F1 = synth.field(1)


def _derive():
    spec = importlib.util.spec_from_file_location("derive_field_constants", os.path.join(ROOT, "tools", "derive_field_constants.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_constants_rederived():
    d = _derive()
    c = d.derive("TweedleFr", *d.FIELDS["TweedleFr"])
    p = (1 << 254) + 4707489544292117082687961190295928833
    assert c["p"] == p == po.FIELDS[1][0] == synth.FIELD_MODULI[1] and d.is_probable_prime(p)
    assert c["two_adicity"] == 34 == po.FIELDS[1][2] and (p - 1) % (1 << 34) == 0 and (p - 1) % (1 << 35) != 0
    assert c["generator"] == 5 == po.FIELDS[1][1]
    assert all(pow(g, (p - 1) // 2, p) == 1 for g in (2, 3, 4)) and pow(5, (p - 1) // 2, p) == p - 1   # 5 is the smallest non-residue
    with po.use_field(1):
        assert po.TWO_ADIC_ROOT == c["root"] and pow(c["root"], 1 << 33, p) == p - 1
        assert po.root_of_unity(4) == pow(c["root"], 1 << 32, p)
    # the first field through the same script gives the published constants
    b = d.derive("Bls12_381Fr", *d.FIELDS["Bls12_381Fr"])
    assert b["root"] == 0x16A2A19EDFE81F20D09B681922C813B4B63683508C2280B93829971F439F0D2B and b["two_adicity"] == 32
    # what the header of the kernels holds is what the script prints
    text = open(os.path.join(ROOT, "plonk_amd", "csrc", "field.cuh")).read()
    for name in ("Bls12_381Fr", "TweedleFr"):
        cc = d.derive(name, *d.FIELDS[name])
        body = text[text.index(f"struct {name} {{"):]
        body = body[:body.index("\n};")]
        for key, field_name in (("p", "P"), ("one", "ONE"), ("r2", "R2"), ("root", "ROOT")):
            import re
            m = re.search(rf"{field_name}\[8\]\s*=\s*\{{([^}}]*)\}}", body)
            words = [int(w.strip().rstrip("u"), 16) for w in m.group(1).replace("\n", " ").split(",")]
            assert words == d.limbs32(cc[key]), (name, field_name)
        assert f"TWO_ADICITY = {cc['two_adicity']};" in body
        assert float.fromhex(re.search(r"QEST = (0x[0-9a-f.]+p-?\d+)f;", body).group(1)) == cc["qest"], name


def test_library_exports_the_field_s_constants():
    lib = _lib.load()
    m, one, r2 = ((C.c_uint64 * 4)() for _ in range(3))
    assert lib.plk_field_constants_of(1, m, one, r2) == 0
    val = lambda a: sum(int(a[i]) << (64 * i) for i in range(4))
    p = F1.P_INT
    assert val(m) == p and val(one) == (1 << 256) % p and val(r2) == (1 << 512) % p
    assert lib.plk_field_constants_of(0, m, None, None) == 0 and val(m) == synth.P_INT
    assert lib.plk_field_constants_of(7, m, one, r2) == _lib.PLK_ERR_ARG
    assert lib.plk_field_name(1) == b"Tweedle Fr" and lib.plk_field_name(0) == b"BLS12-381 Fr" and lib.plk_field_name(9) is None
    h = C.c_void_p()
    assert lib.plk_ctx_create_field(0, 5, C.byref(h)) == _lib.PLK_ERR_ARG and not h.value


def test_c_restatement_equals_big_integers_over_the_second_field():
    """Transforms against the O(n^2) definition, and every function of the path, C against Python big integers."""
    with po.use_field(1), co.use_field(1):
        p = po.P
        x = F1.random_elements(11, 64)
        xi = F1.mont_to_ints(x)
        assert po.ntt(xi, 64) == po.dft_naive(xi, 64)
        assert F1.mont_to_ints(co.ntt(x, 64)) == po.ntt(xi, 64)
        assert F1.mont_to_ints(co.intt(x, 64)) == po.intt(xi, 64)
        assert F1.mont_to_ints(co.ntt(x[:16], 64)) == po.ntt(xi[:16], 64)
        w4 = po.root_of_unity(4)
        assert po.ntt([1, 2, 3, 4], 4) == [10, (1 + 2 * w4 + 3 * w4 ** 2 + 4 * w4 ** 3) % p, p - 2, (1 + 2 * w4 ** 3 + 3 * w4 ** 2 + 4 * w4) % p]
        n, N = 16, 64
        c = F1.make_circuit(n, circuit_id=3)
        wires = [F1.mont_to_ints(w) for w in c.table.wires()]
        sels = {k: F1.mont_to_ints(v) for k, v in c.table.selectors.items()}
        sig = [[int(v) for v in s] for s in c.perm.columns()]
        beta, gamma = F1.mont_to_ints(c.beta)[0], F1.mont_to_ints(c.gamma)[0]
        table = po.GateEvaluationTable(*wires, selectors=sels)
        perm = po.PermutationTable(*sig)
        z = po.build_grand_product_evaluation_polynomial(perm, table, beta, gamma)
        zr, gr, er = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), N, c.beta, c.gamma)
        assert F1.mont_to_ints(zr) == z
        gp = po.gate_evaluation_table_to_gate_polynomial(table, n)
        assert F1.mont_to_ints(gr) == po.build_gate_constraint_evaluation_polynomial(gp, N)
        pp = po.build_permutation_polynomials(perm, table, beta, gamma, n, strict=False)
        assert F1.mont_to_ints(er) == po.build_permutation_constraint_evaluation_polynomial(pp, gp, beta, gamma, N)
        # inverse and from_u64 conventions
        assert F1.mont_to_ints(co.fe_mul(F1.ints_to_mont([3])[0], F1.ints_to_mont([pow(3, -1, p)])[0])) == [1]
    # the switch is scoped: back on the first field
    assert po.P == synth.P_INT and np.array_equal(co.ntt(synth.ints_to_mont([1, 0, 0, 0]), 4), synth.ints_to_mont([1, 1, 1, 1]))
