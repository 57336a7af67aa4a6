"""Quick GPU-vs-oracle check used during development (the real tests live in tests/)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import c_oracle as co
import plonk_amd as pa
from plonk_amd import synth

def eq(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    ok = a.shape == b.shape and np.array_equal(a, b)
    if not ok:
        bad = np.flatnonzero((a.reshape(-1, 4) != b.reshape(-1, 4)).any(axis=1)) if a.shape == b.shape else []
        print(f"  MISMATCH {what}: {len(bad)} rows differ, first {bad[:8]}")
    return ok

sizes = [int(s) for s in sys.argv[1:]] or [1, 2, 4, 8, 64, 1024, 2048, 4096, 1 << 13, 1 << 16, 1 << 17]
allok = True
for n in sizes:
    t0 = time.time()
    c = synth.make_circuit(n)
    dom, dom4 = pa.Radix2EvaluationDomain(n), pa.Radix2EvaluationDomain(4 * n)
    ok = True
    coeff = pa.evaluations_to_dense_polynomial(c.table.input1, dom)
    ok &= eq(coeff, co.intt(c.table.input1, n), "intt")
    ev = pa.evaluate_over_domain(coeff, dom4)
    ok &= eq(ev, co.ntt(coeff, 4 * n), "ntt 4n")
    ok &= eq(pa.evaluate_over_domain(coeff, dom), c.table.input1, "ntt(intt)=id")
    gp = pa.gate_evaluation_table_to_gate_polynomial(c.table, dom)
    wc, sc = co.gate_polys(c.table.wires(), c.table.selectors, n)
    ok &= all(eq(a, b, "gate_polys wire") for a, b in zip(gp.wires(), wc))
    ok &= all(eq(gp.selectors[k], sc[k], "gate_polys sel " + k) for k in sc)
    z = pa.build_grand_product_evaluation_polynomial(c.perm, c.table, c.beta, c.gamma)
    zo = co.grand_product(c.table.wires(), c.perm.columns(), c.beta, c.gamma)
    ok &= eq(z, zo, "grand product")
    pp = pa.build_permutation_polynomials(c.perm, c.table, c.beta, c.gamma, dom)
    so, zc, zl = co.perm_polys(c.table.wires(), c.perm.columns(), c.beta, c.gamma, n)
    ok &= eq(pp.z, zc, "perm_polys z") and eq(pp.z_last, zl, "z_last") and all(eq(a, b, "sigma coeffs") for a, b in zip(pp.sigmas(), so))
    if n <= 1 << 14:
        for N in (n, 4 * n):
            d = pa.Radix2EvaluationDomain(N)
            ok &= eq(pa.build_gate_constraint_evaluation_polynomial(gp, d), co.gate_constraints(wc, sc, N), f"gate constraints N={N}")
            ok &= eq(pa.build_permutation_constraint_evaluation_polynomial(pp, gp, c.beta, c.gamma, d),
                     co.perm_constraints(zc, so, wc, N, c.beta, c.gamma), f"perm constraints N={N}")
        z2, g2, p2 = pa.prove_round23(c.table, c.perm, c.beta, c.gamma, dom, dom4)
        zr, gr, pr = co.round23(c.table.wires(), c.table.selectors, c.perm.columns(), 4 * n, c.beta, c.gamma)
        ok &= eq(z2, zr, "round23 z") and eq(g2, gr, "round23 gate") and eq(p2, pr, "round23 perm")
    print(f"n={n}: {'OK' if ok else 'FAIL'}  ({time.time() - t0:.1f}s)", flush=True)
    allok &= ok
sys.exit(0 if allok else 1)
