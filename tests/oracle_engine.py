"""CPU stand-in for plonk_amd.device.DeviceProver used by the multi-process tests: the same
methods, computed by the C oracle on CPU tensors.  It lets the exchange logic of
plonk_amd.dist run under gloo without a GPU; it is test code, never the product path."""
import numpy as np
import torch

import c_oracle as co
import plonk_oracle as po
from plonk_amd import synth

KINDS = po.AST_KINDS


def t2n(t):
    return t.contiguous().numpy().view(np.uint64)


def n2t(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64))


class OracleEngine:
    device = torch.device("cpu")

    def msm_set_bases(self, bases):
        self._bases = np.ascontiguousarray(bases, dtype=np.uint64)

    def msm(self, scalars):
        s = t2n(scalars)
        return co.g1_msm(s, self._bases[: s.shape[0]])

    def g1_sum(self, points):
        # every point once: a multi-scalar multiplication with all scalars one
        p = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12)
        return co.g1_msm(synth.ints_to_mont([1] * p.shape[0]), p)

    def ntt_rows(self, mat, N, inverse=False):
        rows = [co.intt(t2n(r), N) if inverse else co.ntt(t2n(r), N) for r in mat]
        return n2t(np.stack(rows))

    def mul(self, a, b):
        fa, fb = t2n(a).reshape(-1, 4), t2n(b).reshape(-1, 4)
        out = np.stack([co.fe_mul(x, y) for x, y in zip(fa, fb)])
        return n2t(out.reshape(t2n(a).shape))

    def twiddle_block(self, N, rows, cols, row0, col0=0, inverse=False, scale_by_inverse_n=False):
        w = po.root_of_unity(N)
        if inverse:
            w = pow(w, -1, po.P)
        s = pow(N, -1, po.P) if scale_by_inverse_n else 1
        vals = [pow(w, (row0 + r) * (col0 + c), po.P) * s % po.P for r in range(rows) for c in range(cols)]
        return n2t(synth.ints_to_mont(vals).reshape(rows, cols, 4))

    def grand_product_cols(self, wires, sigma, n, beta, gamma):
        return n2t(co.grand_product([t2n(w) for w in wires], [t2n(s).reshape(-1) for s in sigma], beta, gamma))

    def _terms(self, wires, sigma, first, rows, beta, gamma):
        w = [synth.mont_to_ints(t2n(x)) for x in wires]
        sg = [[int(v) for v in t2n(s).reshape(-1)] for s in sigma]
        b, g = synth.mont_to_ints(np.asarray(beta))[0], synth.mont_to_ints(np.asarray(gamma))[0]
        num, den = [], []
        for j in range(first, first + rows):
            nj = dj = 1
            for k in range(4):
                nj = nj * (w[k][j] + b * w[k][sg[k][j]] + g) % po.P
                dj = dj * (w[k][j] + b * j + g) % po.P
            num.append(nj)
            den.append(dj)
        return num, den

    def grand_product_local(self, wires, sigma, n, first, rows, beta, gamma):
        self._gp = self._terms(wires, sigma, first, rows, beta, gamma)
        tn = td = 1
        for a, b in zip(*self._gp):
            tn, td = tn * a % po.P, td * b % po.P
        return n2t(synth.ints_to_mont([tn, td]))

    def grand_product_finish(self, rows, totals_all, mine):
        tot = synth.mont_to_ints(t2n(totals_all).reshape(-1, 4))      # [num_0, den_0, num_1, den_1, ...]
        z = 1
        for g in range(mine):                                          # z at the range's first row
            z = z * tot[2 * g] % po.P * pow(tot[2 * g + 1], -1, po.P) % po.P
        out = []
        for a, b in zip(*self._gp):
            z = z * a % po.P * pow(b, -1, po.P) % po.P
            out.append(z)
        assert len(out) == rows
        return n2t(synth.ints_to_mont(out))

    def ntt_batch(self, cols, N, inverse=False, kinds=None):
        out = []
        for i, c in enumerate(cols):
            a = t2n(c)
            if kinds is not None and kinds[i] == 1:
                a = synth.ints_to_mont([int(x) for x in a.reshape(-1)])
            out.append(n2t(co.intt(a, N) if inverse else co.ntt(a, N)))
        return out

    def gate_constraint_chunk(self, wire_evals, sel_evals, count):
        w = [synth.mont_to_ints(t2n(x)) for x in wire_evals]
        s = {k: synth.mont_to_ints(t2n(x)) for k, x in zip(KINDS, sel_evals) if x is not None}
        vals = [sum(s[k][i] * po.gate_expression(k, w[0][i], w[1][i], w[2][i], w[3][i]) for k in s) % po.P for i in range(count)]
        return n2t(synth.ints_to_mont(vals))

    def perm_constraint_chunk(self, z_evals, sigma_evals, wire_evals, count, index0, N, beta, gamma, z_shift=1):
        z = synth.mont_to_ints(t2n(z_evals))
        sg = [synth.mont_to_ints(t2n(x)) for x in sigma_evals]
        w = [synth.mont_to_ints(t2n(x)) for x in wire_evals]
        b, g = synth.mont_to_ints(np.asarray(beta))[0], synth.mont_to_ints(np.asarray(gamma))[0]
        vals = []
        for i in range(count):
            lhs = rhs = 1
            for k in range(4):
                lhs = lhs * (w[k][i] + b * (index0 + i) + g) % po.P
                rhs = rhs * (w[k][i] + b * sg[k][i] + g) % po.P
            vals.append((z[i + z_shift] * lhs - z[i] * rhs) % po.P)
        return n2t(synth.ints_to_mont(vals))
