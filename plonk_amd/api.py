"""Host-side mirror of the reference's polynomial interface, running on libplonk_hip.so.

Same function names, argument order and failure behaviour as the Rust functions of
wu-s-john/PLONK (src/polynomial_utils.rs, src/execution_trace.rs:304-431,
src/permutation_polynomial.rs:204-376); a Rust panic becomes a Python exception.

Field elements are rows of 4 little-endian uint64 Montgomery limbs (numpy `(n, 4) uint64`),
the in-memory form of ark-ff's `Fp256`; sigma columns are `(n,) uint64` row indices.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional, Sequence

import numpy as np

from . import _lib

#: `ASTKind` declaration order, src/ast.rs:56-66 — the index of a selector column in the C ABI
AST_KINDS = ("Int", "Bool", "Add", "Sub", "Mult", "Div", "Eq", "Not", "If")
#: marks a wire the row's operation does not use, in the node-id columns of build_permutation_table
NO_WIRE = _lib.NO_WIRE


class PlonkError(RuntimeError):
    def __init__(self, code: int, message: str, row: int = -1):
        super().__init__(f"[{code}] {message}")
        self.code, self.row = code, row


class LengthMismatch(PlonkError, AssertionError):
    """assert_eq! on lengths (src/polynomial_utils.rs:22-26, :56-60)."""


class ZeroDenominator(PlonkError, ZeroDivisionError):
    """`denominator.inverse().unwrap()` on zero (src/permutation_polynomial.rs:258)."""


class SigmaOutOfRange(PlonkError, IndexError):
    """slice index panic (src/permutation_polynomial.rs:235-238)."""


_EXC = {_lib.PLK_ERR_LENGTH: LengthMismatch, _lib.PLK_ERR_ZERO_DENOMINATOR: ZeroDenominator,
        _lib.PLK_ERR_SIGMA_RANGE: SigmaOutOfRange}


@dataclass(frozen=True)
class Radix2EvaluationDomain:
    """What ark-poly's `GeneralEvaluationDomain::new(size)` yields for these fields; only the size
    crosses the ABI (the generator follows from the field's two-adic root)."""
    size_: int

    def size(self) -> int:
        return self.size_


GeneralEvaluationDomain = Radix2EvaluationDomain


@dataclass
class GateEvaluationTable:
    """src/execution_trace.rs:204-214."""
    input1: np.ndarray
    input2: np.ndarray
    input3: np.ndarray
    output: np.ndarray
    selectors: Dict[str, np.ndarray] = field(default_factory=dict)

    def wires(self):
        return (self.input1, self.input2, self.input3, self.output)


@dataclass
class GatePolynomials:
    """src/execution_trace.rs:307-314; coefficient vectors keep their trailing zeros."""
    input1: np.ndarray
    input2: np.ndarray
    input3: np.ndarray
    output: np.ndarray
    selectors: Dict[str, np.ndarray] = field(default_factory=dict)

    def wires(self):
        return (self.input1, self.input2, self.input3, self.output)


@dataclass
class PermutationTable:
    """src/permutation_polynomial.rs:51-61 with F = usize."""
    sigma1: np.ndarray
    sigma2: np.ndarray
    sigma3: np.ndarray
    sigma_output: np.ndarray

    def columns(self):
        return (self.sigma1, self.sigma2, self.sigma3, self.sigma_output)


@dataclass
class PermutationPolynomials:
    """src/permutation_polynomial.rs:263-269 (+ z_last = z[n], see build_permutation_polynomials)."""
    sigma1: np.ndarray
    sigma2: np.ndarray
    sigma3: np.ndarray
    sigma_output: np.ndarray
    z: np.ndarray
    z_last: Optional[np.ndarray] = None

    def sigmas(self):
        return (self.sigma1, self.sigma2, self.sigma3, self.sigma_output)


def degree(coeffs: np.ndarray) -> int:
    """Degree ark-poly's `DensePolynomial::from_coefficients_vec` would report after dropping
    trailing zero coefficients (src/polynomial_utils.rs:32); -1 for the zero polynomial."""
    nz = np.flatnonzero(coeffs.reshape(-1, 4).any(axis=1))
    return int(nz[-1]) if nz.size else -1


# --------------------------------------------------------------------------- plumbing


def _col(a: np.ndarray, name: str) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim != 2 or a.shape[1] != 4:
        raise ValueError(f"{name}: expected (n, 4) uint64 Montgomery limbs, got {a.shape}")
    return a


def _idx(a: np.ndarray, name: str) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim != 1:
        raise ValueError(f"{name}: expected (n,) uint64 indices, got {a.shape}")
    return a


def _p(a: Optional[np.ndarray]):
    return a.ctypes.data_as(_lib.u64p) if a is not None else None


def _table(arrs: Sequence[Optional[np.ndarray]]):
    t = (_lib.u64p * len(arrs))()
    for i, a in enumerate(arrs):
        t[i] = _p(a) if a is not None else None
    return t


def _sel_cols(selectors: Dict[str, np.ndarray]):
    unknown = set(selectors) - set(AST_KINDS)
    if unknown:
        raise KeyError(f"unknown selector kinds {sorted(unknown)}")
    return [(_col(selectors[k], k) if k in selectors else None) for k in AST_KINDS]


def _fe(x) -> np.ndarray:
    a = np.ascontiguousarray(x, dtype=np.uint64).reshape(4)
    return a


class Context:
    """One library context (one GPU, one field).  `default()` creates it on first use on device 0 for the BLS12-381
    scalar field; `field` is a PLK_FIELD_* number or one of `_lib.FIELD_IDS`' names."""
    _default: Optional["Context"] = None

    def __init__(self, device: int = 0, field=0):
        self.lib = _lib.load()
        self.field = _lib.FIELD_IDS[field] if isinstance(field, str) else int(field)
        h = C.c_void_p()
        rc = self.lib.plk_ctx_create_field(device, self.field, C.byref(h))
        if rc != _lib.PLK_OK:
            raise PlonkError(rc, f"plk_ctx_create_field(device={device}, field={field}): {self.lib.plk_strerror(rc).decode()}")
        self.handle = h
        self.device = device

    def close(self):
        if getattr(self, "handle", None):
            self.lib.plk_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def default(cls) -> "Context":
        if cls._default is None:
            cls._default = cls(0)
        return cls._default

    def pinned_empty(self, shape, dtype=np.uint64) -> np.ndarray:
        """A numpy array in page-locked host memory (plk_host_alloc): the host entry points move such
        arrays by DMA.  The memory is released when the array and every view of it are gone."""
        import weakref
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self.check(self.lib.plk_host_alloc(self.handle, nbytes, C.byref(p)))
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
        # plk_host_free ignores its context argument, so the buffer may outlive this context
        weakref.finalize(buf, self.lib.plk_host_free, None, p)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def check(self, rc: int):
        if rc == _lib.PLK_OK:
            return
        msg = self.lib.plk_last_error(self.handle).decode() or self.lib.plk_strerror(rc).decode()
        row = int(self.lib.plk_last_error_row(self.handle))
        raise _EXC.get(rc, PlonkError)(rc, msg, row)


def _ctx(ctx: Optional[Context]) -> Context:
    return ctx if ctx is not None else Context.default()


# --------------------------------------------------------------------------- the seven functions


def evaluations_to_dense_polynomial(evals: np.ndarray, domain: Radix2EvaluationDomain, ctx: Optional[Context] = None) -> np.ndarray:
    """src/polynomial_utils.rs:18-33."""
    c = _ctx(ctx)
    evals = _col(evals, "evals")
    n = domain.size()
    out = np.empty((n, 4), dtype=np.uint64)
    c.check(c.lib.plk_intt(c.handle, _p(evals), evals.shape[0], n, _p(out)))
    return out


def build_and_convert(domain: Radix2EvaluationDomain, builder: Callable[[], np.ndarray], ctx: Optional[Context] = None) -> np.ndarray:
    """src/polynomial_utils.rs:47-62; `builder` returns the evaluation vector."""
    c = _ctx(ctx)
    n = domain.size()
    out = np.empty((n, 4), dtype=np.uint64)

    def thunk(_user, dst, capacity):
        ev = _col(builder(), "builder()")
        m = min(ev.shape[0], capacity)
        if m:
            C.memmove(dst, ev.ctypes.data, m * 32)
        return ev.shape[0]

    c.check(c.lib.plk_build_and_convert(c.handle, n, _lib.BUILDER_FN(thunk), None, _p(out)))
    return out


def evaluate_over_domain(coeffs: np.ndarray, domain: Radix2EvaluationDomain, ctx: Optional[Context] = None) -> np.ndarray:
    """`DensePolynomial::evaluate_over_domain_by_ref` (src/execution_trace.rs:362-370)."""
    c = _ctx(ctx)
    coeffs = _col(coeffs, "coeffs")
    n = domain.size()
    out = np.empty((n, 4), dtype=np.uint64)
    c.check(c.lib.plk_ntt(c.handle, _p(coeffs), coeffs.shape[0], n, _p(out)))
    return out


def gate_evaluation_table_to_gate_polynomial(table: GateEvaluationTable, coset_domain: Radix2EvaluationDomain,
                                             ctx: Optional[Context] = None) -> GatePolynomials:
    """src/execution_trace.rs:318-344."""
    c = _ctx(ctx)
    wires = [_col(w, "wire") for w in table.wires()]
    sels = _sel_cols(table.selectors)
    n, d = wires[0].shape[0], coset_domain.size()
    for a in wires + [s for s in sels if s is not None]:
        if a.shape[0] != d:  # the per-column assert of src/polynomial_utils.rs:22-26
            raise LengthMismatch(_lib.PLK_ERR_LENGTH, "Evaluation length must match the domain size")
    w_out = [np.empty((d, 4), dtype=np.uint64) for _ in range(4)]
    s_out = [np.empty((d, 4), dtype=np.uint64) if s is not None else None for s in sels]
    c.check(c.lib.plk_gate_polys(c.handle, _table(wires), _table(sels), n, d, _table(w_out), _table(s_out)))
    return GatePolynomials(*w_out, selectors={k: s_out[i] for i, k in enumerate(AST_KINDS) if s_out[i] is not None})


def build_gate_constraint_evaluation_polynomial(gate_poly: GatePolynomials, domain: Radix2EvaluationDomain,
                                                ctx: Optional[Context] = None) -> np.ndarray:
    """src/execution_trace.rs:357-431."""
    c = _ctx(ctx)
    wires = [_col(w, "wire") for w in gate_poly.wires()]
    sels = _sel_cols(gate_poly.selectors)
    nc = wires[0].shape[0]
    if any(a.shape[0] != nc for a in wires + [s for s in sels if s is not None]):
        raise ValueError("coefficient vectors must share one length (pad with zero coefficients)")
    N = domain.size()
    out = np.empty((N, 4), dtype=np.uint64)
    c.check(c.lib.plk_gate_constraints(c.handle, _table(wires), _table(sels), nc, N, _p(out)))
    return out


def build_grand_product_evaluation_polynomial(permutation_table: PermutationTable, evaluation_table: GateEvaluationTable,
                                              beta: np.ndarray, gamma: np.ndarray, ctx: Optional[Context] = None) -> np.ndarray:
    """src/permutation_polynomial.rs:217-261 — returns n + 1 elements."""
    c = _ctx(ctx)
    wires = [_col(w, "wire") for w in evaluation_table.wires()]
    sig = [_idx(s, "sigma") for s in permutation_table.columns()]
    n = wires[0].shape[0]  # `evaluation_table.input1.len()`, :223
    for a in wires[1:]:
        if a.shape[0] < n:
            raise SigmaOutOfRange(_lib.PLK_ERR_SIGMA_RANGE, "wire column shorter than input1")
    for s in sig:
        if s.shape[0] < n:
            raise SigmaOutOfRange(_lib.PLK_ERR_SIGMA_RANGE, "sigma column shorter than input1")
    out = np.empty((n + 1, 4), dtype=np.uint64)
    b, g = _fe(beta), _fe(gamma)
    c.check(c.lib.plk_grand_product(c.handle, _table(wires), _table(sig), n, _p(b), _p(g), _p(out)))
    return out


def build_permutation_polynomials(permutation_table: PermutationTable, evaluation_table: GateEvaluationTable,
                                  beta: np.ndarray, gamma: np.ndarray, coset_domain: Radix2EvaluationDomain,
                                  ctx: Optional[Context] = None) -> PermutationPolynomials:
    """src/permutation_polynomial.rs:272-298.  z[0..n) is interpolated; z[n] comes back as `z_last`
    (the reference's own length assert rejects the n+1 vector)."""
    c = _ctx(ctx)
    wires = [_col(w, "wire") for w in evaluation_table.wires()]
    sig = [_idx(s, "sigma") for s in permutation_table.columns()]
    n, d = wires[0].shape[0], coset_domain.size()
    if any(s.shape[0] != d for s in sig) or n != d:
        raise LengthMismatch(_lib.PLK_ERR_LENGTH, "Evaluation length must match the domain size")
    s_out = [np.empty((d, 4), dtype=np.uint64) for _ in range(4)]
    z_out = np.empty((d, 4), dtype=np.uint64)
    z_last = np.empty((4,), dtype=np.uint64)
    b, g = _fe(beta), _fe(gamma)
    c.check(c.lib.plk_perm_polys(c.handle, _table(wires), _table(sig), n, d, _p(b), _p(g), _table(s_out), _p(z_out), _p(z_last)))
    return PermutationPolynomials(*s_out, z=z_out, z_last=z_last)


def build_permutation_constraint_evaluation_polynomial(polynomials: PermutationPolynomials, gate_polynomials: GatePolynomials,
                                                       beta: np.ndarray, gamma: np.ndarray,
                                                       evaluation_domain: Radix2EvaluationDomain, z_shift: int = 1,
                                                       ctx: Optional[Context] = None) -> np.ndarray:
    """src/permutation_polynomial.rs:312-376 (z_shift = 1 is the reference)."""
    c = _ctx(ctx)
    z = _col(polynomials.z, "z")
    sig = [_col(s, "sigma") for s in polynomials.sigmas()]
    wires = [_col(w, "wire") for w in gate_polynomials.wires()]
    nc = z.shape[0]
    if any(a.shape[0] != nc for a in sig + wires):
        raise ValueError("coefficient vectors must share one length (pad with zero coefficients)")
    N = evaluation_domain.size()
    out = np.empty((N, 4), dtype=np.uint64)
    b, g = _fe(beta), _fe(gamma)
    c.check(c.lib.plk_perm_constraints(c.handle, _p(z), _table(sig), _table(wires), nc, N, _p(b), _p(g), z_shift, _p(out)))
    return out


#: F::multiplicative_generator() of BLS12-381 Fr: the coset shift ark-poly 0.3's coset_fft uses
#: `F::multiplicative_generator()` per field id (7 is ark-ff's for BLS12-381 Fr; 5 for the Tweedle field is self-derived)
MULTIPLICATIVE_GENERATORS = {0: 7, 1: 5}
MULTIPLICATIVE_GENERATOR = MULTIPLICATIVE_GENERATORS[0]


def _shift(shift, ctx: "Context") -> np.ndarray:
    if shift is None:
        from .synth import field
        return _fe(field(ctx.field).ints_to_mont([MULTIPLICATIVE_GENERATORS[ctx.field]])[0])
    return _fe(shift)


def coset_fft(coeffs: np.ndarray, domain: Radix2EvaluationDomain, shift=None, ctx: Optional[Context] = None) -> np.ndarray:
    """ark-poly `EvaluationDomain::coset_fft`: evaluations at shift * omega^i (shift defaults to the generator 7)."""
    c = _ctx(ctx)
    coeffs = _col(coeffs, "coeffs")
    out = np.empty((domain.size(), 4), dtype=np.uint64)
    c.check(c.lib.plk_coset_ntt(c.handle, _p(coeffs), coeffs.shape[0], domain.size(), _p(_shift(shift, c)), 0, _p(out)))
    return out


def coset_ifft(evals: np.ndarray, domain: Radix2EvaluationDomain, shift=None, ctx: Optional[Context] = None) -> np.ndarray:
    """ark-poly `EvaluationDomain::coset_ifft`: coefficients from evaluations at shift * omega^i."""
    c = _ctx(ctx)
    evals = _col(evals, "evals")
    out = np.empty((domain.size(), 4), dtype=np.uint64)
    c.check(c.lib.plk_coset_ntt(c.handle, _p(evals), evals.shape[0], domain.size(), _p(_shift(shift, c)), 1, _p(out)))
    return out


def quotient_polynomial(numerators: Sequence[np.ndarray], alpha: np.ndarray, n_domain: Radix2EvaluationDomain,
                        evaluation_domain: Radix2EvaluationDomain, shift=None, ctx: Optional[Context] = None) -> np.ndarray:
    """Coefficients of (sum_i alpha^i numerators[i]) / (x^n - 1), the numerators being evaluation vectors on the
    coset shift * <omega_N>.  Raises ZeroDenominator if x^n - 1 vanishes on that coset."""
    c = _ctx(ctx)
    nums = [_col(a, "numerator") for a in numerators]
    N = evaluation_domain.size()
    if any(a.shape[0] != N for a in nums):
        raise LengthMismatch(_lib.PLK_ERR_LENGTH, "Evaluation length must match the domain size")
    out = np.empty((N, 4), dtype=np.uint64)
    c.check(c.lib.plk_quotient(c.handle, _table(nums), len(nums), _p(_fe(alpha)), n_domain.size(), N, _p(_shift(shift, c)), _p(out)))
    return out


def msm_set_bases(bases: np.ndarray, ctx: Optional[Context] = None) -> None:
    """Copy the affine G1 bases (`(n, 12) uint64`: x then y, Montgomery Fp384 limbs; (0, 0) = infinity) to the GPU."""
    c = _ctx(ctx)
    b = np.ascontiguousarray(bases, dtype=np.uint64)
    if b.ndim != 2 or b.shape[1] != 12:
        raise ValueError(f"bases: expected (n, 12) uint64, got {b.shape}")
    c.check(c.lib.plk_msm_set_bases(c.handle, _p(b), b.shape[0]))


def msm_g1(scalars: np.ndarray, ctx: Optional[Context] = None) -> np.ndarray:
    """sum_i scalars[i] * bases[i] (`VariableBaseMSM::multi_scalar_mul`, the body of `KZG10::commit`): `(12,) uint64`."""
    c = _ctx(ctx)
    s = _col(scalars, "scalars")
    out = np.zeros((12,), dtype=np.uint64)
    c.check(c.lib.plk_msm_g1(c.handle, _p(s), s.shape[0], _p(out)))
    return out


def g1_sum(points: np.ndarray) -> np.ndarray:
    """The sum of affine G1 points `(count, 12) uint64` on the host: `(12,) uint64`."""
    from . import _lib
    p = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros((12,), dtype=np.uint64)
    rc = _lib.load().plk_g1_sum(_p(p), p.shape[0], _p(out))
    if rc != 0:
        raise ValueError("plk_g1_sum: bad argument")
    return out


def msm_g1_batch(scalar_columns: Sequence[np.ndarray], ctx: Optional[Context] = None) -> np.ndarray:
    """One sum per column over the same bases (the commitments of one prover round): `(count, 12) uint64`."""
    c = _ctx(ctx)
    cols = [_col(s, "scalars") for s in scalar_columns]
    if len({s.shape[0] for s in cols}) > 1:
        raise ValueError("scalar columns differ in length")
    out = np.zeros((len(cols), 12), dtype=np.uint64)
    if cols:
        c.check(c.lib.plk_msm_g1_batch(c.handle, _table(cols), len(cols), cols[0].shape[0], _p(out)))
    return out


def row_ops(kinds: Sequence[str], inputs: Sequence[int]) -> np.ndarray:
    """Row descriptors of the flattened trace: `ASTKind` index | number of inputs << 4, one byte per row."""
    k = np.array([AST_KINDS.index(x) for x in kinds], dtype=np.uint8)
    a = np.asarray(inputs, dtype=np.uint8)
    if k.shape != a.shape or ((a < 1) | (a > 3)).any():
        raise ValueError("row_ops: one kind and an input count of 1..3 per row")
    return k | (a << np.uint8(4))


def build_gate_evaluation_table(ops: np.ndarray, node_ids: Sequence[np.ndarray], node_values: np.ndarray, default_val: np.ndarray,
                                kinds_present: Optional[Sequence[str]] = None, ctx: Optional[Context] = None) -> GateEvaluationTable:
    """build_gate_evaluation_table (src/execution_trace.rs:248-302, completed) on the flattened trace:
    `ops` from row_ops(), `node_ids` as for build_permutation_table, `node_values` the `(m, 4)` evaluated
    value of every node.  Selector columns are produced for `kinds_present` (default: the kinds that occur)."""
    c = _ctx(ctx)
    ops = np.ascontiguousarray(ops, dtype=np.uint8)
    cols = [_idx(a, "node_ids") for a in node_ids]
    n = ops.shape[0]
    if len(cols) != 4 or any(a.shape[0] != n for a in cols):
        raise ValueError("node_ids: four columns of one length expected")
    vals = _col(node_values, "node_values")
    if kinds_present is None:
        kinds_present = [AST_KINDS[k] for k in sorted(set((ops & np.uint8(15)).tolist()))]
    w_out = [np.empty((n, 4), dtype=np.uint64) for _ in range(4)]
    s_out = [np.empty((n, 4), dtype=np.uint64) if k in kinds_present else None for k in AST_KINDS]
    dv = _fe(default_val)
    c.check(c.lib.plk_gate_table_fill(c.handle, ops.ctypes.data_as(_lib.u8p), _table(cols), n, _p(vals), vals.shape[0], _p(dv),
                                      _table(w_out), _table(s_out)))
    return GateEvaluationTable(*w_out, selectors={k: s_out[i] for i, k in enumerate(AST_KINDS) if s_out[i] is not None})


def build_permutation_table(node_ids: Sequence[np.ndarray], ctx: Optional[Context] = None) -> PermutationTable:
    """build_permutation_trace + build_permutation_table (src/permutation_polynomial.rs:18-39, :79-202) on
    the flattened trace: `node_ids[w][j]` is the node id on wire w (inputs 0-2, output 3) of global row j,
    `NO_WIRE` where the row's operation has no such wire.  Returns the row-valued sigma columns."""
    c = _ctx(ctx)
    cols = [_idx(a, "node_ids") for a in node_ids]
    if len(cols) != 4 or any(a.shape[0] != cols[0].shape[0] for a in cols):
        raise ValueError("node_ids: four columns of one length expected")
    n = cols[0].shape[0]
    out = [np.empty((n,), dtype=np.uint64) for _ in range(4)]
    c.check(c.lib.plk_sigma_table(c.handle, _table(cols), n, _table(out)))
    return PermutationTable(*out)


def prove_round23(evaluation_table: GateEvaluationTable, permutation_table: PermutationTable, beta: np.ndarray,
                  gamma: np.ndarray, n_domain: Radix2EvaluationDomain, evaluation_domain: Radix2EvaluationDomain,
                  z_shift: int = 1, ctx: Optional[Context] = None, out=None):
    """Rounds 2+3 in one device-resident call: returns (z [n+1], gate constraints [N], permutation constraints [N]).
    `out` may supply the three result arrays (e.g. from Context.pinned_empty)."""
    c = _ctx(ctx)
    wires = [_col(w, "wire") for w in evaluation_table.wires()]
    sels = _sel_cols(evaluation_table.selectors)
    sig = [_idx(s, "sigma") for s in permutation_table.columns()]
    n, N = n_domain.size(), evaluation_domain.size()
    if any(a.shape[0] != n for a in wires + sig + [s for s in sels if s is not None]):
        raise LengthMismatch(_lib.PLK_ERR_LENGTH, "Evaluation length must match the domain size")
    if out is not None:
        z, g_out, p_out = out
        if (z.shape, g_out.shape, p_out.shape) != ((n + 1, 4), (N, 4), (N, 4)) or not all(
                a.dtype == np.uint64 and a.flags.c_contiguous for a in out):
            raise ValueError("out: contiguous uint64 arrays of shapes (n + 1, 4), (N, 4), (N, 4) expected")
    else:
        z = np.empty((n + 1, 4), dtype=np.uint64)
        g_out = np.empty((N, 4), dtype=np.uint64)
        p_out = np.empty((N, 4), dtype=np.uint64)
    b, g = _fe(beta), _fe(gamma)
    c.check(c.lib.plk_round23(c.handle, _table(wires), _table(sels), _table(sig), n, N, _p(b), _p(g), z_shift,
                              _p(z), _p(g_out), _p(p_out)))
    return z, g_out, p_out
