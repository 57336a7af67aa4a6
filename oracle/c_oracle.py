"""ctypes binding of oracle/liboracle.so (the C restatement in plonk_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Arrays are numpy `(n, 4) uint64` Montgomery limbs, sigma `(n,) uint64`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
AST_KINDS = ("Int", "Bool", "Add", "Sub", "Mult", "Div", "Eq", "Not", "If")
u64p = C.POINTER(C.c_uint64)
u64pp = C.POINTER(u64p)

ERRORS = {-1: "length", -2: "not_pow2", -3: "two_adicity", -4: "zero_denominator", -5: "sigma_range", -8: "nomem"}


class OracleError(RuntimeError):
    def __init__(self, code, row=-1):
        super().__init__(f"oracle error {code} ({ERRORS.get(code, '?')}) row={row}")
        self.code, self.row = code, row


#: the same restatement compiled over the second field (ORACLE_FIELD=1: Tweedle Fr)
LIB_PATHS = {0: LIB_PATH, 1: os.path.join(_HERE, "liboracle_tweedle.so")}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so", "liboracle_tweedle.so"])


_libs = {}
_field = 0


def lib():
    if _field not in _libs:
        if not os.path.exists(LIB_PATHS[_field]):
            build()
        l = C.CDLL(LIB_PATHS[_field])
        l.orc_fe_inverse.restype = C.c_int
        _libs[_field] = l
    return _libs[_field]


class use_field:
    """`with use_field(1): ...` routes every call of this module to the library built for that field (the
    G1 functions exist for field 0 only)."""

    def __init__(self, field_id: int):
        self.id = field_id

    def __enter__(self):
        global _field
        self.prev, _field = _field, self.id
        return self

    def __exit__(self, *exc):
        global _field
        _field = self.prev
        return False


def _p(a):
    return a.ctypes.data_as(u64p) if a is not None else None


def _tab(arrs):
    t = (u64p * len(arrs))()
    for i, a in enumerate(arrs):
        t[i] = _p(a) if a is not None else None
    return t


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _chk(rc, row=-1):
    if rc != 0:
        raise OracleError(rc, row)


def fe_mul(a, b):
    out = np.empty(4, dtype=np.uint64); lib().orc_fe_mul(_p(_c(a)), _p(_c(b)), _p(out)); return out


def fe_add(a, b):
    out = np.empty(4, dtype=np.uint64); lib().orc_fe_add(_p(_c(a)), _p(_c(b)), _p(out)); return out


def fe_sub(a, b):
    out = np.empty(4, dtype=np.uint64); lib().orc_fe_sub(_p(_c(a)), _p(_c(b)), _p(out)); return out


def fe_inverse(a):
    out = np.empty(4, dtype=np.uint64)
    if not lib().orc_fe_inverse(_p(_c(a)), _p(out)):
        raise ZeroDivisionError
    return out


def fe_from_u64(x):
    out = np.empty(4, dtype=np.uint64); lib().orc_fe_from_u64(C.c_uint64(int(x)), _p(out)); return out


def to_mont(a):
    a = _c(a); out = np.empty_like(a); lib().orc_to_mont(_p(a), C.c_size_t(a.shape[0]), _p(out)); return out


def from_mont(a):
    a = _c(a); out = np.empty_like(a); lib().orc_from_mont(_p(a), C.c_size_t(a.shape[0]), _p(out)); return out


def ntt(coeffs, N):
    coeffs = _c(coeffs); out = np.empty((N, 4), dtype=np.uint64)
    _chk(lib().orc_ntt(_p(coeffs), C.c_size_t(coeffs.shape[0]), C.c_size_t(N), _p(out))); return out


def intt(evals, domain_size):
    evals = _c(evals); out = np.empty((domain_size, 4), dtype=np.uint64)
    _chk(lib().orc_intt(_p(evals), C.c_size_t(evals.shape[0]), C.c_size_t(domain_size), _p(out))); return out


def _sel_list(selectors):
    return [(_c(selectors[k]) if k in selectors else None) for k in AST_KINDS]


def gate_polys(wires, selectors, domain_size):
    wires = [_c(w) for w in wires]; sels = _sel_list(selectors); n = wires[0].shape[0]
    w_out = [np.empty((domain_size, 4), dtype=np.uint64) for _ in range(4)]
    s_out = [np.empty((domain_size, 4), dtype=np.uint64) if s is not None else None for s in sels]
    _chk(lib().orc_gate_polys(_tab(wires), _tab(sels), C.c_size_t(n), C.c_size_t(domain_size), _tab(w_out), _tab(s_out)))
    return w_out, {k: s_out[i] for i, k in enumerate(AST_KINDS) if s_out[i] is not None}


def gate_constraints(wire_coeffs, sel_coeffs, N):
    wires = [_c(w) for w in wire_coeffs]; sels = _sel_list(sel_coeffs); nc = wires[0].shape[0]
    out = np.empty((N, 4), dtype=np.uint64)
    _chk(lib().orc_gate_constraints(_tab(wires), _tab(sels), C.c_size_t(nc), C.c_size_t(N), _p(out))); return out


def grand_product(wires, sigma, beta, gamma):
    wires = [_c(w) for w in wires]; sigma = [_c(s) for s in sigma]; n = wires[0].shape[0]
    out = np.empty((n + 1, 4), dtype=np.uint64); row = C.c_int64(-1)
    rc = lib().orc_grand_product(_tab(wires), _tab(sigma), C.c_size_t(n), _p(_c(beta)), _p(_c(gamma)), _p(out), C.byref(row))
    _chk(rc, row.value); return out


def perm_polys(wires, sigma, beta, gamma, domain_size):
    wires = [_c(w) for w in wires]; sigma = [_c(s) for s in sigma]; n = wires[0].shape[0]
    s_out = [np.empty((domain_size, 4), dtype=np.uint64) for _ in range(4)]
    z_out = np.empty((domain_size, 4), dtype=np.uint64); z_last = np.empty(4, dtype=np.uint64); row = C.c_int64(-1)
    rc = lib().orc_perm_polys(_tab(wires), _tab(sigma), C.c_size_t(n), C.c_size_t(domain_size), _p(_c(beta)), _p(_c(gamma)),
                              _tab(s_out), _p(z_out), _p(z_last), C.byref(row))
    _chk(rc, row.value); return s_out, z_out, z_last


def perm_constraints(z_coeffs, sigma_coeffs, wire_coeffs, N, beta, gamma, z_shift=1):
    z = _c(z_coeffs); sg = [_c(s) for s in sigma_coeffs]; w = [_c(x) for x in wire_coeffs]
    out = np.empty((N, 4), dtype=np.uint64)
    _chk(lib().orc_perm_constraints(_p(z), _tab(sg), _tab(w), C.c_size_t(z.shape[0]), C.c_size_t(N), _p(_c(beta)), _p(_c(gamma)),
                                    C.c_size_t(z_shift), _p(out))); return out


def round23(wires, selectors, sigma, N, beta, gamma, z_shift=1):
    wires = [_c(w) for w in wires]; sels = _sel_list(selectors); sigma = [_c(s) for s in sigma]; n = wires[0].shape[0]
    z = np.empty((n + 1, 4), dtype=np.uint64); g = np.empty((N, 4), dtype=np.uint64); p = np.empty((N, 4), dtype=np.uint64)
    row = C.c_int64(-1)
    rc = lib().orc_round23(_tab(wires), _tab(sels), _tab(sigma), C.c_size_t(n), C.c_size_t(N), _p(_c(beta)), _p(_c(gamma)),
                           C.c_size_t(z_shift), _p(z), _p(g), _p(p), C.byref(row))
    _chk(rc, row.value); return z, g, p


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(t):
    lib().orc_set_num_threads(C.c_int(t))


def sigma_table(node_ids):
    """orc_sigma_table: four `(n,) uint64` node-id columns -> four `(n,) uint64` sigma columns."""
    cols = [_c(a) for a in node_ids]
    n = cols[0].shape[0]
    out = [np.empty((n,), dtype=np.uint64) for _ in range(4)]
    rc = lib().orc_sigma_table(_tab(cols), C.c_size_t(n), _tab(out))
    if rc:
        raise OracleError(rc)
    return out


def gate_table_fill(ops, node_ids, node_values, default_val, kinds_present=AST_KINDS):
    """orc_gate_table_fill -> (4 wire columns, {kind: selector column})."""
    ops = np.ascontiguousarray(ops, dtype=np.uint8)
    cols = [_c(a) for a in node_ids]
    vals = _c(node_values)
    n = ops.shape[0]
    w_out = [np.empty((n, 4), dtype=np.uint64) for _ in range(4)]
    s_out = [np.empty((n, 4), dtype=np.uint64) if k in kinds_present else None for k in AST_KINDS]
    rc = lib().orc_gate_table_fill(ops.ctypes.data_as(C.POINTER(C.c_uint8)), _tab(cols), C.c_size_t(n), _p(vals),
                                   C.c_size_t(vals.shape[0]), _p(_c(default_val)), _tab(w_out), _tab(s_out))
    if rc:
        raise OracleError(rc)
    return w_out, {k: s_out[i] for i, k in enumerate(AST_KINDS) if s_out[i] is not None}


def coset_ntt(coeffs, N, shift):
    coeffs = _c(coeffs); out = np.empty((N, 4), dtype=np.uint64)
    _chk(lib().orc_coset_ntt(_p(coeffs), C.c_size_t(coeffs.shape[0]), C.c_size_t(N), _p(_c(shift)), _p(out))); return out


def coset_intt(evals, shift):
    evals = _c(evals); N = evals.shape[0]; out = np.empty((N, 4), dtype=np.uint64)
    _chk(lib().orc_coset_intt(_p(evals), C.c_size_t(N), _p(_c(shift)), _p(out))); return out


def quotient(numerators, alpha, n, shift):
    nums = [_c(a) for a in numerators]; N = nums[0].shape[0]; out = np.empty((N, 4), dtype=np.uint64)
    _chk(lib().orc_quotient(_tab(nums), C.c_size_t(len(nums)), _p(_c(alpha)), C.c_size_t(n), C.c_size_t(N), _p(_c(shift)), _p(out)))
    return out


def g1_msm(scalars, bases):
    """orc_g1_msm: scalars `(n, 4) uint64` (Montgomery Fr), bases `(n, 12) uint64` affine -> `(12,) uint64` affine point."""
    s = _c(scalars); b = _c(bases); out = np.zeros(12, dtype=np.uint64)
    assert b.shape[0] >= s.shape[0]
    _chk(lib().orc_g1_msm(_p(s), _p(b), C.c_size_t(s.shape[0]), _p(out))); return out
