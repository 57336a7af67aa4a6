"""Device-resident driver of the hot path: columns live in HBM as torch tensors, the library's
`plk_dev_*` entry points are enqueued on torch's current HIP stream.

torch is used for device memory, streams and (in plonk_amd.dist) collectives only; every
arithmetic step runs in libplonk_hip.so.  A column is an `(n, 4) int64` CUDA tensor holding the
same bytes as the `(n, 4) uint64` Montgomery limbs of the host interface; sigma is `(n,) int64`.
"""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .api import AST_KINDS, Context
from .synth import Circuit


def _ptr(t: Optional[torch.Tensor]):
    return C.cast(C.c_void_p(t.data_ptr()), _lib.u64p) if t is not None else None


def _table(ts: Sequence[Optional[torch.Tensor]]):
    """A C array of column pointers (one ctypes cast for the whole table: a cast per pointer costs microseconds each,
    which is what a small circuit's call would then spend most of its time on)."""
    arr = (C.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts])
    return C.cast(arr, _lib.u64pp)


def _host_fe(x) -> np.ndarray:
    return np.ascontiguousarray(x, dtype=np.uint64).reshape(4)


def to_device(a: np.ndarray, device: torch.device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(device)


def to_host(t: torch.Tensor) -> np.ndarray:
    return t.cpu().numpy().view(np.uint64)


@dataclass
class DeviceCircuit:
    n: int
    wires: List[torch.Tensor]
    selectors: Dict[str, torch.Tensor]
    sigma: List[torch.Tensor]
    beta: np.ndarray
    gamma: np.ndarray
    _tabs: Optional[tuple] = None
    _tab_ptrs: Optional[tuple] = None

    def tables(self):
        """(wires, selectors, sigma) pointer tables, rebuilt only when a column has been rebound to another tensor
        (17 addresses compared per call; a cast per pointer per call is what a small circuit's call would otherwise
        spend most of its time on).  beta and gamma may be replaced at will, they are passed by value each call."""
        sels = [self.selectors.get(k) for k in AST_KINDS]
        ptrs = tuple(t.data_ptr() if t is not None else 0 for t in (*self.wires, *sels, *self.sigma))
        if self._tabs is None or ptrs != self._tab_ptrs:
            self._tabs = (_table(self.wires), _table(sels), _table(self.sigma))
            self._tab_ptrs = ptrs
        return self._tabs


class DeviceProver:
    """One GPU's context plus helpers to run the path on device-resident columns."""

    def __init__(self, device_index: int = 0, field=0):
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the hot path has no CPU fallback")
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.ctx = Context(device_index, field)
        self.lib = self.ctx.lib

    # -- plumbing
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def empty(self, rows: int) -> torch.Tensor:
        return torch.empty((rows, 4), dtype=torch.int64, device=self.device)

    def upload(self, c: Circuit) -> DeviceCircuit:
        d = self.device
        return DeviceCircuit(c.n, [to_device(w, d) for w in c.table.wires()],
                             {k: to_device(v, d) for k, v in c.table.selectors.items()},
                             [to_device(s, d) for s in c.perm.columns()], _host_fe(c.beta), _host_fe(c.gamma))

    def status(self):
        """Wait for the stream and raise if a data-dependent failure was recorded."""
        self.ctx.check(self.lib.plk_dev_status(self.ctx.handle, self._stream()))

    # -- entry points
    def ntt_batch(self, cols: Sequence[torch.Tensor], N: int, inverse: bool = False, n_in: Optional[int] = None,
                  kinds: Optional[Sequence[int]] = None, out: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
        n_in = cols[0].shape[0] if n_in is None else n_in
        out = [self.empty(N) for _ in cols] if out is None else list(out)
        karr = (C.c_uint32 * len(cols))(*kinds) if kinds is not None else None
        self.ctx.check(self.lib.plk_dev_ntt_batch(self.ctx.handle, self._stream(), _table(cols), karr, len(cols), n_in, N,
                                                  1 if inverse else 0, _table(out)))
        return out

    def g1_multiples(self, base_xy: np.ndarray, first: int, count: int) -> torch.Tensor:
        """`(count, 12) int64` affine points (first + i) * base on the device."""
        out = torch.empty((count, 12), dtype=torch.int64, device=self.device)
        b = np.ascontiguousarray(base_xy, dtype=np.uint64).reshape(12)
        self.ctx.check(self.lib.plk_dev_g1_multiples(self.ctx.handle, self._stream(), b.ctypes.data_as(_lib.u64p), first, count, _ptr(out)))
        return out

    def msm_set_bases(self, bases: torch.Tensor) -> None:
        """Use a device table of affine bases (kept alive by this object) for the following msm calls."""
        self._msm_bases = bases.contiguous()
        self.ctx.check(self.lib.plk_dev_msm_set_bases(self.ctx.handle, _ptr(self._msm_bases), self._msm_bases.shape[0]))

    def msm(self, scalars: torch.Tensor) -> np.ndarray:
        """sum_i scalars[i] * bases[i] for device scalars `(n, 4) int64`; returns the affine point `(12,) uint64` (host)."""
        out = np.zeros((12,), dtype=np.uint64)
        self.ctx.check(self.lib.plk_dev_msm_g1(self.ctx.handle, self._stream(), _ptr(scalars), scalars.shape[0], out.ctypes.data_as(_lib.u64p)))
        return out

    def g1_sum(self, points: np.ndarray) -> np.ndarray:
        """The sum of affine points `(count, 12) uint64` (host arithmetic)."""
        from .api import g1_sum
        return g1_sum(points)

    def msm_batch(self, scalar_columns: Sequence[torch.Tensor]) -> np.ndarray:
        """One sum per device column over the same bases; returns `(count, 12) uint64` (host)."""
        out = np.zeros((len(scalar_columns), 12), dtype=np.uint64)
        if len(scalar_columns):
            self.ctx.check(self.lib.plk_dev_msm_g1_batch(self.ctx.handle, self._stream(), _table(scalar_columns), len(scalar_columns),
                                                         scalar_columns[0].shape[0], out.ctypes.data_as(_lib.u64p)))
        return out

    def coset_ntt(self, cols: Sequence[torch.Tensor], N: int, shift, inverse: bool = False, n_in: Optional[int] = None) -> List[torch.Tensor]:
        """coset_fft (n_in coefficients -> N evaluations on shift * <omega_N>) or coset_ifft of each column."""
        n_in = cols[0].shape[0] if n_in is None else n_in
        out = [self.empty(N) for _ in cols]
        sh = _host_fe(shift)
        self.ctx.check(self.lib.plk_dev_coset_ntt(self.ctx.handle, self._stream(), _table(cols), len(cols), n_in, N,
                                                  sh.ctypes.data_as(_lib.u64p), 1 if inverse else 0, _table(out)))
        return out

    def quotient(self, numerators: Sequence[torch.Tensor], alpha, n: int, shift) -> torch.Tensor:
        """coefficients of (sum_i alpha^i numerators[i]) / (x^n - 1) from evaluation vectors on shift * <omega_N>."""
        N = numerators[0].shape[0]
        out = self.empty(N)
        a, sh = _host_fe(alpha), _host_fe(shift)
        self.ctx.check(self.lib.plk_dev_quotient(self.ctx.handle, self._stream(), _table(numerators), len(numerators),
                                                 a.ctypes.data_as(_lib.u64p), n, N, sh.ctypes.data_as(_lib.u64p), _ptr(out)))
        return out

    def gate_table_fill(self, ops: torch.Tensor, node_ids: Sequence[torch.Tensor], node_values: torch.Tensor, default_val,
                        kinds_present: Sequence[str] = AST_KINDS):
        """Gate evaluation table on device: `ops` `(n,) uint8`, node ids `(n,) int64`, `node_values` `(m, 4) int64`.
        Returns (4 wire columns, {kind: selector column})."""
        n = ops.shape[0]
        wires = [self.empty(n) for _ in range(4)]
        sels = {k: self.empty(n) for k in kinds_present}
        dv = _host_fe(default_val)
        self.ctx.check(self.lib.plk_dev_gate_table_fill(
            self.ctx.handle, self._stream(), C.cast(C.c_void_p(ops.data_ptr()), _lib.u8p), _table(node_ids), n, _ptr(node_values),
            node_values.shape[0], dv.ctypes.data_as(_lib.u64p), _table(wires), _table([sels.get(k) for k in AST_KINDS])))
        return wires, sels

    def sigma_table(self, node_ids: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """Permutation table from four `(n,) int64` node-id columns (-1 = unused wire)."""
        n = node_ids[0].shape[0]
        out = [torch.empty((n,), dtype=torch.int64, device=self.device) for _ in range(4)]
        self.ctx.check(self.lib.plk_dev_sigma_table(self.ctx.handle, self._stream(), _table(node_ids), n, _table(out)))
        return out

    def extend4(self, cols: Sequence[torch.Tensor], n: int, kinds: Optional[Sequence[int]] = None) -> List[torch.Tensor]:
        """Interpolate each column (n rows) and evaluate it on the 4n-point domain, natural order."""
        out = [self.empty(4 * n) for _ in cols]
        karr = (C.c_uint32 * len(cols))(*kinds) if kinds is not None else None
        self.ctx.check(self.lib.plk_dev_extend4(self.ctx.handle, self._stream(), _table(cols), karr, len(cols), n, None, _table(out),
                                                0, 0, 0))
        return out

    def extend4_chunked(self, cols: Sequence[torch.Tensor], n: int, kinds, parts: int, slots: int, halo: int) -> torch.Tensor:
        """The same evaluations laid out for an exchange between `parts` ranks: returns a
        `(parts, slots, 4n / parts + halo, 4)` tensor whose [h, b] holds column b's points of range h followed by the
        first `halo` points of the next range (cyclically); rows b >= len(cols) stay zero."""
        chunk = 4 * n // parts
        out = torch.zeros((parts, slots, chunk + halo, 4), dtype=torch.int64, device=self.device)
        karr = (C.c_uint32 * len(cols))(*kinds) if kinds is not None else None
        self.ctx.check(self.lib.plk_dev_extend4(self.ctx.handle, self._stream(), _table(cols), karr, len(cols), n, None,
                                                _table([out[0, b] for b in range(len(cols))]), chunk, slots * (chunk + halo), halo))
        return out

    def grand_product(self, dc: DeviceCircuit, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        z = self.empty(dc.n + 1) if out is None else out
        self.ctx.check(self.lib.plk_dev_grand_product(self.ctx.handle, self._stream(), _table(dc.wires), _table(dc.sigma), dc.n,
                                                      dc.beta.ctypes.data_as(_lib.u64p), dc.gamma.ctypes.data_as(_lib.u64p), _ptr(z)))
        return z

    def grand_product_noalloc(self, dc: DeviceCircuit, out: torch.Tensor) -> torch.Tensor:
        """The grand product through the entry point that allocates nothing; `out` is the caller's `(n + 1, 4)` tensor."""
        self.ctx.check(self.lib.plk_dev_grand_product_noalloc(self.ctx.handle, self._stream(), _table(dc.wires), _table(dc.sigma), dc.n,
                                                              dc.beta.ctypes.data_as(_lib.u64p), dc.gamma.ctypes.data_as(_lib.u64p), _ptr(out)))
        return out

    def gate_constraint_eval(self, wire_evals, sel_evals: Dict[str, torch.Tensor], N: int, out=None) -> torch.Tensor:
        out = self.empty(N) if out is None else out
        sels = [sel_evals.get(k) for k in AST_KINDS]
        self.ctx.check(self.lib.plk_dev_gate_constraint_eval(self.ctx.handle, self._stream(), _table(wire_evals), _table(sels), N, _ptr(out)))
        return out

    def perm_constraint_eval(self, z_evals, sigma_evals, wire_evals, N: int, beta, gamma, z_shift: int = 1, out=None) -> torch.Tensor:
        out = self.empty(N) if out is None else out
        b, g = _host_fe(beta), _host_fe(gamma)
        self.ctx.check(self.lib.plk_dev_perm_constraint_eval(self.ctx.handle, self._stream(), _ptr(z_evals), _table(sigma_evals),
                                                             _table(wire_evals), N, b.ctypes.data_as(_lib.u64p),
                                                             g.ctypes.data_as(_lib.u64p), z_shift, _ptr(out)))
        return out

    def round23(self, dc: DeviceCircuit, N: int, z_shift: int = 1, out=None):
        """Enqueue rounds 2+3 for one circuit; returns (z [n+1], gate [N], perm [N]) device tensors."""
        z, g, p = out if out is not None else (self.empty(dc.n + 1), self.empty(N), self.empty(N))
        wires, sels, sigma = dc.tables()
        rc = self.lib.plk_dev_round23(self.ctx.handle, self._stream(), wires, sels, sigma, dc.n, N, dc.beta.ctypes.data, dc.gamma.ctypes.data,
                                      z_shift, z.data_ptr(), g.data_ptr(), p.data_ptr())
        if rc:
            self.ctx.check(rc)
        return z, g, p

    # -- pieces of the multi-GPU paths (plonk_amd.dist)
    def grand_product_cols(self, wires, sigma, n: int, beta, gamma) -> torch.Tensor:
        z = self.empty(n + 1)
        b, g = _host_fe(beta), _host_fe(gamma)
        self.ctx.check(self.lib.plk_dev_grand_product(self.ctx.handle, self._stream(), _table(wires), _table(sigma), n,
                                                      b.ctypes.data_as(_lib.u64p), g.ctypes.data_as(_lib.u64p), _ptr(z)))
        return z

    def grand_product_local(self, wires, sigma, n: int, first: int, rows: int, beta, gamma) -> torch.Tensor:
        """First half of the grand product over the rows [first, first + rows): returns the `(2, 4)` tensor
        (numerator product, denominator product) of the range; the scans stay in the context."""
        totals = self.empty(2)
        b, g = _host_fe(beta), _host_fe(gamma)
        self.ctx.check(self.lib.plk_dev_grand_product_local(self.ctx.handle, self._stream(), _table(wires), _table(sigma), n, first, rows,
                                                            b.ctypes.data_as(_lib.u64p), g.ctypes.data_as(_lib.u64p), _ptr(totals)))
        return totals

    def grand_product_finish(self, rows: int, totals_all: torch.Tensor, mine: int) -> torch.Tensor:
        """Second half: `totals_all` is `(ranges, 2, 4)`; returns z[first + 1 .. first + rows] as `(rows, 4)`."""
        out = self.empty(rows)
        totals_all = totals_all.contiguous()
        self.ctx.check(self.lib.plk_dev_grand_product_finish(self.ctx.handle, self._stream(), rows, _ptr(totals_all),
                                                             totals_all.shape[0], mine, _ptr(out)))
        return out

    def gate_constraint_chunk(self, wire_evals, sel_evals, count: int) -> torch.Tensor:
        out = self.empty(count)
        self.ctx.check(self.lib.plk_dev_gate_constraint_eval(self.ctx.handle, self._stream(), _table(wire_evals), _table(sel_evals),
                                                             count, _ptr(out)))
        return out

    def perm_constraint_chunk(self, z_evals, sigma_evals, wire_evals, count: int, index0: int, N: int, beta, gamma,
                              z_shift: int = 1) -> torch.Tensor:
        """`count` points from `index0`; z_evals carries count + z_shift points (range + halo)."""
        out = self.empty(count)
        b, g = _host_fe(beta), _host_fe(gamma)
        self.ctx.check(self.lib.plk_dev_perm_constraint_eval_range(self.ctx.handle, self._stream(), _ptr(z_evals), _table(sigma_evals),
                                                                   _table(wire_evals), count, index0, b.ctypes.data_as(_lib.u64p),
                                                                   g.ctypes.data_as(_lib.u64p), z_shift, _ptr(out)))
        return out

    def ntt_rows(self, mat: torch.Tensor, N: int, inverse: bool = False) -> torch.Tensor:
        """Independent transforms of the rows of a contiguous `[rows, len, 4]` matrix -> `[rows, N, 4]`."""
        rows, length = mat.shape[0], mat.shape[1]
        out = torch.empty((rows, N, 4), dtype=torch.int64, device=self.device)
        self.ctx.check(self.lib.plk_dev_ntt_strided(self.ctx.handle, self._stream(), _ptr(mat), length, _lib.INPUT_FIELD, rows,
                                                    length, N, 1 if inverse else 0, _ptr(out), N))
        return out

    def fourstep_cols(self, recv1: torch.Tensor, N: int, ranks: int, rank: int, inverse: bool = False) -> torch.Tensor:
        """Phase A of a sharded transform on the buffer exchange 1 delivered; returns the send buffer of exchange 2."""
        out = torch.empty_like(recv1)
        self.ctx.check(self.lib.plk_dev_fourstep_cols(self.ctx.handle, self._stream(), _ptr(recv1), N, ranks, rank, 1 if inverse else 0, _ptr(out)))
        return out

    def fourstep_rows(self, recv2: torch.Tensor, N: int, ranks: int, rank: int, inverse: bool = False, natural_layout: bool = True) -> torch.Tensor:
        """Phase B on the buffer exchange 2 delivered; returns the send buffer of exchange 3, or the transposed block."""
        out = torch.empty_like(recv2)
        self.ctx.check(self.lib.plk_dev_fourstep_rows(self.ctx.handle, self._stream(), _ptr(recv2), N, ranks, rank, 1 if inverse else 0,
                                                      1 if natural_layout else 0, _ptr(out)))
        return out

    def mul(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(a)
        self.ctx.check(self.lib.plk_dev_mul(self.ctx.handle, self._stream(), _ptr(a), _ptr(b), a.numel() // 4, _ptr(out)))
        return out

    def twiddle_block(self, N: int, rows: int, cols: int, row0: int, col0: int = 0, inverse: bool = False,
                      scale_by_inverse_n: bool = False) -> torch.Tensor:
        out = torch.empty((rows, cols, 4), dtype=torch.int64, device=self.device)
        self.ctx.check(self.lib.plk_dev_twiddle_block(self.ctx.handle, self._stream(), N, 1 if inverse else 0,
                                                      1 if scale_by_inverse_n else 0, rows, cols, row0, col0, _ptr(out)))
        return out

    # -- per-launch timing
    def profile_begin(self):
        self.lib.plk_profile_begin(self.ctx.handle)

    def profile_end(self) -> list:
        return json.loads(self.lib.plk_profile_end(self.ctx.handle).decode())
