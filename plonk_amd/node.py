"""One node = G ranks in ONE process, one GPU each: the library's native multi-GPU entry points (`plk_node_*` in
include/plonk_hip.h).  Where `plonk_amd.dist` runs one process per GPU and exchanges through torch.distributed, this
form needs neither Python nor torch - a C or Rust host calls the same symbols; the wrapper here exists for the tests and
the benchmark, with torch tensors standing in for device memory.

The reference has no distributed code of any kind (src/polynomial_utils.rs:29 is one shared-memory transform)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .api import AST_KINDS, PlonkError
from .device import DeviceCircuit, _table


class Node:
    """`exchange`: "rccl" (one GPU per rank, ncclAllToAll) or "copy" (device-to-device copies; several ranks may share
    a GPU, which is how the sharded paths are rehearsed on a single GPU)."""

    def __init__(self, devices: Sequence[int], exchange: str = "rccl", field: int = 0):
        self.lib = _lib.load()
        self.devices = list(devices)
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        mode = {"rccl": _lib.NODE_RCCL, "copy": _lib.NODE_COPY}[exchange]
        rc = self.lib.plk_node_create_field(arr, len(self.devices), mode, field, C.byref(h))
        if rc != _lib.PLK_OK:
            raise PlonkError(rc, f"plk_node_create(devices={self.devices}, exchange={exchange}): {self.lib.plk_strerror(rc).decode()}")
        self.handle = h
        self.G = len(self.devices)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.plk_node_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc != _lib.PLK_OK:
            msg = self.lib.plk_node_last_error(self.handle).decode() or self.lib.plk_strerror(rc).decode()
            raise PlonkError(rc, msg)

    def device(self, g: int) -> torch.device:
        return torch.device("cuda", self.devices[g])

    def sync(self):
        self.check(self.lib.plk_node_sync(self.handle))

    def ntt(self, shards: Sequence[torch.Tensor], N: int, inverse: bool = False, natural_output: bool = True,
            out: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
        """One transform of N points; shards[g] is rank g's `(N / G, 4) int64` tensor on its GPU, complete before the
        call (the ranks' streams know nothing of torch's).  Returns the output shards; wait with `sync()`."""
        out = [torch.empty_like(s) for s in shards] if out is None else list(out)
        for g, s in enumerate(shards):
            torch.cuda.synchronize(s.device)
        self.check(self.lib.plk_node_ntt(self.handle, _table(shards), N, 1 if inverse else 0, 1 if natural_output else 0, _table(out)))
        return out

    def ntt_host(self, x: np.ndarray, inverse: bool = False) -> np.ndarray:
        """The transform of a host array `(N, 4) uint64`, natural order in and out."""
        x = np.ascontiguousarray(x, dtype=np.uint64)
        out = np.empty_like(x)
        self.check(self.lib.plk_node_ntt_host(self.handle, x.ctypes.data_as(_lib.u64p), x.shape[0], 1 if inverse else 0,
                                              out.ctypes.data_as(_lib.u64p)))
        return out

    def _io(self, circuits: Sequence[DeviceCircuit], outs):
        arr = (_lib.CircuitIO * len(circuits))()
        for io, dc, (z, g, p) in zip(arr, circuits, outs):
            for k in range(4):
                io.wires[k] = dc.wires[k].data_ptr()
                io.sigma[k] = dc.sigma[k].data_ptr()
            for k, kind in enumerate(AST_KINDS):
                t = dc.selectors.get(kind)
                io.sel[k] = t.data_ptr() if t is not None else None
            for k in range(4):
                io.beta[k] = int(dc.beta[k])
                io.gamma[k] = int(dc.gamma[k])
            io.z_out, io.gate_out, io.perm_out = z.data_ptr(), g.data_ptr(), p.data_ptr()
        for d in set(self.devices):
            torch.cuda.synchronize(d)
        return arr

    def round23_batch(self, circuits: Sequence[DeviceCircuit], N: int, outs, z_shift: int = 1):
        """Rounds 2+3 of `circuits[i]` (resident on rank i % G's GPU) into `outs[i] = (z, gate, perm)` tensors on the
        same GPU; returns when all are done."""
        arr = self._io(circuits, outs)
        self.check(self.lib.plk_node_round23_batch(self.handle, arr, len(circuits), circuits[0].n, N, z_shift))
        return outs

    def round23(self, copies: Sequence[DeviceCircuit], N: int, outs, z_shift: int = 1):
        """Rounds 2+3 of ONE circuit over the G ranks: `copies[g]` is the circuit resident on rank g's GPU,
        `outs[g] = (z of n + 1, gate of N / G, perm of N / G)` tensors there - rank g's range of the constraint vectors."""
        assert len(copies) == self.G
        arr = self._io(copies, outs)
        self.check(self.lib.plk_node_round23(self.handle, arr, copies[0].n, N, z_shift))
        return outs

    def round23_host(self, circuit, N: int, z_shift: int = 1):
        """Rounds 2+3 of one circuit given as host arrays (`plonk_amd.synth.Circuit`), over the node's ranks: returns
        (z, gate, perm) as host arrays, the shapes `plonk_amd.api.prove_round23` returns."""
        n = circuit.n
        keep = []

        def table(arrays):
            arr = (C.c_void_p * len(arrays))()
            for i, a in enumerate(arrays):
                if a is None:
                    arr[i] = None
                    continue
                a = np.ascontiguousarray(a, dtype=np.uint64)
                keep.append(a)
                arr[i] = a.ctypes.data
            return C.cast(arr, _lib.u64pp)

        beta = np.ascontiguousarray(circuit.beta, dtype=np.uint64).reshape(4)
        gamma = np.ascontiguousarray(circuit.gamma, dtype=np.uint64).reshape(4)
        z = np.empty((n + 1, 4), dtype=np.uint64)
        gate = np.empty((N, 4), dtype=np.uint64)
        perm = np.empty((N, 4), dtype=np.uint64)
        self.check(self.lib.plk_node_round23_host(
            self.handle, table(circuit.table.wires()), table([circuit.table.selectors.get(k) for k in AST_KINDS]),
            table(circuit.perm.columns()), n, N, beta.ctypes.data_as(_lib.u64p), gamma.ctypes.data_as(_lib.u64p), z_shift,
            z.ctypes.data_as(_lib.u64p), gate.ctypes.data_as(_lib.u64p), perm.ctypes.data_as(_lib.u64p)))
        return z, gate, perm
