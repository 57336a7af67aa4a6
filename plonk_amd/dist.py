"""Multi-GPU forms of the hot path: one process per GPU, torch.distributed (RCCL on GPUs, gloo on
CPU for the tests) for the exchanges, libplonk_hip.so for every local computation.

Stream protocol: the library enqueues every kernel on torch's CURRENT stream (plonk_amd.device passes
`torch.cuda.current_stream()` to each plk_dev_* call; side streams the library uses internally are joined back into
it before the call returns).  torch.distributed's RCCL process group runs a collective on a stream of its own, which
it orders behind the current stream when the collective is issued, and - for the blocking form used here - makes the
current stream wait for before the call returns.  So a kernel, the collective that sends its output and the kernel
that consumes the received buffer run in issue order with no host synchronisation, and the caching allocator is told
of the second stream by the process group itself.  Any other backend is handed host copies (`_staged`).

Three ways the path shards (the reference has no distribution of any kind):

* `shard_circuits`  - batched proofs: whole circuits round-robin over ranks, no collective.
* `four_step_ntt`   - ONE transform of size N over G ranks.  N = N1 * N2 is viewed as an
  N1 x N2 matrix (n = n1 * N2 + n2, k = k1 + N1 * k2).  Ranks hold contiguous index ranges, i.e.
  blocks of rows.  all-to-all -> column transforms (size N1) -> twiddle omega_N^(k1 n2) ->
  all-to-all -> row transforms (size N2) -> all-to-all back to natural order.  Each all-to-all
  moves (G-1)/G of a rank's N/G elements, one message per peer - on MI355X one per xGMI link.
* `round23_poly_sharded` - ONE proof over G ranks: the 18 columns are dealt to the ranks for
  interpolation and extension, one all-to-all turns "my columns, all points" into "all columns,
  my points" (plus a one-point halo of Z), and every rank evaluates the constraints on its range.

The local computations come from an `engine` object (plonk_amd.device.DeviceProver on GPUs); the
tensors are `(n, 4) int64` rows of Montgomery limbs on the engine's device.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import os

import torch
import torch.distributed as dist

#: PLONK_DIST_FORCE_COLLECTIVES=1 issues the collectives even in a one-rank group, where they move nothing: a way
#: to run RCCL itself, and its ordering against the library's kernels, on a machine with a single GPU
_FORCE = os.environ.get("PLONK_DIST_FORCE_COLLECTIVES") == "1"


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _staged(t: torch.Tensor, group=None) -> bool:
    """RCCL runs a collective on GPU tensors in stream order.  Any other backend (gloo, used to rehearse
    several ranks on one GPU) is given host copies instead: its own handling of GPU tensors hangs
    on this ROCm/PyTorch pair whenever kernels are in flight, with or without this library
    (tools/gloo_a2a_probe.py ... heavy), and now and then even on an idle device."""
    return t.is_cuda and dist.get_backend(group) != "nccl"


def _all_to_all(send: torch.Tensor, group=None) -> torch.Tensor:
    """send[h] goes to rank h; returns recv with recv[g] = what rank g sent to us."""
    _, world = _world(group)
    if world == 1 and not (_FORCE and dist.is_initialized()):
        return send
    if _staged(send, group):
        host = send.contiguous().cpu()
        recv = torch.empty_like(host)
        dist.all_to_all_single(recv, host, group=group)
        return recv.to(send.device)
    send = send.contiguous()                 # (the collective takes contiguous buffers on both sides)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    return recv


def _all_gather(t: torch.Tensor, group=None) -> List[torch.Tensor]:
    """every rank's `t` (same shape everywhere), in rank order."""
    _, world = _world(group)
    if _staged(t, group):
        host = t.contiguous().cpu()
        out = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(out, host, group=group)
        return [o.to(t.device) for o in out]
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t.contiguous(), group=group)
    return out


def shard_circuits(num_circuits: int, group=None) -> List[int]:
    """Circuit ids this rank proves in batched-proof mode (round-robin; no data-path collective)."""
    rank, world = _world(group)
    return list(range(rank, num_circuits, world))


def split_domain(N: int):
    """N = N1 * N2 with N1 >= N2, both powers of two."""
    log_n = N.bit_length() - 1
    if N <= 0 or (1 << log_n) != N:
        raise ValueError("N must be a power of two")
    n1 = 1 << ((log_n + 1) // 2)
    return n1, N // n1


def four_step_ntt(x_local: torch.Tensor, N: int, engine, group=None, inverse: bool = False,
                  natural_output: bool = True) -> torch.Tensor:
    """Transform of size N whose input is sharded contiguously by index: this rank passes
    elements [rank * N/G, (rank + 1) * N/G).  Returns the same range of the result
    (natural_output=True) or, saving the last exchange, the block [k1 in my rows][all k2] holding
    X[k1 + N1 * k2] as an `(N1/G, N2, 4)` tensor.

    With an engine that has the fused phases (`fourstep_cols` / `fourstep_rows`, the HIP library) the local kernels
    read what an exchange delivered and write what the next one sends: the only copies left are the pack of the
    caller's contiguous input before exchange 1 and, for natural order, the transposition after exchange 3.
    Bytes per exchange and rank: (G - 1) / G of its N / G elements, 32 bytes each, one message per peer."""
    rank, G = _world(group)
    N1, N2 = split_domain(N)
    if N1 % G or N2 % G:
        raise ValueError(f"{G} ranks do not divide the {N1} x {N2} view of the domain")
    r1, c2 = N1 // G, N2 // G
    if x_local.shape != (N // G, 4):
        raise ValueError(f"expected a ({N // G}, 4) shard, got {tuple(x_local.shape)}")
    if not _fused_four_step_ok(engine, N1, N2, G):
        return _four_step_ntt_unfused(x_local, N, engine, group, inverse, natural_output)

    # exchange 1: my rows, everyone's columns  ->  everyone's rows, my columns.  The receive buffer, blocks in source
    # order, is the row-major matrix [n1][my c2 columns].
    send = x_local.view(r1, G, c2, 4).permute(1, 0, 2, 3).contiguous() if G > 1 else x_local
    recv = _all_to_all(send, group)
    send2 = engine.fourstep_cols(recv.view(N // G, 4), N, G, rank, inverse=inverse)        # [k1][c2], twiddled
    # exchange 2: destination h's rows k1 in [h r1, (h + 1) r1) are one contiguous block
    recv2 = _all_to_all(send2.view(G, r1 * c2, 4), group)                                 # [source][k1 local][c2]
    out = engine.fourstep_rows(recv2.view(N // G, 4), N, G, rank, inverse=inverse, natural_layout=natural_output)
    if not natural_output:
        return out.view(r1, N2, 4)                                                         # [my k1][k2] = X[k1 + N1 k2]
    # exchange 3: [destination h][k2 local][k1 local] -> [source g][k2 local][k1 local]; k = (g r1 + k1 local) + N1 k2
    recv3 = _all_to_all(out.view(G, c2 * r1, 4), group)
    if G == 1:
        return recv3.view(N // G, 4)
    return recv3.view(G, c2, r1, 4).permute(1, 0, 2, 3).contiguous().view(N // G, 4)


def _fused_four_step_ok(engine, N1: int, N2: int, G: int) -> bool:
    """The library's fused phases need a power-of-two number of ranks and, for sides above 2^7, at least
    2^floor(log2(N2) / 2) columns per rank (plk_dev_fourstep_cols in plonk_hip.h)."""
    if not (hasattr(engine, "fourstep_cols") and hasattr(engine, "fourstep_rows")) or G & (G - 1):
        return False
    m2 = N2.bit_length() - 1
    low = 0 if m2 <= 7 else m2 - (m2 + 1) // 2
    return (N2 // G) >= (1 << low) and N1 >= 4 and N2 >= 4


def _four_step_ntt_unfused(x_local: torch.Tensor, N: int, engine, group=None, inverse: bool = False,
                           natural_output: bool = True) -> torch.Tensor:
    """The same transform from generic pieces (independent row transforms, an element-wise product, torch copies for
    every change of layout): what an engine without the fused phases runs, e.g. the CPU oracle engine of the tests."""
    rank, G = _world(group)
    N1, N2 = split_domain(N)
    r1, c2 = N1 // G, N2 // G

    # exchange 1: my rows, everyone's columns  ->  everyone's rows, my columns
    send = x_local.view(r1, G, c2, 4).permute(1, 0, 2, 3).contiguous()
    recv = _all_to_all(send, group)
    cols = recv.view(N1, c2, 4).permute(1, 0, 2).contiguous()           # [my column][n1]
    cols = engine.ntt_rows(cols, N1, inverse=inverse)                   # [my column][k1]
    # this rank's twiddle block is a constant of (N, rank, G, direction): built once, kept on the engine
    cache = engine.__dict__.setdefault("_four_step_twiddles", {})
    key = (N, rank, G, inverse)
    if key not in cache:
        cache[key] = engine.twiddle_block(N, rows=c2, cols=N1, row0=rank * c2, col0=0, inverse=inverse)
    tw = cache[key]
    cols = engine.mul(cols, tw)                                         # * omega_N^(n2 * k1)

    # exchange 2: my columns (n2), everyone's k1  ->  my k1 rows, every n2
    send = cols.view(c2, G, r1, 4).permute(1, 0, 2, 3).contiguous()
    recv = _all_to_all(send, group)
    rows = recv.view(N2, r1, 4).permute(1, 0, 2).contiguous()           # [my k1][n2]
    rows = engine.ntt_rows(rows, N2, inverse=inverse)                   # [my k1][k2] = X[k1 + N1 k2]
    if not natural_output:
        return rows

    # exchange 3: back to contiguous ranges of k = k1 + N1 * k2 (k2 is the slow digit)
    send = rows.view(r1, G, c2, 4).permute(1, 0, 2, 3).contiguous()
    recv = _all_to_all(send, group)
    return recv.view(N1, c2, 4).permute(1, 0, 2).contiguous().view(N // G, 4)   # [my k2][k1]


def grand_product_sharded(engine, wires: Sequence[torch.Tensor], sigma: Sequence[torch.Tensor], n: int, beta, gamma,
                          group=None) -> torch.Tensor:
    """The grand product Z (n + 1 values, replicated on return) with the rows split over the ranks.
    Every rank holds the whole columns (the numerator reads any row) and owns a contiguous block of
    rows: local terms and scans -> all-gather of one (numerator, denominator) product pair per rank
    (64 bytes each) -> carries and the one inversion, locally -> all-gather of the z blocks."""
    rank, G = _world(group)
    if G == 1 or n < G:
        return engine.grand_product_cols(wires, sigma, n, beta, gamma)
    per = (n + G - 1) // G
    first = min(rank * per, n)
    rows = min(per, n - first)
    one = engine.grand_product_cols([w[:0] for w in wires], [s[:0] for s in sigma], 0, beta, gamma)   # z of no rows = [1]
    if rows:
        totals = engine.grand_product_local(wires, sigma, n, first, rows, beta, gamma)
    else:                                          # a trailing rank without rows contributes the neutral pair
        totals = torch.cat([one, one])
    all_totals = _all_gather(totals, group)
    block = torch.zeros((per, 4), dtype=totals.dtype, device=totals.device)
    if rows:
        block[:rows] = engine.grand_product_finish(rows, torch.stack(all_totals), rank)
    blocks = _all_gather(block, group)
    return torch.cat([one] + blocks)[:n + 1].contiguous()


def msm_sharded(engine, scalars_local: torch.Tensor, group=None):
    """One multi-scalar multiplication with its terms dealt to the ranks: every rank has set ITS shard of the bases
    (`engine.msm_set_bases`) and passes the matching shard of the scalars; the partial sums (96 bytes each) are
    all-gathered and added up on the host.  Returns the affine point `(12,) uint64` on every rank."""
    import numpy as np
    part = np.ascontiguousarray(engine.msm(scalars_local), dtype=np.uint64)
    t = torch.from_numpy(part.view(np.int64).copy()).to(scalars_local.device)
    parts = _all_gather(t, group)
    return engine.g1_sum(np.stack([p.cpu().numpy().view(np.uint64) for p in parts]))


def assign_columns(num_columns: int, group=None) -> List[int]:
    """Columns (0..num_columns-1) this rank interpolates and extends in `round23_poly_sharded`."""
    rank, world = _world(group)
    return list(range(rank, num_columns, world))


def round23_poly_sharded(engine, wires: Sequence[torch.Tensor], selectors: Sequence[Optional[torch.Tensor]],
                         sigma: Sequence[torch.Tensor], beta, gamma, n: int, N: int, z_shift: int = 1, group=None):
    """Rounds 2+3 of ONE circuit over all ranks.  Every rank holds the input columns (n rows each).
    Returns (z [n+1] replicated, gate-constraint chunk, permutation-constraint chunk) where the
    chunks cover evaluation points [rank * N/G, (rank + 1) * N/G).

    Column numbering: wires 0-3, selectors 4-12 (ASTKind order, None = absent), sigma 13-16, Z 17."""
    rank, G = _world(group)
    if N % G:
        raise ValueError("ranks must divide the evaluation domain")
    chunk = N // G
    z = grand_product_sharded(engine, wires, sigma, n, beta, gamma, group)   # rows split over the ranks, z replicated
    cols: List[Optional[torch.Tensor]] = list(wires) + list(selectors) + list(sigma) + [z[:n]]
    kinds = [0] * 13 + [1] * 4 + [0]
    present = [i for i, c in enumerate(cols) if c is not None]
    mine = [present[i] for i in assign_columns(len(present), group)]
    per_rank = (len(present) + G - 1) // G

    # my columns: interpolate, extend to N points, laid out [destination rank][slot][its points + halo of z_shift points]
    my_cols, my_kinds = [cols[i].contiguous() for i in mine], [kinds[i] for i in mine]
    if N == 4 * n and n >= 4 and z_shift <= chunk and hasattr(engine, "extend4_chunked"):
        # three n-point coset transforms per column, written straight into the exchange layout
        send = engine.extend4_chunked(my_cols, n, my_kinds, G, per_rank, z_shift)
    else:
        evals = torch.zeros((per_rank, N + z_shift, 4), dtype=torch.int64, device=z.device)
        if mine:
            ext = engine.ntt_batch(engine.ntt_batch(my_cols, n, inverse=True, kinds=my_kinds), N)
            for slot, e in enumerate(ext):
                evals[slot, :N] = e
                evals[slot, N:] = e[:z_shift]                            # wrap-around halo for z[(i + shift) mod N]
        send = torch.stack([evals[:, h * chunk:(h + 1) * chunk + z_shift] for h in range(G)]).contiguous()
    # exchange: my slots, every rank's points  ->  every rank's slots, my points (+ halo)
    recv = _all_to_all(send, group)                                      # [source rank][slot][chunk+halo]
    local = {}
    for src in range(G):
        src_cols = [present[i] for i in range(src, len(present), G)]
        for slot, col in enumerate(src_cols):
            local[col] = recv[src, slot]
    w_ev = [local[i][:chunk].contiguous() for i in range(4)]
    s_ev = [local[4 + k][:chunk].contiguous() if (4 + k) in local else None for k in range(9)]
    g_ev = [local[13 + k][:chunk].contiguous() for k in range(4)]
    z_ev = local[17].contiguous()                                        # chunk + halo points
    gate = engine.gate_constraint_chunk(w_ev, s_ev, chunk)
    perm = engine.perm_constraint_chunk(z_ev, g_ev, w_ev, chunk, rank * chunk, N, beta, gamma, z_shift)
    return z, gate, perm
