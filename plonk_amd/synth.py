"""Seeded synthetic circuits (host side, numpy only) for parity tests and the benchmark.

A circuit here is exactly what the hot path consumes: a `GateEvaluationTable` (4 wire columns +
one-hot selector columns) and a `PermutationTable` (4 index columns).  The reference has no
trace builder (nothing calls `ExecutionTable::push_row`, src/execution_trace.rs:27-29), so the
columns are generated directly.

Generator: splitmix64.  Stream `k` of circuit `c` is seeded with SEED_BASE + k + 1000003 * c.
  wire column k (k = 0..3): element j = outputs 4j..4j+3 as little-endian limbs, reduced mod p
      by subtraction; the reduced value is used AS the Montgomery residue (any residue < p is
      a valid element, and no conversion pass is needed at 2^24 rows).
  sigma column k (stream 4 + k): the stable argsort of n outputs — a uniform permutation of [0, n).
  selectors (stream 8): row j gets kind (output_j mod 9); that column holds one, the others zero.
  beta, gamma (streams 9, 10): one element each, forced non-zero.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .api import AST_KINDS, GateEvaluationTable, PermutationTable

SEED_BASE = 0x504C4F4E4B  # "PLONK"
CIRCUIT_STRIDE = 1000003

#: moduli of the fields the library is instantiated for, by field id (include/plonk_hip.h: PLK_FIELD_*)
FIELD_MODULI = {
    0: 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,      # BLS12-381 Fr
    1: (1 << 254) + 4707489544292117082687961190295928833,                        # Tweedle Fr (Tweedledee's scalar field)
}

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, count: int) -> np.ndarray:
    """The first `count` outputs of splitmix64 seeded with `seed` (vectorised, wrapping uint64)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


class FieldSpec:
    """The host-side arithmetic the generators need, for one field: reduction of 256-bit integers below p and
    conversions between Python integers and Montgomery limb rows."""

    def __init__(self, field_id: int):
        self.id = field_id
        self.P_INT = FIELD_MODULI[field_id]
        self.P_LIMBS = np.array([(self.P_INT >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
        self.ONE_MONT_INT = (1 << 256) % self.P_INT
        self.ONE_MONT = np.array([(self.ONE_MONT_INT >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
        self.R_INV_INT = pow(1 << 256, -1, self.P_INT)
        self._subtractions = (1 << 256) // self.P_INT          # a 256-bit integer is below (this + 1) p

    def _geq_p(self, a: np.ndarray) -> np.ndarray:
        """a >= p, limb-wise on (n, 4) uint64."""
        ge = np.ones(a.shape[0], dtype=bool)
        decided = np.zeros(a.shape[0], dtype=bool)
        for i in (3, 2, 1, 0):
            gt, lt = a[:, i] > self.P_LIMBS[i], a[:, i] < self.P_LIMBS[i]
            ge = np.where(~decided & lt, False, ge)
            decided |= gt | lt
        return ge

    def _sub_p(self, a: np.ndarray, mask: np.ndarray) -> np.ndarray:
        out = a.copy()
        borrow = np.zeros(a.shape[0], dtype=np.uint64)
        with np.errstate(over="ignore"):
            for i in range(4):
                sub = np.where(mask, self.P_LIMBS[i], np.uint64(0))
                d = a[:, i] - sub
                b1 = (a[:, i] < sub).astype(np.uint64)
                d2 = d - borrow
                b2 = (d < borrow).astype(np.uint64)
                out[:, i] = d2
                borrow = b1 | b2
        return out

    def reduce_mod_p(self, a: np.ndarray) -> np.ndarray:
        """(n, 4) uint64 256-bit integers -> reduced below p by conditional subtractions (2^256 < 3 p for BLS12-381 Fr,
        < 4 p for the Tweedle field)."""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        for _ in range(self._subtractions):
            a = self._sub_p(a, self._geq_p(a))
        return a

    def random_elements(self, seed: int, n: int) -> np.ndarray:
        return self.reduce_mod_p(splitmix64(seed, 4 * n).reshape(n, 4))

    def ints_to_mont(self, values) -> np.ndarray:
        out = np.empty((len(values), 4), dtype=np.uint64)
        for i, v in enumerate(values):
            m = (int(v) % self.P_INT) * self.ONE_MONT_INT % self.P_INT
            for j in range(4):
                out[i, j] = (m >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
        return out

    def mont_to_ints(self, a: np.ndarray):
        a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
        return [sum(int(a[i, j]) << (64 * j) for j in range(4)) * self.R_INV_INT % self.P_INT for i in range(a.shape[0])]

    def make_circuit(self, n: int, circuit_id: int = 0, kinds=AST_KINDS) -> "Circuit":
        return make_circuit(n, circuit_id, kinds, field=self)


_FIELDS = {}


def field(field_id: int = 0) -> FieldSpec:
    if field_id not in _FIELDS:
        _FIELDS[field_id] = FieldSpec(field_id)
    return _FIELDS[field_id]


# the default field (BLS12-381 Fr) under the module's own names
_F0 = field(0)
P_INT, P_LIMBS, ONE_MONT_INT, ONE_MONT = _F0.P_INT, _F0.P_LIMBS, _F0.ONE_MONT_INT, _F0.ONE_MONT
reduce_mod_p = _F0.reduce_mod_p
random_elements = _F0.random_elements


def random_permutation(seed: int, n: int) -> np.ndarray:
    return np.argsort(splitmix64(seed, n), kind="stable").astype(np.uint64)


@dataclass
class Circuit:
    n: int
    table: GateEvaluationTable
    perm: PermutationTable
    beta: np.ndarray
    gamma: np.ndarray


def make_circuit(n: int, circuit_id: int = 0, kinds=AST_KINDS, field: FieldSpec = _F0) -> Circuit:
    random_elements, ONE_MONT = field.random_elements, field.ONE_MONT
    base = SEED_BASE + CIRCUIT_STRIDE * circuit_id
    wires = [random_elements(base + k, n) for k in range(4)]
    sigmas = [random_permutation(base + 4 + k, n) for k in range(4)]
    kind_of_row = (splitmix64(base + 8, n) % np.uint64(len(AST_KINDS))).astype(np.int64)
    selectors = {}
    for k, name in enumerate(AST_KINDS):
        if name not in kinds:
            continue
        col = np.zeros((n, 4), dtype=np.uint64)
        col[kind_of_row == k] = ONE_MONT
        selectors[name] = col
    beta = random_elements(base + 9, 1)[0]
    gamma = random_elements(base + 10, 1)[0]
    if not beta.any():
        beta = ONE_MONT.copy()
    if not gamma.any():
        gamma = ONE_MONT.copy()
    return Circuit(n, GateEvaluationTable(*wires, selectors=selectors), PermutationTable(*sigmas), beta, gamma)


def make_node_ids(n: int, circuit_id: int = 0, leaves: int = 0):
    """Seeded flattened trace for the sigma-table step: four `(n,) uint64` node-id columns.

    Row j is an operation of arity 1 + (r_j mod 3) whose output is a fresh node `leaves + j`; each
    used input wire carries a uniformly chosen earlier node (a leaf or an earlier row's output), so
    nodes are shared by a varying number of positions; unused wires hold NO_WIRE.  `leaves` defaults
    to max(n // 4, 1).  Streams 16..19 of the circuit: arity, then one per input wire."""
    from .api import NO_WIRE
    leaves = leaves or max(n // 4, 1)
    base = SEED_BASE + CIRCUIT_STRIDE * circuit_id
    j = np.arange(n, dtype=np.uint64)
    arity = splitmix64(base + 16, n) % np.uint64(3) + np.uint64(1)
    cols = []
    for w in range(3):
        pick = splitmix64(base + 17 + w, n) % (np.uint64(leaves) + j)
        cols.append(np.where(arity > np.uint64(w), pick, np.uint64(NO_WIRE)))
    cols.append(np.uint64(leaves) + j)
    return cols


@dataclass
class FlatTrace:
    """A flattened execution trace: what the table-building steps before the hot path consume."""
    ops: np.ndarray          # (n,) uint8: ASTKind index | inputs << 4
    node_ids: list           # four (n,) uint64 columns, NO_WIRE on unused wires
    node_values: np.ndarray  # (m, 4) uint64 Montgomery limbs, one evaluated value per node


def make_trace(n: int, circuit_id: int = 0, leaves: int = 0) -> FlatTrace:
    """make_node_ids plus operation kinds consistent with each row's input count (1: Not, 2: one of
    Add/Sub/Mult/Div/Eq from stream 20, 3: If) and one random value per node (stream 21).  The values
    are NOT required to satisfy the gates (as in make_circuit)."""
    leaves = leaves or max(n // 4, 1)
    base = SEED_BASE + CIRCUIT_STRIDE * circuit_id
    ids = make_node_ids(n, circuit_id, leaves)
    inputs = (splitmix64(base + 16, n) % np.uint64(3) + np.uint64(1)).astype(np.uint8)
    two = (splitmix64(base + 20, n) % np.uint64(5) + np.uint64(2)).astype(np.uint8)
    kind = np.where(inputs == 1, np.uint8(7), np.where(inputs == 2, two, np.uint8(8))).astype(np.uint8)
    return FlatTrace(kind | (inputs << np.uint8(4)), ids, random_elements(base + 21, leaves + n))


# ---- conversions between Python integers and limb rows (small inputs, tests and fixtures), default field
ints_to_mont = _F0.ints_to_mont
mont_to_ints = _F0.mont_to_ints
R_INV_INT = _F0.R_INV_INT
