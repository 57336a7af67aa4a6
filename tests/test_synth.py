"""CPU tests of the synthetic-circuit generator (host logic)."""
import numpy as np

from plonk_amd import synth


def splitmix64_scalar(seed, count):
    out, s, M = [], seed & (2**64 - 1), 2**64 - 1
    for _ in range(count):
        s = (s + 0x9E3779B97F4A7C15) & M
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        out.append(z ^ (z >> 31))
    return out


def test_splitmix64_matches_scalar_definition():
    assert synth.splitmix64(0x504C4F4E4B, 64).tolist() == splitmix64_scalar(0x504C4F4E4B, 64)
    # published first outputs for seed 0 ... (well-known test vector of splitmix64)
    assert synth.splitmix64(0, 3).tolist() == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_reduce_mod_p_against_bigint():
    raw = synth.splitmix64(7, 4 * 500).reshape(500, 4)
    raw[0] = synth.P_LIMBS            # exactly p -> 0
    raw[1] = np.array([2**64 - 1] * 4, dtype=np.uint64)  # 2^256 - 1
    red = synth.reduce_mod_p(raw)
    for i in range(500):
        v = sum(int(raw[i, j]) << (64 * j) for j in range(4))
        r = sum(int(red[i, j]) << (64 * j) for j in range(4))
        assert r == v % synth.P_INT


def test_circuit_shape_and_determinism():
    a, b = synth.make_circuit(256), synth.make_circuit(256)
    assert all(np.array_equal(x, y) for x, y in zip(a.table.wires(), b.table.wires()))
    other = synth.make_circuit(256, circuit_id=1)
    assert not np.array_equal(a.table.input1, other.table.input1)
    for s in a.perm.columns():
        assert sorted(s.tolist()) == list(range(256))
    hot = sum((v != 0).any(axis=1).astype(int) for v in a.table.selectors.values())
    assert hot.tolist() == [1] * 256 and len(a.table.selectors) == 9
    assert a.beta.any() and a.gamma.any()


def test_int_conversions_roundtrip():
    vals = [0, 1, 42, synth.P_INT - 1, 123456789 << 200]
    assert synth.mont_to_ints(synth.ints_to_mont(vals)) == vals
    assert synth.ints_to_mont([1])[0].tolist() == synth.ONE_MONT.tolist()
