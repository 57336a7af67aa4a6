"""Interval replay of the point formulas of plonk_amd/csrc/fq28.cuh.

The 28-bit-limb field arithmetic is lazy: limbs may exceed 28 bits and values may exceed q between
multiplications.  Every operation there has preconditions (stated in the header of fq28.cuh); this test replays
the three point formulas on (largest limb, largest value / q) pairs and asserts each precondition, and that the
outputs satisfy the bounds the inputs were assumed to have (a fixed point: coordinates normalised, below 7.75 q)."""
from fractions import Fraction as Fr

Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R_OVER_Q = Fr(1 << 392, Q)              # a b / 2^392 = (alpha beta / R_OVER_Q) q for a = alpha q, b = beta q
L28 = 1 << 28
MAX_Q_LIMB = L28 - 1


class V:
    """limb: exclusive bound of every limb; val: exclusive bound of the value in units of q"""
    def __init__(self, limb, val):
        self.limb, self.val = limb, Fr(val)


def mul(a, b):
    # one column: 14 products a_i b_j, 14 products m_i q_j, the carry of the column before
    assert 14 * (a.limb - 1) * (b.limb - 1) + 14 * (L28 - 1) * MAX_Q_LIMB + (1 << 37) < 1 << 64, "column overflow"
    assert a.val * b.val <= 1800, "product too large for the output bound"
    out = a.val * b.val / R_OVER_Q + 1
    assert out <= Fr(7, 4)
    return V(L28, Fr(7, 4))


def sqr(a):
    assert 2 * (a.limb - 1) < 1 << 32, "doubled operand overflows a limb"
    return mul(a, a)


def add(a, b):
    assert a.limb + b.limb <= 1 << 32
    return V(a.limb + b.limb, a.val + b.val)


def shl(a, s):
    assert a.limb << s <= 1 << 32
    return V(a.limb << s, a.val * (1 << s))


def sub(k, s, a, b):
    """a + (k q with limbs raised by 2^s) - b"""
    assert b.limb <= 1 << s, "subtrahend limbs exceed the raised bias"
    assert b.val <= k - Fr(1, 4), "subtrahend value too close to the bias (top limb could go negative)"
    limb = a.limb + MAX_Q_LIMB + (1 << s) + 1
    assert limb <= 1 << 32
    return V(limb, a.val + k)


def normalize(a):
    assert a.val * Q < 1 << 392 + 4      # the top limb takes what is above 2^364
    return V(L28, a.val)


COORD = V(L28, Fr(31, 4))               # stored coordinates
AFFINE = V(L28, Fr(7, 4))               # table entries come out of a multiplication
ONE = V(L28, 1)


def check_output(p):
    for c in p:
        assert c.limb <= COORD.limb and c.val <= COORD.val, (c.limb, c.val)


def double(x, y, z):
    A, B = sqr(x), sqr(y)
    B4 = shl(B, 2)
    D = mul(x, B4)
    E = normalize(add(shl(A, 1), A))
    x3 = normalize(sub(4, 29, sqr(E), shl(D, 1)))
    C8 = mul(shl(B, 1), B4)
    y3 = normalize(sub(2, 28, mul(E, sub(8, 28, D, x3)), C8))
    z3 = normalize(shl(mul(y, z), 1))
    return x3, y3, z3


def add_affine(x1, y1, z1, x2, y2):
    Z1Z1 = sqr(z1)
    U2 = mul(x2, Z1Z1)
    S2 = mul(mul(y2, z1), Z1Z1)
    H = sub(8, 28, U2, x1)
    HH = sqr(H)
    rr = normalize(shl(sub(8, 28, S2, y1), 1))
    mul(rr, ONE)                                        # the zero test of the rare branch
    I = shl(HH, 2)
    J, Vv = mul(H, I), mul(x1, I)
    x3 = normalize(sub(4, 29, sub(2, 28, sqr(rr), J), shl(Vv, 1)))
    y3 = normalize(sub(4, 29, mul(rr, sub(8, 28, Vv, x3)), shl(mul(y1, J), 1)))
    z3 = normalize(sub(2, 28, sub(2, 28, sqr(add(z1, H)), Z1Z1), HH))
    return x3, y3, z3


def add_full(x1, y1, z1, x2, y2, z2):
    Z1Z1, Z2Z2 = sqr(z1), sqr(z2)
    U1, U2 = mul(x1, Z2Z2), mul(x2, Z1Z1)
    S1, S2 = mul(mul(y1, z2), Z2Z2), mul(mul(y2, z1), Z1Z1)
    H = sub(2, 28, U2, U1)
    HH = sqr(H)
    rr = normalize(shl(sub(2, 28, S2, S1), 1))
    mul(rr, ONE)
    I = shl(HH, 2)
    J, Vv = mul(H, I), mul(U1, I)
    x3 = normalize(sub(4, 29, sub(2, 28, sqr(rr), J), shl(Vv, 1)))
    y3 = normalize(sub(4, 29, mul(rr, sub(8, 28, Vv, x3)), shl(mul(S1, J), 1)))
    z3 = mul(sub(2, 28, sub(2, 28, sqr(add(z1, z2)), Z1Z1), Z2Z2), H)
    return x3, y3, z3


def test_doubling_stays_in_bounds():
    check_output(double(COORD, COORD, COORD))


def test_mixed_addition_stays_in_bounds():
    check_output(add_affine(COORD, COORD, COORD, AFFINE, AFFINE))


def test_full_addition_stays_in_bounds():
    check_output(add_full(COORD, COORD, COORD, COORD, COORD, COORD))


def test_conversions_stay_in_bounds():
    mul(V(L28, 1), V(L28, 1))            # ABI -> working form: canonical value times a canonical constant
    mul(COORD, V(L28, 1))                # working form -> ABI


def test_header_constants():
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plonk_amd", "csrc", "fq28.cuh")).read()

    def limbs(name):
        body = re.search(r"constexpr uint32_t " + name + r"\[14\] = \{(.*?)\}", src, re.S).group(1)
        return sum(int(t.strip().rstrip("u"), 16) << (28 * i) for i, t in enumerate(body.split(",")))

    assert limbs("Q") == Q
    assert limbs("ONE") == pow(2, 392, Q)
    assert limbs("TO_INTERNAL") == pow(2, 400, Q)
    assert limbs("TO_ABI") == pow(2, 384, Q)
    inv = int(re.search(r"constexpr uint32_t INV = (0x[0-9a-f]+)u", src).group(1), 16)
    assert (Q * inv + 1) % (1 << 28) == 0
