#!/usr/bin/env python3
"""Derives, from a prime modulus alone, every constant a field instantiation of the library needs, and prints them as the
C++ struct plonk_amd/csrc/field.cuh holds and as the table oracle/plonk_oracle.py holds.  Nothing here comes from the
reference or from arkworks: the convention followed is ark-ff 0.3's as published (Montgomery radix 2^256;
TWO_ADIC_ROOT_OF_UNITY = GENERATOR^((p - 1) / 2^s) with s the two-adicity), and GENERATOR is taken as the smallest
quadratic non-residue that generates the whole 2-Sylow subgroup (for the fields below it also generates the full group
when the co-factor's prime factors are checked, which needs a factorisation not attempted here).  For BLS12-381 Fr the
result (generator 7, root 0x16a2a19e...) equals the widely published constants, which is the check of this script; for the
Tweedle fields no published ark-ff constants are available offline, so their generator / root are SELF-DERIVED and parity
with arkworks is unverifiable.

    python tools/derive_field_constants.py            # prints all fields
"""
import struct

FIELDS = {
    # name: (modulus, generator or None = smallest suitable)
    "Bls12_381Fr": (0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, 7),
    # Tweedledee's scalar field = Tweedledum's base field: 2^254 + 4707489544292117082687961190295928833
    "TweedleFr": ((1 << 254) + 4707489544292117082687961190295928833, None),
    # Tweedledee's base field = Tweedledum's scalar field: 2^254 + 4707489545178046908921067385359695873
    "TweedleFq": ((1 << 254) + 4707489545178046908921067385359695873, None),
}


def is_probable_prime(n: int) -> bool:
    """Miller-Rabin with the first 40 primes as bases (deterministic far beyond 2^64; for 255-bit n a proof of
    compositeness for composites with overwhelming probability, which is what a constants script needs)."""
    small = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113,
             127, 131, 137, 139, 149, 151, 157, 163, 167, 173]
    if n < 2:
        return False
    for q in small:
        if n % q == 0:
            return n == q
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in small:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def limbs32(x: int, n: int = 8):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def float_floor(x_num: int, x_den: int) -> float:
    """The largest IEEE single not above x_num / x_den (a positive rational)."""
    lo, hi = 0, (1 << 31) - 1          # positive floats order like their bit patterns
    while lo < hi:
        mid = (lo + hi + 1) // 2
        f = struct.unpack("<f", struct.pack("<I", mid))[0]
        num, den = f.as_integer_ratio()
        if num * x_den <= x_num * den:
            lo = mid
        else:
            hi = mid - 1
    return struct.unpack("<f", struct.pack("<I", lo))[0]


def derive(name: str, p: int, gen):
    assert is_probable_prime(p), name
    assert p.bit_length() <= 255 and p % (1 << 29) == 1, "the 9 x 29-bit transform arithmetic needs p < 2^255, p = 1 mod 2^29"
    s = ((p - 1) & -(p - 1)).bit_length() - 1
    if gen is None:
        gen = next(g for g in range(2, 1000) if pow(g, (p - 1) // 2, p) == p - 1)
    assert pow(gen, (p - 1) // 2, p) == p - 1
    root = pow(gen, (p - 1) >> s, p)
    assert pow(root, 1 << (s - 1), p) == p - 1          # order exactly 2^s
    R = 1 << 256
    out = {
        "name": name, "p": p, "two_adicity": s, "generator": gen, "root": root, "one": R % p, "r2": R * R % p,
        "inv32": (-pow(p, -1, 1 << 32)) % (1 << 32), "inv64": (-pow(p, -1, 1 << 64)) % (1 << 64),
        # fe_reduce_f9's quotient estimate: (2^232 / p) (1 - 2^-20), rounded down to a float
        "qest": float_floor((1 << 232) * ((1 << 20) - 1), p * (1 << 20)),
    }
    return out


def cxx(c) -> str:
    def arr(x):
        w = limbs32(x)
        return "{" + ", ".join(f"0x{v:08x}u" for v in w[:4]) + ",\n" + " " * 37 + ", ".join(f"0x{v:08x}u" for v in w[4:]) + "}"
    return (f"struct {c['name']} {{\n"
            f"  static constexpr uint32_t P[8]  = {arr(c['p'])};\n"
            f"  static constexpr uint32_t ONE[8] = {arr(c['one'])};\n"
            f"  static constexpr uint32_t R2[8] = {arr(c['r2'])};\n"
            f"  static constexpr uint32_t INV32 = 0x{c['inv32']:08x}u;\n"
            f"  static constexpr int TWO_ADICITY = {c['two_adicity']};\n"
            f"  static constexpr uint32_t ROOT[8] = {arr(c['root'])};   // {c['generator']}^((p-1)/2^{c['two_adicity']})\n"
            f"  static constexpr float QEST = {c['qest'].hex()}f;\n}};\n")


if __name__ == "__main__":
    for name, (p, gen) in FIELDS.items():
        c = derive(name, p, gen)
        print(f"# {name}: p = 0x{c['p']:064x}\n#   two-adicity {c['two_adicity']}, generator {c['generator']}, root 0x{c['root']:064x}")
        print(f"#   R mod p 0x{c['one']:064x}\n#   R^2 mod p 0x{c['r2']:064x}\n#   -1/p mod 2^64 0x{c['inv64']:016x}, qest {c['qest'].hex()}")
        print(cxx(c))
