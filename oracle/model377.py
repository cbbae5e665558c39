"""Pure-Python bigint model of BLS12-377 G1 (y^2 = x^3 + 1 over the 377-bit base field) for BASELINE config 5.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference holds NO code, test or vector for this curve (only prose,
README.md:57-73,279-287,325-349: the moduli, the curve equation, 48-byte little-endian coordinates, 96n-byte point and
48n-byte scalar buffers, "projective algorithms" for the group law, the same cuZK pipeline).  What pins this model is
public mathematics: the BLS12 parameterisation (q, r, trace and cofactor all follow from the seed x = 0x8508c00000000001
and are checked in tests/test_oracle_bls377.py), the curve equation, and the group axioms (r*G = O, associativity,
naive sum == Pippenger).  The pipeline follows the Twisted-Edwards one of oracle/model.py step for step (signed
digits miscellaneous/utils.ts:52-95, bucket sums smvp.ts:37-102, running-sum reduction bpr.ts:4-131, Horner
submission.ts:369-407) with the group law swapped, as README.md:279-287 describes.
"""
from __future__ import annotations

Q = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177
R_ORDER = 8444461749428370424248824938781546531375899335154063827935233455917409239041   # scalar field = subgroup order
SEED_X = 0x8508C00000000001
COFACTOR = (SEED_X - 1) ** 2 // 3
B_COEFF = 1
# generator of the order-r subgroup (the standard one of the BLS12-377 specification; r*G = O is tested)
GX = 81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695
GY = 241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030
G = (GX, GY)
INF = None      # point at infinity (affine model)
COORD_BYTES, SCALAR_BYTES = 48, 48


def inv(a: int) -> int:
    return pow(a % Q, Q - 2, Q)


def on_curve(pt) -> bool:
    if pt is INF:
        return True
    x, y = pt
    return (y * y - x * x * x - B_COEFF) % Q == 0


def neg(pt):
    return INF if pt is INF else (pt[0], (-pt[1]) % Q)


def add(p1, p2):
    """affine chord-and-tangent law with every special case"""
    if p1 is INF:
        return p2
    if p2 is INF:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return INF
        lam = 3 * x1 * x1 * inv(2 * y1) % Q
    else:
        lam = (y2 - y1) * inv(x2 - x1) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return (x3, (lam * (x1 - x3) - y1) % Q)


# projective (X : Y : Z), complete formulas (Renes-Costello-Batina 2016, a = 0, b3 = 3b) -- fast path for scalar_mul
def _padd(p1, p2):
    X1, Y1, Z1 = p1
    X2, Y2, Z2 = p2
    b3 = 3 * B_COEFF
    t0, t1, t2 = X1 * X2 % Q, Y1 * Y2 % Q, Z1 * Z2 % Q
    t3 = ((X1 + Y1) * (X2 + Y2) - t0 - t1) % Q
    t4 = ((Y1 + Z1) * (Y2 + Z2) - t1 - t2) % Q
    y3 = ((X1 + Z1) * (X2 + Z2) - t0 - t2) % Q
    t0 = 3 * t0 % Q
    t2 = b3 * t2 % Q
    z3 = (t1 + t2) % Q
    t1 = (t1 - t2) % Q
    y3 = b3 * y3 % Q
    X3 = (t3 * t1 - t4 * y3) % Q
    Y3 = (t1 * z3 + y3 * t0) % Q
    Z3 = (z3 * t4 + t0 * t3) % Q
    return (X3, Y3, Z3)


def _to_affine(p):
    X, Y, Z = p
    if Z % Q == 0:
        return INF
    zi = inv(Z)
    return (X * zi % Q, Y * zi % Q)


def scalar_mul(k: int, pt):
    """k*pt over the integer k"""
    if pt is INF or k == 0:
        return INF
    acc = (0, 1, 0)
    base = (pt[0], pt[1], 1)
    while k:
        if k & 1:
            acc = _padd(acc, base)
        base = _padd(base, base)
        k >>= 1
    return _to_affine(acc)


def msm_naive(points, scalars):
    acc = INF
    for pt, k in zip(points, scalars):
        acc = add(acc, scalar_mul(k, pt))
    return acc


def decompose_scalar_signed(s: int, num_words: int, c: int):
    """miscellaneous/utils.ts:52-95: stored digit = signed digit + 2^(c-1)"""
    half, full, carry, out = 1 << (c - 1), 1 << c, 0, []
    for w in range(num_words):
        v = ((s >> (c * w)) & (full - 1)) + carry
        carry = 0
        if v >= half:
            v -= full
            carry = 1
        out.append(v + half)
    if carry:
        raise ValueError("final carry is 1")
    return out


def msm_pipeline(points, scalars, c: int):
    """the cuZK pipeline of the reference with this curve's group law (README.md:279-287)"""
    W = (256 + c - 1) // c
    half = 1 << (c - 1)
    digits = [decompose_scalar_signed(s, W, c) for s in scalars]
    result = INF
    for w in range(W - 1, -1, -1):
        buckets = [INF] * half
        for i, pt in enumerate(points):
            d = digits[i][w] - half
            if d == 0:
                continue
            j = abs(d) - 1
            buckets[j] = add(buckets[j], pt if d > 0 else neg(pt))
        run, tot = INF, INF
        for j in range(half - 1, -1, -1):          # running sum: sum_j (j + 1) * B_j
            run = add(run, buckets[j])
            tot = add(tot, run)
        for _ in range(c):
            result = add(result, result)
        result = add(result, tot)
    return result


# ---------------------------------------------------------------- wire format (README.md:325-331)
def le48(v: int) -> bytes:
    return int(v).to_bytes(48, "little")


def points_to_bytes(pts) -> bytes:
    return b"".join(le48(x) + le48(y) for x, y in pts)


def scalars_to_bytes(ks) -> bytes:
    return b"".join(le48(k) for k in ks)


def xy_from_bytes(b: bytes):
    return (int.from_bytes(b[:48], "little"), int.from_bytes(b[48:96], "little"))


def result_to_bytes(pt) -> bytes:
    """affine result x || y; the point at infinity is written as 96 zero bytes"""
    return bytes(96) if pt is INF else le48(pt[0]) + le48(pt[1])


# ---------------------------------------------------------------- seeded inputs (same scheme as oracle/model.py)
def _splitmix64(state: int):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def _rand_mod_r(state: int):
    v = 0
    for i in range(4):
        state, w = _splitmix64(state)
        v |= w << (64 * i)
    while v >= R_ORDER:
        v -= R_ORDER
    return state, v


def gen_scalars(seed: int, n: int):
    out, s = [], seed
    for _ in range(n):
        s, v = _rand_mod_r(s)
        out.append(v)
    return out


def gen_points(seed: int, n: int):
    """P_i = (a + i*b) * G"""
    s = seed ^ 0xA5A5A5A55A5A5A5A
    s, a = _rand_mod_r(s)
    s, b = _rand_mod_r(s)
    p, qd = scalar_mul(a, G), scalar_mul(b, G)
    out = []
    for _ in range(n):
        out.append(p)
        p = add(p, qd)
    return out
