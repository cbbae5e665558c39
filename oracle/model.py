"""Pure-Python bigint model of the reference's curve and MSM pipeline (small cases only).

TEST INFRASTRUCTURE ONLY -- see oracle/te_oracle.c's header.  This is the slow, obviously-right twin of
the C oracle; tests pin both against the reference's known-answer vectors and against each other.

Reference citations (paths relative to /root/reference/src):
  constants      reference/utils/FieldMath.ts:7-10,104-137, reference/params/AleoConstants.ts:2-4
  decompress     reference/utils/FieldMath.ts:31-55 (getPointFromX)
  signed digits  submission/miscellaneous/utils.ts:52-95
  pipeline       submission/miscellaneous/tests/cuzk.test.ts:28-141
  13-bit params  submission/implementation/cuzk/utils.ts:518-586 (compute_misc_params), :440-496 (to_words_le)
"""
from __future__ import annotations

P = 8444461749428370424248824938781546531375899335154063827935233455917409239041
A = P - 1                      # EDWARDS_A
D = 3021                       # EDWARDS_D
L = 2111115437357092606062206234695386632838870926408408195193685246394721360383  # prime subgroup order
COFACTOR = 4
GX = 1540945439182663264862696551825005342995406165131907382295858612069623286213
GY = 8003546896475222703853313610036801932325312921786952001586936882361378122196
# harness fixed point, ui/AllBenchmarks.tsx:107-110 and miscellaneous/tests/cuzk.test.ts:15-25
HX = 2796670805570508460920584878396618987767121022598342527208237783066948667246
HY = 8134280397689638111748378379571739274369602049665521098046934931245960532166

MASK64 = (1 << 64) - 1
ZERO = (0, 1)


def inv(a: int) -> int:
    return pow(a % P, P - 2, P)


def on_curve(pt) -> bool:
    x, y = pt
    return (-x * x + y * y - 1 - D * x * x * y * y) % P == 0


def add(p1, p2):
    """Complete affine twisted-Edwards addition, a = -1."""
    x1, y1 = p1
    x2, y2 = p2
    k = D * x1 * x2 * y1 * y2 % P
    x3 = (x1 * y2 + y1 * x2) * inv(1 + k) % P
    y3 = (y1 * y2 + x1 * x2) * inv(1 - k) % P
    return (x3, y3)


def neg(p1):
    return ((-p1[0]) % P, p1[1])


# extended coordinates for speed in scalar_mul (same group law, add-2008-hwcd, a = -1)
def _ext_add(p1, p2):
    X1, Y1, T1, Z1 = p1
    X2, Y2, T2, Z2 = p2
    Aa = X1 * X2 % P
    B = Y1 * Y2 % P
    C = D * T1 * T2 % P
    Dd = Z1 * Z2 % P
    E = ((X1 + Y1) * (X2 + Y2) - Aa - B) % P
    F = (Dd - C) % P
    G = (Dd + C) % P
    H = (B + Aa) % P
    return (E * F % P, G * H % P, E * H % P, F * G % P)


def scalar_mul(k: int, pt):
    """k*pt over the integer k (no reduction of k), as noble's multiplyUnsafe (FieldMath.ts:73-88)."""
    acc = (0, 1, 0, 1)
    base = (pt[0], pt[1], pt[0] * pt[1] % P, 1)
    while k:
        if k & 1:
            acc = _ext_add(acc, base)
        base = _ext_add(base, base)
        k >>= 1
    zi = inv(acc[3])
    return (acc[0] * zi % P, acc[1] * zi % P)


def sqrt_mod_p(a: int):
    """Tonelli-Shanks (p - 1 = 2^47 * q). Returns a root or None."""
    a %= P
    if a == 0:
        return 0
    if pow(a, (P - 1) // 2, P) != 1:
        return None
    q, s = P - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (P - 1) // 2, P) != P - 1:
        z += 1
    m, c, t, r = s, pow(z, q, P), pow(a, q, P), pow(a, (q + 1) // 2, P)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % P
            i += 1
        b = pow(c, 1 << (m - i - 1), P)
        m, c = i, b * b % P
        t, r = t * c % P, r * b % P
    return r


def point_from_x(x: int):
    """getPointFromX, FieldMath.ts:31-55: y^2 = (a x^2 - 1)/(d x^2 - 1); pick the root in the prime subgroup."""
    x2 = x * x % P
    y2 = (A * x2 - 1) * inv(D * x2 - 1) % P
    y = sqrt_mod_p(y2)
    if y is None:
        raise ValueError("x is not on the curve")
    for cand in (y, (-y) % P):
        if scalar_mul(L, (x, cand)) == ZERO:
            return (x, cand)
    # FieldMath returns the negated root when the first is not annihilated by l
    return (x, (-y) % P)


# ------------------------------------------------------------------ wire format
def le32(v: int) -> bytes:
    return int(v).to_bytes(32, "little")


def points_to_bytes(pts) -> bytes:
    return b"".join(le32(x) + le32(y) for x, y in pts)


def scalars_to_bytes(ks) -> bytes:
    return b"".join(le32(k) for k in ks)


def xy_from_bytes(b: bytes):
    return (int.from_bytes(b[:32], "little"), int.from_bytes(b[32:64], "little"))


# ------------------------------------------------------------------ deterministic synthetic inputs
def _splitmix64(state: int):
    state = (state + 0x9E3779B97F4A7C15) & MASK64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return state, z ^ (z >> 31)


def _rand_mod_p(state: int):
    v = 0
    for i in range(4):
        state, w = _splitmix64(state)
        v |= w << (64 * i)
    return state, v % P


def gen_scalars(seed: int, n: int):
    """Mirror of ora_gen_scalars (te_oracle.c): 256 random bits reduced mod p."""
    out, s = [], seed & MASK64
    for _ in range(n):
        s, v = _rand_mod_p(s)
        out.append(v)
    return out


def gen_points(seed: int, n: int):
    """Mirror of ora_gen_points: P_i = (a + i*b)*G, chain P_{i+1} = P_i + b*G."""
    s = (seed ^ 0xA5A5A5A55A5A5A5A) & MASK64
    s, a = _rand_mod_p(s)
    s, b = _rand_mod_p(s)
    g = (GX, GY)
    p0, q = scalar_mul(a, g), scalar_mul(b, g)
    out = []
    cur = p0
    for _ in range(n):
        out.append(cur)
        cur = add(cur, q)
    return out


# ------------------------------------------------------------------ the reference pipeline, tiny sizes
def gen_points_random(seed: int, n: int):
    """Mirror of ora_gen_points_random: P_i = a_i * G, a_i from the splitmix64 stream of seed ^ 0x5A5A5A5AA5A5A5A5."""
    s = (seed ^ 0x5A5A5A5AA5A5A5A5) & MASK64
    out = []
    for _ in range(n):
        s, a = _rand_mod_p(s)
        out.append(scalar_mul(a, (GX, GY)))
    return out


def to_words_le(val: int, num_words: int, word_size: int):
    """utils.ts:440-465"""
    mask = (1 << word_size) - 1
    return [(val >> (word_size * i)) & mask for i in range(num_words)]


def decompose_scalar_signed(s: int, num_words: int, c: int):
    """miscellaneous/utils.ts:52-95 for one scalar; returns digit + 2^(c-1) per window."""
    l, shift = 1 << c, 1 << (c - 1)
    limbs = to_words_le(s, num_words, c)
    out, carry = [], 0
    for i in range(num_words):
        v = limbs[i] + carry
        if v >= l // 2:
            v = -(l - v)
            carry = 1
        else:
            carry = 0
        out.append(v + shift)
    if carry:
        raise ValueError("final carry is 1")
    return out


def msm_pipeline(points, scalars, c: int):
    """cuzk.test.ts:28-141 with affine arithmetic: signed digits -> per-window buckets ->
    running-sum reduction -> Horner."""
    num_words = -(-256 // c)
    h = 1 << (c - 1)
    digits = [decompose_scalar_signed(k, num_words, c) for k in scalars]
    wsum = []
    for w in range(num_words):
        buckets = [ZERO] * h
        for i, pt in enumerate(points):
            dgt = digits[i][w] - h
            if dgt == 0:
                continue
            if dgt > 0:
                buckets[dgt] = add(buckets[dgt], pt)
            else:
                slot = (-dgt) % h          # digit -h lands in slot 0, which carries weight h
                buckets[slot] = add(buckets[slot], neg(pt))
        # running sum, bpr.ts:4-24: g = h*B[0] + sum_{t>=1} t*B[t]
        m = buckets[0]
        g = m
        for i in range(h - 1):
            m = add(m, buckets[h - 1 - i])
            g = add(g, m)
        wsum.append(g)
    res = wsum[-1]
    for w in range(num_words - 2, -1, -1):
        res = scalar_mul(1 << c, res)
        res = add(res, wsum[w])
    return res


def msm_naive(points, scalars):
    acc = ZERO
    for pt, k in zip(points, scalars):
        if k:
            acc = add(acc, scalar_mul(k, pt))
    return acc


def compute_misc_params(p: int, word_size: int):
    """utils.ts:518-586 (the values pinned by miscellaneous/tests/utils.test.ts:146-183)."""
    p_width = p.bit_length()
    num_words = -(-p_width // word_size)
    max_terms = num_words * 2
    k = 1
    while k * 2 ** (2 * word_size) <= 2 ** 32:
        k += 1
    nsafe = k // 2
    r = 1 << (num_words * word_size)
    rinv = pow(r, -1, p)
    n0 = (-pow(p, -1, r)) % r % (1 << word_size)
    return {"num_words": num_words, "max_terms": max_terms, "k": k, "nsafe": nsafe,
            "r": r % p, "rinv": rinv, "n0": n0, "edwards_d": D * r % p}


# ------------------------------------------------------------------ harness codecs (reference/webgpu/utils.ts)
def bigint_to_u32_array(v: int):
    """bigIntToU32Array, reference/webgpu/utils.ts:47-62: 8 words, most significant first."""
    return [(v >> (32 * (7 - i))) & 0xFFFFFFFF for i in range(8)]


def u32_array_to_bigints(words):
    """u32ArrayToBigInts, reference/webgpu/utils.ts:64-79"""
    out = []
    for i in range(0, len(words), 8):
        v = 0
        for j, w in enumerate(words[i:i + 8]):
            v |= int(w) << (32 * (7 - j))
        out.append(v)
    return out


def bigints_to_buffer_le(vals, bits: int = 256) -> bytes:
    """bigIntsToBufferLE, reference/webgpu/utils.ts:90-99 -- the wire format of compute_msm's Buffers."""
    return b"".join(int(v).to_bytes(bits // 8, "little") for v in vals)


def read_bigints_from_buffer_le(buf: bytes, bits: int = 256):
    """readBigIntsFromBufferLE, reference/webgpu/utils.ts:101-112"""
    step = bits // 8
    return [int.from_bytes(buf[i:i + step], "little") for i in range(0, len(buf), step)]
