"""Interval-arithmetic check of the limb rule of csrc/fq377.hpp for the 14-limb instantiation of csrc/curve.hpp (BLS12-377 G1
in its twisted-Edwards form): record conversion, the 8-product addition, the full addition and the four-lane team addition
are replayed on BOUNDS (largest ordinary limb, largest top limb, largest value in units of q) instead of values, iterated
to a fixed point, and every product / subtraction must satisfy its precondition:
  * product: 14 * max(a_i) * max(b_j) + 13 * 2^58 + 2^29 < 2^64  (a 64-bit column never wraps),
  * a - b + K*q: b normalised and its top limb not above the offset's top limb (no borrow out of the number),
  * normalisation: nothing exceeds 32 bits on the way.
Random tests cannot show this (limbs are rarely at their maxima); the GPU has no overflow trap."""
LB, NL = 29, 14
LM = (1 << LB) - 1
Q = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177


class B:
    def __init__(self, lim, top, val):
        self.lim, self.top, self.val = lim, top, val


def N(val):
    return B(LM, int(val * Q) >> (LB * (NL - 1)), val)


def add(a, b):
    return B(a.lim + b.lim, a.top + b.top, a.val + b.val)


def offset(K):
    v = K * Q
    l = [(v >> (LB * i)) & LM for i in range(NL)]
    l[NL - 1] = v >> (LB * (NL - 1))
    for i in range(NL - 1):
        l[i] += 1 << LB
        l[i + 1] -= 1
    return max(l[:NL - 1]), l[NL - 1]


def sub(a, b, K):
    ol, ot = offset(K)
    assert b.lim <= LM, "subtrahend must be normalised"
    assert b.top <= ot, f"subtrahend top limb {b.top} above the offset's {ot} (K = {K})"
    return B(a.lim + ol, a.top + ot, a.val + K)


def mul3(a):
    return B(a.lim * 3, a.top * 3, a.val * 3)


def norm(a):
    assert a.lim < 1 << 32 and a.top + (a.lim >> LB) + 1 < 1 << 32
    return N(a.val)


WORST = [0.0]


def mul(a, b):
    ma, mb = max(a.lim, a.top), max(b.lim, b.top)
    col = NL * ma * mb + (NL - 1) * LM * LM + LM
    WORST[0] = max(WORST[0], col / 2.0 ** 64)
    assert col < 1 << 64, f"column overflow: limbs up to 2^{ma.bit_length()} x 2^{mb.bit_length()}"
    return N(a.val * b.val * Q / 2.0 ** (LB * NL) + 1.0)


def close(E, H, F, G):
    """ete_close<14>: E and G are normalised, (X3, Y3, T3, Z3) = (E F, H G, E H, G F)"""
    En, Gn = norm(E), norm(G)
    return mul(En, F), mul(H, Gn), mul(En, H), mul(Gn, F)          # x, y, t, z


def ete_madd(a, rec):
    x, y, z, t = a
    hm, hp, dt, rz = rec
    A, Bp, Cn, D = mul(sub(y, x, 2), hm), mul(add(y, x), hp), mul(t, dt), mul(z, rz)
    X3, Y3, T3, Z3 = close(sub(Bp, A, 2), add(Bp, A), add(D, Cn), sub(D, Cn, 2))
    return X3, Y3, Z3, T3


def ete_add(a, b):
    x1, y1, z1, t1 = a
    x2, y2, z2, t2 = b
    A = mul(norm(sub(y1, x1, 2)), sub(y2, x2, 2))
    Bp = mul(norm(add(y1, x1)), add(y2, x2))
    C = mul(mul(t1, t2), N(1.0))                                   # times the constant 2d R (class N, < q)
    zz = mul(z1, z2)
    D = add(zz, zz)
    E, H, F, G = norm(sub(Bp, A, 2)), add(Bp, A), norm(sub(D, C, 2)), norm(add(D, C))
    return mul(E, F), mul(H, G), mul(F, G), mul(E, H)              # x, y, z, t  (the team addition forms the same products)


def prep(x, y):
    """pnt_from_sw377 on plain integers: any 384-bit value in class N"""
    c = N(1.0)                                                     # every constant is a residue below q
    sx, yM, fx = mul(x, c), mul(y, c), mul(x, c)
    w, bb, aa = norm(add(sx, c)), add(sx, c), norm(add(fx, c))
    X, Y, Z, T = mul(w, aa), mul(yM, bb), mul(yM, w), mul(aa, bb)
    return norm(sub(Y, X, 2)), norm(add(Y, X)), mul(T, c), norm(add(Z, Z))      # hm, hp, dt, z


def _join(x, y):
    return tuple(B(max(p.lim, q.lim), max(p.top, q.top), max(p.val, q.val)) for p, q in zip(x, y))


def test_limb_rule_holds_at_the_fixed_point_of_the_bounds():
    ol4, ot4 = offset(4)
    raw = B(LM, (1 << 384) >> (LB * (NL - 1)), 2.0 ** 384 / Q)             # a non-canonical 384-bit coordinate
    rec = prep(raw, raw)
    assert all(c.val < 4.1 for c in rec)
    rec_neg = (rec[1], rec[0], B(ol4, ot4, 4.0), rec[3])                   # negated record: hm <-> hp, dt -> 4q - dt
    acc = (N(1.0), N(1.0), N(1.0), N(1.0))
    for _ in range(40):                                                    # accumulate: bounds reach a fixed point
        for r in (rec, rec_neg):
            acc = _join(acc, ete_madd(acc, r))
    s = acc
    for _ in range(40):                                                    # reductions: sums of such accumulators
        s = _join(s, ete_add(s, s))
        s = _join(s, ete_add(s, acc))
    assert all(c.lim <= LM for c in s) and all(c.val < 2.1 for c in s)
    assert 0.5 < WORST[0] < 1.0                                            # the rule is tight, not vacuous
