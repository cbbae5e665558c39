"""Interval-arithmetic check of the limb rule of csrc/fq377.hpp for the two group-law routines of csrc/curve377.hpp:
the formulas are replayed on BOUNDS (largest ordinary limb, largest top limb, largest value in units of q) instead of
values, iterated to a fixed point, and every product / subtraction must satisfy its precondition:
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


def finish(t0x3, t1n, t3n, t4, y3n, z3n):
    p0, p1, p2 = mul(y3n, t4), mul(t3n, t1n), mul(y3n, t0x3)
    s0, s1, s2 = mul(t1n, z3n), mul(t3n, t0x3), mul(z3n, t4)
    return sub(p1, p0, 2), add(s0, p2), add(s2, s1)


def g1_madd(a, bx, by):
    X1, Y1, Z1 = (norm(c) for c in a)
    bsum = norm(add(bx, by))
    t0, t1, t3 = mul(X1, bx), mul(Y1, by), mul(bsum, add(X1, Y1))
    s0, s1 = mul(Z1, by), mul(Z1, bx)
    t3n = norm(sub(sub(t3, t0, 2), t1, 2))
    t4 = add(s0, Y1)
    y3n = norm(mul3(add(s1, X1)))
    t0x3 = mul3(t0)
    z3n = norm(add(t1, mul3(Z1)))
    t1n = norm(sub(sub(sub(t1, Z1, 4), Z1, 4), Z1, 4))
    return finish(t0x3, t1n, t3n, t4, y3n, z3n)


def g1_add(a, b):
    X1, Y1, Z1 = (norm(c) for c in a)
    X2, Y2, Z2 = (norm(c) for c in b)
    t0, t1, t2 = mul(X1, X2), mul(Y1, Y2), mul(Z1, Z2)
    u0 = mul(norm(add(X1, Y1)), add(X2, Y2))
    u1 = mul(norm(add(Y1, Z1)), add(Y2, Z2))
    u2 = mul(norm(add(X1, Z1)), add(X2, Z2))
    t3n = norm(sub(sub(u0, t0, 2), t1, 2))
    t4n = norm(sub(sub(u1, t1, 2), t2, 2))
    y3n = norm(mul3(norm(sub(sub(u2, t0, 2), t2, 2))))
    t0x3 = mul3(t0)
    z3n = norm(add(t1, mul3(t2)))
    t1n = norm(sub(sub(sub(t1, t2, 2), t2, 2), t2, 2))
    return finish(t0x3, t1n, t3n, t4n, y3n, z3n)


def _join(x, y):
    return tuple(B(max(p.lim, q.lim), max(p.top, q.top), max(p.val, q.val)) for p, q in zip(x, y))


def test_limb_rule_holds_at_the_fixed_point_of_the_bounds():
    ol2, ot2 = offset(2)
    rec_x, rec_y, rec_y_neg = N(1.01), N(1.01), B(ol2, ot2, 2.0)          # record: products; negated y = 2q - y
    acc = (N(1.0), N(1.0), N(1.0))
    for _ in range(40):                                                    # accumulate: bounds reach a fixed point
        for by in (rec_y, rec_y_neg):
            acc = _join(acc, g1_madd(acc, rec_x, by))
    s = acc
    for _ in range(40):                                                    # reductions: sums of such accumulators
        s = _join(s, g1_add(s, s))
        s = _join(s, g1_add(s, acc))
    assert all(c.lim < 1 << 31 for c in s) and all(c.val < 4.1 for c in s)
    assert 0.5 < WORST[0] < 1.0                                            # the rule is tight, not vacuous
