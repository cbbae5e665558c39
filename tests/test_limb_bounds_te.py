"""Interval-arithmetic check of the limb rule of csrc/fp.hpp (9 limbs x 29 bits) for ete_madd / ete_add of
csrc/curve.hpp, as tests/test_limb_bounds_bls377.py does for the other curve: formulas replayed on bounds, iterated to
a fixed point; every 64-bit column must stay below 2^64 and every offset subtraction must have a normalised subtrahend
whose top limb the offset covers."""
LB, NL = 29, 9
LM = (1 << LB) - 1
P = 8444461749428370424248824938781546531375899335154063827935233455917409239041


class B:
    def __init__(self, lim, top, val):
        self.lim, self.top, self.val = lim, top, val


def N(val):
    return B(LM, int(val * P) >> (LB * (NL - 1)), val)


def add(a, b):
    return B(a.lim + b.lim, a.top + b.top, a.val + b.val)


def offset(K):
    v = K * P
    l = [(v >> (LB * i)) & LM for i in range(NL)]
    l[NL - 1] = v >> (LB * (NL - 1))
    for i in range(NL - 1):
        l[i] += 1 << LB
        l[i + 1] -= 1
    return max(l[:NL - 1]), l[NL - 1]


def sub(a, b, K):
    ol, ot = offset(K)
    assert b.lim <= LM and b.top <= ot, "subtrahend must be normalised and below the offset"
    return B(a.lim + ol, a.top + ot, a.val + K)


def neg(b, K):
    ol, ot = offset(K)
    assert b.lim <= LM and b.top <= ot
    return B(ol, ot, K)


def norm(a):
    assert a.lim < 1 << 32 and a.top + (a.lim >> LB) + 1 < 1 << 32
    return N(a.val)


WORST = [0.0]


def mul(a, b):
    ma, mb = max(a.lim, a.top), max(b.lim, b.top)
    col = NL * ma * mb + (NL - 1) * LM * LM + LM
    WORST[0] = max(WORST[0], col / 2.0 ** 64)
    assert col < 1 << 64, f"column overflow: limbs up to 2^{ma.bit_length()} x 2^{mb.bit_length()}"
    return N(a.val * b.val * P / 2.0 ** (LB * NL) + 1.0)


def ete_madd(a, hm, hp, dt):
    x, y, z, t = a
    A, Bp, Cn = mul(sub(y, x, 2), hm), mul(add(y, x), hp), mul(t, dt)
    E, H, F, G = sub(Bp, A, 2), add(Bp, A), add(z, Cn), sub(z, Cn, 2)
    return mul(E, F), mul(H, G), mul(G, F), mul(E, H)          # x, y, z, t


def ete_add(a, b):
    x1, y1, z1, t1 = a
    x2, y2, z2, t2 = b
    A = mul(norm(sub(y1, x1, 2)), sub(y2, x2, 2))
    Bp = mul(add(y1, x1), add(y2, x2))
    tt, zz = mul(t1, t2), mul(z1, z2)
    C = N(1.0002)                                      # fp_mul_k2d(tt): class N, below 1.0001 p (test_small_constant_product)
    D = add(zz, zz)
    E, H = sub(Bp, A, 2), add(Bp, A)                  # N = 9: E and G stay wide (curve.hpp, fe_norm_if_needed)
    F, G = norm(sub(D, C, 2)), add(D, C)
    return mul(E, F), mul(H, G), mul(F, G), mul(E, H)


def test_limb_rule_holds_for_every_product_of_the_point_formulas():
    rec = N(1.1)
    rec_dt_neg = neg(N(1.1), 4)                     # pnt_cneg: 4p - dt
    acc = (N(1.0),) * 4
    for _ in range(20):
        for dt in (rec, rec_dt_neg):
            out = ete_madd(acc, rec, rec, dt)
            acc = tuple(B(max(p.lim, q.lim), max(p.top, q.top), max(p.val, q.val)) for p, q in zip(acc, out))
    assert all(c.lim == LM and c.val < 1.1 for c in acc)           # accumulators are product outputs: class N, < 1.1p
    s = ete_add(acc, acc)
    assert all(c.lim == LM and c.val < 1.1 for c in s)
    assert 0.5 < WORST[0] < 1.0
