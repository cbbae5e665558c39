// field.hpp -- one vocabulary for the two base fields of the engine, so that the curve (curve.hpp) and the kernels
// (kernels.hip.hpp) are written once for any limb count N:
//   N = 9   fp  (fp.hpp):     base field of the Twisted-Edwards BLS12 curve, 253 bits, R = 2^261
//   N = 14  fq  (fq377.hpp):  base field of BLS12-377 (G1 is handled in its twisted-Edwards form), 377 bits, R = 2^406
// Everything here is an overload on fel<N> that forwards to the field's own function; the arithmetic lives in the two
// field headers.  LIMB RULES differ: with 9 limbs a 64-bit column takes a "difference x sum" product (limbs 2^30.6 x 2^30),
// with 14 limbs one operand of every product must be normalised -- fe_wide_ok<N>() tells the curve code which holds.
#pragma once
#include "fp.hpp"
#include "fq377.hpp"

namespace te {

TE_HD fel<9> fe_add(const fel<9>& a, const fel<9>& b) { return fp_add(a, b); }
TE_HD fel<14> fe_add(const fel<14>& a, const fel<14>& b) { return te377::fq_add(a, b); }
template <int K> TE_HD fel<9> fe_sub(const fel<9>& a, const fel<9>& b) { return fp_sub<K>(a, b); }
template <int K> TE_HD fel<14> fe_sub(const fel<14>& a, const fel<14>& b) { return te377::fq_sub<K>(a, b); }
template <int K> TE_HD fel<9> fe_neg(const fel<9>& a) { return fp_neg<K>(a); }
template <int K> TE_HD fel<14> fe_neg(const fel<14>& a) { return te377::fq_neg<K>(a); }
TE_HD fel<9> fe_norm(const fel<9>& a) { return fp_norm(a); }
TE_HD fel<14> fe_norm(const fel<14>& a) { return te377::fq_norm(a); }
// one latency-bound product (fp: the variant with the shorter dependency chain)
TE_HD fel<9> fe_mul(const fel<9>& a, const fel<9>& b) { return mont_mul(a, b); }
TE_HD fel<14> fe_mul(const fel<14>& a, const fel<14>& b) { return te377::mont_mul(a, b); }
// M independent products in lockstep (throughput-bound code)
template <int M> TE_HD void fe_mul_x(const fel<9> (&a)[M], const fel<9> (&b)[M], fel<9> (&r)[M]) { mont_mul_x<M>(a, b, r); }
template <int M> TE_HD void fe_mul_x(const fel<14> (&a)[M], const fel<14> (&b)[M], fel<14> (&r)[M]) { te377::mont_mul_x<M>(a, b, r); }

template <int N> TE_HD fel<N> fe_zero() { fel<N> r; for (int i = 0; i < N; i++) r.v[i] = 0; return r; }
template <int N> TE_HD fel<N> fe_one();        // R mod p: 1 in Montgomery form
template <int N> TE_HD fel<N> fe_k2d();        // 2 d in Montgomery form (the curve constant of add-2008-hwcd-3)
template <> TE_HD fel<9> fe_one<9>() { return fp_R1(); }
template <> TE_HD fel<14> fe_one<14>() { return te377::fq_R1(); }
template <> TE_HD fel<9> fe_k2d<9>() { return fp_K2D_MONT(); }
template <> TE_HD fel<14> fe_k2d<14>() { return te377::fq_K2D_MONT(); }

// a * 2d: a full product with the constant in Montgomery form -- except on the Twisted-Edwards BLS12 curve, whose 2d = 6042 is 13
// bits wide (fp_mul_k2d: two multiply-accumulates per limb).  Result: class N, value below 1.1 p either way.
TE_HD fel<9> fe_mul_k2d(const fel<9>& a) { return fp_mul_k2d(a); }
TE_HD fel<14> fe_mul_k2d(const fel<14>& a) { return fe_mul(a, fe_k2d<14>()); }

// may a product take a difference in offset form (limbs < 2^30.6) times a sum (limbs < 2^30)?  9 * 2^60.6 + 8 * 2^58 < 2^64
// holds, 14 * 2^60.6 does not (fq377.hpp: one operand normalised, the other below 2^30.8)
template <int N> constexpr bool fe_wide_ok() { return N <= 9; }
// normalise only where the limb rule of the field requires it
template <int N> TE_HD fel<N> fe_norm_if_needed(const fel<N>& a) { if constexpr (fe_wide_ok<N>()) return a; else return fe_norm(a); }

}  // namespace te
