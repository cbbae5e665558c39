// curve.hpp -- point arithmetic on a twisted Edwards curve -x^2 + y^2 = 1 + d x^2 y^2 (a = -1) over fel<N>:
//   N = 9   the Twisted-Edwards BLS12 curve of the competition, d = 3021 (reference/params/AleoConstants.ts:2-4,
//           reference/utils/FieldMath.ts:104-137)
//   N = 14  BLS12-377 G1 in its twisted-Edwards form (BASELINE config 5; the reference has no code for this curve,
//           README.md:57-73,279-287).  d = -(A - 2) / (A + 2) for the Montgomery form of y^2 = x^3 + 1 (tools/gen_constants.py);
//           d is a square there, so the addition law is complete on the subgroup of odd prime order r only -- where MSM
//           inputs live -- not on the cofactor part.
//
// Replaces add_points / double_point / negate_point of the reference (wgsl/curve/ec.template.wgsl:7-66,
// wgsl/cuzk/smvp.template.wgsl:47-56).  The reference adds two full extended points with add-2008-hwcd (10 field products
// + T = x*y recomputed per gathered point, smvp.template.wgsl:107-108).  Here the n input points are converted ONCE into a
// RECORD and bucket accumulation is a 7-product (N = 9: affine record) or 8-product (N = 14: projective record, no
// inversion in the conversion) addition with no reduction modulo p at all (limb-magnitude rules in fp.hpp / fq377.hpp).
// Any correct group law gives the same affine result, which is the only thing compared with the reference.
#pragma once
#include "field.hpp"

namespace te {

// Extended twisted Edwards accumulator (X : Y : Z : T), x = X/Z, y = Y/Z, T = XY/Z.
// Every coordinate is a product output: limb class N, value < 1.1p.  4 N words, x | y | z | t.
template <int N> struct ete_t { fel<N> x, y, z, t; };
using ete = ete_t<9>;            // 144 bytes

// Record of an input point, up to a common factor of its coordinates (projectively the result of an addition does not
// depend on it):  hm ~ (Y - X)/2,  hp ~ (Y + X)/2,  dt ~ -d T,  z ~ Z  of the extended point (X : Y : Z : T).
//   N = 9 : affine, z = 1 is not stored: ((y-x)/2, (y+x)/2, -d*x*y), 27 words in a 128-byte slot.
//   N = 14: projective (converted from short Weierstrass without a division, see pnt_from_sw377): 56 words = 224 bytes.
// Products of class N, values < 1.1p; a negated record holds 4p - dt with limbs < 2^30.6.
// dt carries the MINUS sign so that F = D' + C' below is a sum and no product meets two differences.
template <int N> struct pnt_t;
template <> struct pnt_t<9> { fel<9> hm, hp, dt; };
template <> struct pnt_t<14> { fel<14> hm, hp, dt, z; };
using pnt = pnt_t<9>;
// BLS12-377 G1, AFFINE record of a BOUND point set (te_msm_bind_points: the conversion is paid once, so its one inversion per
// point -- Montgomery's trick, k_affine377 -- costs nothing per MSM): the projective record divided by its z,
// ((y-x)/2, (y+x)/2, -d x y) of the Edwards point like pnt_t<9>, z = 1 not stored.  42 words = 168 bytes instead of 224, and
// an addition of 7 products instead of 8 (ete_madd below).  Products of class N, values < 1.1q.
struct pnt_aff377 { fel<14> hm, hp, dt; };

template <int N> TE_HD ete_t<N> ete_identity_t() { ete_t<N> r; r.x = fe_zero<N>(); r.y = fe_one<N>(); r.z = fe_one<N>(); r.t = fe_zero<N>(); return r; }
TE_HD ete ete_identity() { return ete_identity_t<9>(); }

// -(x, y) = (-x, y): swaps hm and hp, negates dt.  The selection is bitwise under a per-lane mask: gfx950 does
// (b & m) | (a & ~m) in one v_bitop3_b32 at full rate (2.7 cycles per wave, tools/ubench); v_cndmask_b32 and v_bfi_b32
// cost 4.3.
TE_HD uint32_t mask_select(uint32_t m, uint32_t b, uint32_t a) {          // m ? b : a, bit by bit
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(m, b, a, 0xCA);
#else
  return (b & m) | (a & ~m);
#endif
}
template <int N> TE_HD pnt_t<N> pnt_cneg(const pnt_t<N>& a, bool neg) {
  pnt_t<N> r = a; const fel<N> ndt = fe_neg<4>(a.dt);
  const uint32_t m = neg ? 0xffffffffu : 0u;
#pragma unroll
  for (int i = 0; i < N; i++) {
    r.hm.v[i] = mask_select(m, a.hp.v[i], a.hm.v[i]);
    r.hp.v[i] = mask_select(m, a.hm.v[i], a.hp.v[i]);
    r.dt.v[i] = mask_select(m, ndt.v[i], a.dt.v[i]);
  }
  return r;
}

TE_HD pnt_aff377 pnt_cneg(const pnt_aff377& a, bool neg) {
  pnt_aff377 r = a; const fel<14> ndt = fe_neg<4>(a.dt);
  const uint32_t m = neg ? 0xffffffffu : 0u;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    r.hm.v[i] = mask_select(m, a.hp.v[i], a.hm.v[i]);
    r.hp.v[i] = mask_select(m, a.hm.v[i], a.hp.v[i]);
    r.dt.v[i] = mask_select(m, ndt.v[i], a.dt.v[i]);
  }
  return r;
}

// Affine (x, y) as plain integers (class N, ANY 256-bit value) -> record, in four products and nothing else:
//   hm = (y - x + 16p) * (R^2/2) / R,   hp = (y + x) * (R^2/2) / R,   dt = (x * y / R) * (-d R^3) / R.
// The constants fold the conversion to Montgomery form, the halving and the factor d into the products
// (16p > 2^256 keeps y - x non-negative for non-canonical inputs).  Values: hm, hp < 1.07p, dt < 1.01p.
TE_HD pnt pnt_from_affine_raw(const fp& x, const fp& y) {
  const fp a[3] = {fp_sub<16>(y, x), fp_add(y, x), x}, b[3] = {fp_R2_HALF(), fp_R2_HALF(), y};
  fp o[3];
  mont_mul_x<3>(a, b, o);
  pnt r;
  r.hm = o[0]; r.hp = o[1];
  r.dt = mont_mul(o[2], fp_NEG_D_R3());
  return r;
}

// BLS12-377 G1: short-Weierstrass affine (x, y) as plain integers (class N, any 384-bit value) -> projective twisted-Edwards
// record, 8 products and no division.  With s = 1/sqrt(3), f = sqrt(-(A+2)/B) (tools/gen_constants.py):
//   Montgomery  u = s (x + 1), v = s y;   Edwards  X = f u / v = f (x + 1) / y,   Y = (u - 1)/(u + 1) = (s x + s - 1)/(s x + s + 1).
//   With a = f (x + 1), b = s x + s - 1, w = s x + s + 1:  X = a / y, Y = b / w, and the extended point is simply
//   (X : Y : Z : T) = (a w : b y : y w : a b)   (X Y / Z = a b).
//   record, times 2:  hm = Y - X,  hp = Y + X,  dt = -2 d T,  z = 2 Z.
// Undefined (all-zero record, the neutral element's weight is lost) for y = 0 or w = 0: points of order 2 and 4, never in G1.
TE_HD pnt_t<14> pnt_from_sw377(const fel<14>& x, const fel<14>& y) {
  using namespace te377;
  const fq a1[3] = {x, y, x}, b1[3] = {fq_S_R2(), fq_R2(), fq_F_R2()};
  fq o1[3];
  fe_mul_x<3>(a1, b1, o1);                                   // s x, y, f x   (Montgomery form, class N)
  const fq w = fe_norm(fe_add(o1[0], fq_SP1_MONT())), bb = fe_add(o1[0], fq_SM1_MONT()), aa = fe_norm(fe_add(o1[2], fq_F_MONT()));
  const fq a2[4] = {w, o1[1], o1[1], aa}, b2[4] = {aa, bb, w, bb};
  fq o2[4];
  fe_mul_x<4>(a2, b2, o2);                                   // X = a w,  Y = b y,  Z = y w,  T = a b
  pnt_t<14> r;
  r.hm = fe_norm(fe_sub<2>(o2[1], o2[0]));
  r.hp = fe_norm(fe_add(o2[1], o2[0]));
  r.z = fe_norm(fe_add(o2[2], o2[2]));
  r.dt = fe_mul(o2[3], fq_NEG_2D_MONT());
  return r;
}

// the four closing products of every addition: (X3, Y3, T3, Z3) = (E F, H G, E H, G F) with E, G differences (offset form)
// and H, F sums.  N = 9: every product is "difference x sum", 9 * 2^30.6 * 2^30 + 8 * 2^58 = 0.9 * 2^64: nothing is
// normalised.  N = 14: one operand of every product must be normalised -- E and G are.
template <int N> TE_HD ete_t<N> ete_close(const fel<N>& E, const fel<N>& H, const fel<N>& F, const fel<N>& G) {
  const fel<N> En = fe_norm_if_needed(E), Gn = fe_norm_if_needed(G);
  const fel<N> l[4] = {En, H, En, Gn}, rr[4] = {F, Gn, H, F};
  fel<N> o[4];
  fe_mul_x<4>(l, rr, o);
  ete_t<N> r;
  r.x = o[0]; r.y = o[1]; r.t = o[2]; r.z = o[3];
  return r;
}

// Mixed addition acc + b with an affine record, 7 products, unified and complete (a = -1 is a square, d is not).
// With A' = (Y1-X1)(y2-x2)/2, B' = (Y1+X1)(y2+x2)/2:  E = B'-A' = X1 y2 + Y1 x2,
// H = B'+A' = Y1 y2 + X1 x2, C = d T1 x2 y2, F = Z1 - C, G = Z1 + C;  (X3,Y3,T3,Z3) = (EF, HG, EH, GF).
// The record holds -d x2 y2, so the product below is C' = -C and F = Z1 + C' is a SUM, G = Z1 - C' the difference.
// Limb classes: A' = D x N, B' = S x N, C' = N x (<2^30.6);  E and G are D (differences in offset form), H and F are S.
// Values: everything is below 5p before a product and below 1.1p after it; nothing is reduced mod p.
TE_HD ete ete_madd(const ete& a, const pnt& b) {
  const fp in1[3] = {fp_sub<2>(a.y, a.x), fp_add(a.y, a.x), a.t}, in2[3] = {b.hm, b.hp, b.dt};
  fp abc[3];
  mont_mul_x<3>(in1, in2, abc);
  const fp &A = abc[0], &B = abc[1], &Cn = abc[2];
  return ete_close<9>(fp_sub<2>(B, A), fp_add(B, A), fp_add(a.z, Cn), fp_sub<2>(a.z, Cn));
}
// The same with a projective record: D' = Z1 z2 takes the place of Z1 -- 8 products.
TE_HD ete_t<14> ete_madd(const ete_t<14>& a, const pnt_t<14>& b) {
  const fel<14> in1[4] = {fe_sub<2>(a.y, a.x), fe_add(a.y, a.x), a.t, a.z}, in2[4] = {b.hm, b.hp, b.dt, b.z};
  fel<14> p[4];
  fe_mul_x<4>(in1, in2, p);
  const fel<14> &A = p[0], &B = p[1], &Cn = p[2], &D = p[3];
  return ete_close<14>(fe_sub<2>(B, A), fe_add(B, A), fe_add(D, Cn), fe_sub<2>(D, Cn));
}

// BLS12-377 with an AFFINE record (bound bases): the 7-product form of the other curve under the 14-limb rule -- one operand of
// every product normalised: the record's coordinates are (a negated dt has limbs < 2^30.6 and meets T1, class N), E and G are
// normalised by ete_close; F = Z1 + C' is a sum of two product outputs (limbs < 2^30), H likewise.
TE_HD ete_t<14> ete_madd(const ete_t<14>& a, const pnt_aff377& b) {
  const fel<14> in1[3] = {fe_sub<2>(a.y, a.x), fe_add(a.y, a.x), a.t}, in2[3] = {b.hm, b.hp, b.dt};
  fel<14> p[3];
  fe_mul_x<3>(in1, in2, p);
  const fel<14> &A = p[0], &B = p[1], &Cn = p[2];
  return ete_close<14>(fe_sub<2>(B, A), fe_add(B, A), fe_add(a.z, Cn), fe_sub<2>(a.z, Cn));
}

// neutral element + b without the products whose result is known: with (X1 : Y1 : Z1 : T1) = (0 : 1 : 1 : 0) the mixed addition
// has A' = hm, B' = hp, C' = 0, D' = z2, i.e. (X3, Y3, T3, Z3) = (E z2, H z2, E H, z2^2) with E = hp - hm, H = hp + hm: three
// products for an affine record (z2 = 1: E and H only pass through a product by one to become class N below 2p), four for a
// projective one, instead of 7 / 8.  Every segment of k_accumulate starts with this step (one in ~31 additions).
TE_HD ete ete_from_pnt(const pnt& b) {
  const fp one = fp_R1();
  const fp E = fp_sub<2>(b.hp, b.hm), H = fp_add(b.hp, b.hm);
  const fp l[3] = {E, H, E}, rr[3] = {one, one, H};
  fp o[3];
  mont_mul_x<3>(l, rr, o);
  ete r;
  r.x = o[0]; r.y = o[1]; r.t = o[2]; r.z = one;
  return r;
}
TE_HD ete_t<14> ete_from_pnt(const pnt_t<14>& b) {
  const fel<14> En = fe_norm(fe_sub<2>(b.hp, b.hm)), H = fe_add(b.hp, b.hm);
  const fel<14> l[4] = {En, H, En, b.z}, rr[4] = {b.z, b.z, H, b.z};
  fel<14> o[4];
  fe_mul_x<4>(l, rr, o);
  ete_t<14> r;
  r.x = o[0]; r.y = o[1]; r.t = o[2]; r.z = o[3];
  return r;
}

TE_HD ete_t<14> ete_from_pnt(const pnt_aff377& b) {          // z2 = 1: (E, H, E H, 1), three products
  const fel<14> one = fe_one<14>();
  const fel<14> En = fe_norm(fe_sub<2>(b.hp, b.hm)), H = fe_add(b.hp, b.hm);
  const fel<14> l[3] = {En, H, En}, rr[3] = {one, one, H};
  fel<14> o[3];
  fe_mul_x<3>(l, rr, o);
  ete_t<14> r;
  r.x = o[0]; r.y = o[1]; r.t = o[2]; r.z = one;
  return r;
}

// Full addition a + b of two accumulators (add-2008-hwcd-3 shape, k = 2d), 9 products.
// One operand of the opening product of two differences and F are normalised, so that no product sees two wide operands:
// with F of class N the closing products are E F (difference x N), H G (sum x sum-of-a-sum, limbs 2^30 x 2^30.6: the same
// "difference x sum" column bound as the mixed addition, 0.97 * 2^64 in tests/test_limb_bounds_te.py), E H and F G.  With 14
// limbs one operand of EVERY product must be normalised: E and G are as well.  (Until round 4 all three were normalised for
// N = 9 too: two carry chains of 27 dependent instructions per addition that nothing needed.)
template <int N> TE_HD ete_t<N> ete_add(const ete_t<N>& a, const ete_t<N>& b) {
  const fel<N> in1[4] = {fe_norm(fe_sub<2>(a.y, a.x)), fe_norm_if_needed(fe_add(a.y, a.x)), a.t, a.z};
  const fel<N> in2[4] = {fe_sub<2>(b.y, b.x), fe_add(b.y, b.x), b.t, b.z};
  fel<N> p1[4];
  fe_mul_x<4>(in1, in2, p1);
  const fel<N> &A = p1[0], &B = p1[1];
  const fel<N> C = fe_mul_k2d(p1[2]);                 // N = 9: the 13-bit constant applied limb-wise, not a product
  const fel<N> D = fe_add(p1[3], p1[3]);
  const fel<N> E = fe_norm_if_needed(fe_sub<2>(B, A));
  const fel<N> H = fe_add(B, A);
  const fel<N> F = fe_norm(fe_sub<2>(D, C));
  const fel<N> G = fe_norm_if_needed(fe_add(D, C));
  const fel<N> l[4] = {E, H, E, F}, rr[4] = {F, G, H, G};
  fel<N> o[4];
  fe_mul_x<4>(l, rr, o);
  ete_t<N> r;
  r.x = o[0]; r.y = o[1]; r.t = o[2]; r.z = o[3];
  return r;
}

}  // namespace te
