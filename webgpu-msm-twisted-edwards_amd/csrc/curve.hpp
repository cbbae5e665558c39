// curve.hpp -- point arithmetic on -x^2 + y^2 = 1 + 3021 x^2 y^2 over fp (a = -1, d = 3021;
// reference/params/AleoConstants.ts:2-4, reference/utils/FieldMath.ts:104-137).
//
// Replaces add_points / double_point / negate_point of the reference
// (wgsl/curve/ec.template.wgsl:7-66, wgsl/cuzk/smvp.template.wgsl:47-56).  The reference adds two
// full extended points with add-2008-hwcd (10 field products + T = x*y recomputed per gathered
// point, smvp.template.wgsl:107-108).  Here the n input points are converted ONCE into a record
// ((y-x)/2, (y+x)/2, d*x*y) and bucket accumulation is a 7-product mixed addition with no
// reduction modulo p at all (limb-magnitude rules in fp.hpp).  Any correct group law gives the same affine result, which
// is the only thing compared with the reference.
#pragma once
#include "fp.hpp"

namespace te {

// Extended twisted Edwards accumulator (X : Y : Z : T), x = X/Z, y = Y/Z, T = XY/Z.
// Every coordinate is a mont_mul output: limb class N, value < 1.1p.  144 bytes, x | y | z | t.
struct ete { fp x, y, z, t; };

// Precomputed affine point: hm = (y - x)/2, hp = (y + x)/2, dt = -d*x*y (mod p, lazily reduced: values < 1.1p,
// class N; a negated record holds 4p - dt with limbs < 2^30).  108 bytes, stored in 128-byte slots.
// dt carries the MINUS sign so that in ete_madd F = Z1 - C becomes a sum and no product meets two differences.
struct pnt { fp hm, hp, dt; };

TE_HD ete ete_identity() { ete r; r.x = fp_zero(); r.y = fp_R1(); r.z = fp_R1(); r.t = fp_zero(); return r; }

// -(x, y) = (-x, y): swaps (y-x)/2 and (y+x)/2, negates d*x*y.
TE_HD pnt pnt_cneg(const pnt& a, bool neg) {
  pnt r; const fp ndt = fp_neg<4>(a.dt);
#pragma unroll
  for (int i = 0; i < NL; i++) {
    r.hm.v[i] = neg ? a.hp.v[i] : a.hm.v[i];
    r.hp.v[i] = neg ? a.hm.v[i] : a.hp.v[i];
    r.dt.v[i] = neg ? ndt.v[i] : a.dt.v[i];
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

// Mixed addition acc + b, 7 products, unified and complete (a = -1 is a square, d is not).
// With A' = (Y1-X1)(y2-x2)/2, B' = (Y1+X1)(y2+x2)/2:  E = B'-A' = X1 y2 + Y1 x2,
// H = B'+A' = Y1 y2 + X1 x2, C = d T1 x2 y2, F = Z1 - C, G = Z1 + C;  (X3,Y3,T3,Z3) = (EF, HG, EH, GF).
// The record holds -d x2 y2, so the product below is C' = -C and F = Z1 + C' is a SUM, G = Z1 - C' the difference.
// Limb classes: A' = D x N, B' = S x N, C' = N x (<2^30);  E and G are D (differences in offset form), H and F are S:
// every closing product EF, HG, EH, GF is D x S (9 * 2^30.6 * 2^30 + 8 * 2^58 = 0.9 * 2^64) -- no operand is normalised.
// Values: everything is below 5p before a product and below 1.1p after it; nothing is reduced mod p.
TE_HD ete ete_madd(const ete& a, const pnt& b) {
  const fp in1[3] = {fp_sub<2>(a.y, a.x), fp_add(a.y, a.x), a.t}, in2[3] = {b.hm, b.hp, b.dt};
  fp abc[3];
  mont_mul_x<3>(in1, in2, abc);
  const fp &A = abc[0], &B = abc[1], &Cn = abc[2];
  const fp E = fp_sub<2>(B, A);
  const fp H = fp_add(B, A);
  const fp F = fp_add(a.z, Cn);
  const fp G = fp_sub<2>(a.z, Cn);
  const fp l[4] = {E, H, E, G}, rr[4] = {F, G, H, F};
  fp o[4];
  mont_mul_x<4>(l, rr, o);
  ete r;
  r.x = o[0]; r.y = o[1]; r.t = o[2]; r.z = o[3];
  return r;
}

// Full addition a + b of two accumulators (add-2008-hwcd-3 shape, k = 2d), 9 products.
// (Y1-X1), F = 2 Z1Z2 - C and G = 2 Z1Z2 + C are normalised so that no product sees two wide operands.
TE_HD ete ete_add(const ete& a, const ete& b) {
  const fp in1[4] = {fp_norm(fp_sub<2>(a.y, a.x)), fp_add(a.y, a.x), a.t, a.z};
  const fp in2[4] = {fp_sub<2>(b.y, b.x), fp_add(b.y, b.x), b.t, b.z};
  fp p1[4];
  mont_mul_x<4>(in1, in2, p1);
  const fp &A = p1[0], &B = p1[1];
  const fp C = mont_mul(p1[2], fp_K2D_MONT());
  const fp D = fp_add(p1[3], p1[3]);
  const fp E = fp_norm(fp_sub<2>(B, A));
  const fp H = fp_add(B, A);
  const fp F = fp_norm(fp_sub<2>(D, C));
  const fp G = fp_norm(fp_add(D, C));
  const fp l[4] = {E, H, E, F}, rr[4] = {F, G, H, G};
  fp o[4];
  mont_mul_x<4>(l, rr, o);
  ete r;
  r.x = o[0]; r.y = o[1]; r.t = o[2]; r.z = o[3];
  return r;
}

}  // namespace te
