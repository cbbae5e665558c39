// curve.hpp -- point arithmetic on -x^2 + y^2 = 1 + 3021 x^2 y^2 over fp (a = -1, d = 3021;
// reference/params/AleoConstants.ts:2-4, reference/utils/FieldMath.ts:104-137).
//
// Replaces add_points / double_point / negate_point of the reference
// (wgsl/curve/ec.template.wgsl:7-66, wgsl/cuzk/smvp.template.wgsl:47-56).  The reference adds two
// full extended points with add-2008-hwcd (10 field products + T = x*y recomputed per gathered
// point, smvp.template.wgsl:107-108).  Here the n input points are converted ONCE into a 96-byte
// record ((y-x)/2, (y+x)/2, d*x*y) and bucket accumulation is a 7-product mixed addition with no
// reduction step at all (bounds below).  Any correct group law gives the same affine result, which
// is the only thing compared with the reference.
#pragma once
#include "fp.hpp"

namespace te {

// Extended twisted Edwards accumulator (X : Y : Z : T), x = X/Z, y = Y/Z, T = XY/Z.
// Invariant: every coordinate < 2p (lazy Montgomery form).  128 bytes, x | y | z | t.
struct ete { fp x, y, z, t; };

// Precomputed affine point: hm = (y - x)/2, hp = (y + x)/2, dt = d*x*y, all canonical (< p; a
// negated record may hold dt = p).  96 bytes.
struct pnt { fp hm, hp, dt; };

TE_HD ete ete_identity() { ete r; r.x = fp_zero(); r.y = fp_R1(); r.z = fp_R1(); r.t = fp_zero(); return r; }

// -(x, y) = (-x, y): swaps (y-x)/2 and (y+x)/2, negates d*x*y.
TE_HD pnt pnt_neg(const pnt& a) { pnt r; r.hm = a.hp; r.hp = a.hm; r.dt = fp_neg<1>(a.dt); return r; }
TE_HD pnt pnt_cneg(const pnt& a, bool neg) {
  pnt r; const fp ndt = fp_neg<1>(a.dt);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r.hm.v[i] = neg ? a.hp.v[i] : a.hm.v[i];
    r.hp.v[i] = neg ? a.hm.v[i] : a.hp.v[i];
    r.dt.v[i] = neg ? ndt.v[i] : a.dt.v[i];
  }
  return r;
}

// Affine (x, y) in Montgomery form, canonical -> record.
TE_HD pnt pnt_from_affine_mont(const fp& xm, const fp& ym) {
  pnt r;
  r.hm = fp_half(fp_csub<1>(fp_sub<1>(ym, xm)));
  r.hp = fp_half(fp_csub<1>(fp_add(ym, xm)));
  r.dt = fp_csub<1>(mont_mul(mont_mul(xm, ym), fp_D_MONT()));   // < 1.08p -> canonical
  return r;
}

// Mixed addition acc + b, 7 products, unified and complete (a = -1 is a square, d is not).
// With A' = (Y1-X1)(y2-x2)/2, B' = (Y1+X1)(y2+x2)/2:  E = B'-A' = X1 y2 + Y1 x2,
// H = B'+A' = Y1 y2 + X1 x2, C = d T1 x2 y2, F = Z1 - C, G = Z1 + C;  (X3,Y3,T3,Z3) = (EF, GH, EH, FG).
// Bounds (units of p; inputs X1,Y1,Z1,T1 < 2, record <= 1):
//   Y1-X1+2p < 4, Y1+X1 < 4 -> A', B' < 1.30;  C < 1.15;  E < 3.30, H < 2.60, F < 4, G < 3.15
//   X3 < 1.97, Y3 < 1.60, T3 < 1.63, Z3 < 1.92   -- all < 2: the invariant holds with no reduction.
TE_HD ete ete_madd(const ete& a, const pnt& b) {
  const fp A = mont_mul(fp_sub<2>(a.y, a.x), b.hm);
  const fp B = mont_mul(fp_add(a.y, a.x), b.hp);
  const fp C = mont_mul(a.t, b.dt);
  const fp E = fp_sub<2>(B, A);
  const fp H = fp_add(B, A);
  const fp F = fp_sub<2>(a.z, C);
  const fp G = fp_add(a.z, C);
  ete r;
  r.x = mont_mul(E, F);
  r.y = mont_mul(H, G);
  r.t = mont_mul(E, H);
  r.z = mont_mul(G, F);
  return r;
}

// Full addition a + b of two accumulators (add-2008-hwcd-3 shape, k = 2d), 9 products.
// Bounds (inputs < 2): A, B < 2.17; C < 1.10; D < 2.60; E = B-A+3p < 5.17; H < 4.34; F = D-C+2p < 4.6;
// G < 3.7;  outputs < 2.75 -> one conditional subtraction of 2p each restores < 2.
TE_HD ete ete_add(const ete& a, const ete& b) {
  const fp A = mont_mul(fp_sub<2>(a.y, a.x), fp_sub<2>(b.y, b.x));
  const fp B = mont_mul(fp_add(a.y, a.x), fp_add(b.y, b.x));
  const fp C = mont_mul(mont_mul(a.t, b.t), fp_K2D_MONT());
  const fp zz = mont_mul(a.z, b.z);
  const fp D = fp_add(zz, zz);
  const fp E = fp_sub<3>(B, A);
  const fp H = fp_add(B, A);
  const fp F = fp_sub<2>(D, C);
  const fp G = fp_add(D, C);
  ete r;
  r.x = fp_csub<2>(mont_mul(E, F));
  r.y = fp_csub<2>(mont_mul(G, H));
  r.t = fp_csub<2>(mont_mul(E, H));
  r.z = fp_csub<2>(mont_mul(F, G));
  return r;
}

}  // namespace te
