// curve377.hpp -- group law of BLS12-377 G1, y^2 = x^3 + 1 over fq (BASELINE config 5).
//
// The reference names "projective algorithms" for this curve (README.md:285-287) and ships no code for it.  Used here:
// the COMPLETE projective formulas of Renes-Costello-Batina 2016 for a = 0 (Algorithms 7 and 8, b3 = 3b = 3): one
// formula for addition, doubling and the point at infinity (0 : 1 : 0), so the bucket kernels need no special cases --
// exactly what the complete twisted-Edwards law gives the other curve.  (Exceptions exist only for points of even
// order; MSM inputs live in the subgroup of odd prime order r.)
//
// Limb classes (fq377.hpp): every product needs one normalised operand.  Coordinates are stored "loose" (a difference
// or a sum of two products) and normalised on entry; sums that meet in a product are normalised on one side.
#pragma once
#include "fq377.hpp"

namespace te377 {

struct g1p { fq x, y, z; };      // projective accumulator (X : Y : Z), Montgomery form; 168 bytes, x | y | z
struct g1a { fq x, y; };         // affine input point in Montgomery form, class N, values < 1.01 q; 112 bytes in a 128-byte slot

TE_HD g1p g1_identity() { g1p r; r.x = fq_zero(); r.y = fq_R1(); r.z = fq_zero(); return r; }

// -(x, y) = (x, -y): y' = 2q - y (limbs < 2^30.6)
TE_HD g1a g1a_cneg(const g1a& a, bool neg) {
  g1a r; const fq ny = fq_neg<2>(a.y);
  r.x = a.x;
#pragma unroll
  for (int i = 0; i < NL; i++) r.y.v[i] = neg ? ny.v[i] : a.y.v[i];
  return r;
}

// plain integers (class N, any 384-bit value) -> Montgomery record: x * R^2 / R, y * R^2 / R
TE_HD g1a g1a_from_raw(const fq& x, const fq& y) {
  const fq a[2] = {x, y}, b[2] = {fq_R2(), fq_R2()};
  fq o[2];
  mont_mul_x<2>(a, b, o);
  g1a r; r.x = o[0]; r.y = o[1];
  return r;
}

// the six closing products and sums shared by both algorithms
TE_HD g1p g1_finish(const fq& t0x3, const fq& t1n, const fq& t3n, const fq& t4, const fq& y3n, const fq& z3n) {
  const fq l1[3] = {y3n, t3n, y3n}, r1[3] = {t4, t1n, t0x3};
  const fq l2[3] = {t1n, t3n, z3n}, r2[3] = {z3n, t0x3, t4};
  fq p[3], s[3];
  mont_mul_x<3>(l1, r1, p);          // p0 = t4*Y3, p1 = t3*t1, p2 = Y3*t0
  mont_mul_x<3>(l2, r2, s);          // s0 = t1*Z3, s1 = t0*t3, s2 = Z3*t4
  g1p o;
  o.x = fq_sub<2>(p[1], p[0]);
  o.y = fq_add(s[0], p[2]);
  o.z = fq_add(s[2], s[1]);
  return o;
}

// Mixed addition acc + b (RCB16 Algorithm 8, Z2 = 1): 11 products.
TE_HD g1p g1_madd(const g1p& a, const g1a& b) {
  const fq X1 = fq_norm(a.x), Y1 = fq_norm(a.y), Z1 = fq_norm(a.z);
  const fq bsum = fq_norm(fq_add(b.x, b.y));
  const fq l1[3] = {X1, Y1, bsum}, r1[3] = {b.x, b.y, fq_add(X1, Y1)};
  const fq l2[2] = {Z1, Z1}, r2[2] = {b.y, b.x};
  fq p[3], s[2];
  mont_mul_x<3>(l1, r1, p);          // t0 = X1 x2, t1 = Y1 y2, t3 = (x2 + y2)(X1 + Y1)
  mont_mul_x<2>(l2, r2, s);          // y2 Z1, x2 Z1
  const fq t3n = fq_norm(fq_sub<2>(fq_sub<2>(p[2], p[0]), p[1]));              // X1 y2 + Y1 x2
  const fq t4 = fq_add(s[0], Y1);                                               // Y1 + y2 Z1        (limbs < 2^30)
  const fq y3n = fq_norm(fq_mul3(fq_add(s[1], X1)));                            // 3 (X1 + x2 Z1)
  const fq t0x3 = fq_mul3(p[0]);                                                // 3 X1 x2           (limbs < 2^30.6)
  const fq z3n = fq_norm(fq_add(p[1], fq_mul3(Z1)));                            // Y1 y2 + 3 Z1
  const fq t1n = fq_norm(fq_sub<4>(fq_sub<4>(fq_sub<4>(p[1], Z1), Z1), Z1));    // Y1 y2 - 3 Z1   (Z1 may reach 2q: offset 4q)
  return g1_finish(t0x3, t1n, t3n, t4, y3n, z3n);
}

// Full addition a + b (RCB16 Algorithm 7): 12 products.
TE_HD g1p g1_add(const g1p& a, const g1p& b) {
  const fq X1 = fq_norm(a.x), Y1 = fq_norm(a.y), Z1 = fq_norm(a.z);
  const fq X2 = fq_norm(b.x), Y2 = fq_norm(b.y), Z2 = fq_norm(b.z);
  const fq l1[3] = {X1, Y1, Z1}, r1[3] = {X2, Y2, Z2};
  const fq l2[3] = {fq_norm(fq_add(X1, Y1)), fq_norm(fq_add(Y1, Z1)), fq_norm(fq_add(X1, Z1))};
  const fq r2[3] = {fq_add(X2, Y2), fq_add(Y2, Z2), fq_add(X2, Z2)};
  fq t[3], u[3];
  mont_mul_x<3>(l1, r1, t);          // t0 = X1 X2, t1 = Y1 Y2, t2 = Z1 Z2
  mont_mul_x<3>(l2, r2, u);          // (X1+Y1)(X2+Y2), (Y1+Z1)(Y2+Z2), (X1+Z1)(X2+Z2)
  const fq t3n = fq_norm(fq_sub<2>(fq_sub<2>(u[0], t[0]), t[1]));               // X1 Y2 + X2 Y1
  const fq t4n = fq_norm(fq_sub<2>(fq_sub<2>(u[1], t[1]), t[2]));               // Y1 Z2 + Y2 Z1
  const fq y3n = fq_norm(fq_mul3(fq_norm(fq_sub<2>(fq_sub<2>(u[2], t[0]), t[2]))));   // 3 (X1 Z2 + X2 Z1)
  const fq t0x3 = fq_mul3(t[0]);
  const fq z3n = fq_norm(fq_add(t[1], fq_mul3(t[2])));                          // Y1 Y2 + 3 Z1 Z2
  const fq t1n = fq_norm(fq_sub<2>(fq_sub<2>(fq_sub<2>(t[1], t[2]), t[2]), t[2]));   // Y1 Y2 - 3 Z1 Z2
  return g1_finish(t0x3, t1n, t3n, t4n, y3n, z3n);
}

}  // namespace te377
