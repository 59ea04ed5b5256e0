// rt_math.h -- float32 building blocks of the render path, usable from HIP device code and
// (for exhaustive bit-comparison tests on a CPU) from plain C++.
//
// Every routine keeps the operation order and rounding of the reference routine it names, so the
// accept/reject decisions and the accepted values are bit-identical; translation units including
// this header must be built with -ffp-contract=off.
#ifndef RT_MATH_H
#define RT_MATH_H

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define RT_HD inline
#endif

struct RtVec3 { float x, y, z; };

RT_HD float rt_dot(const RtVec3& a, const RtVec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // vector.cu:106-108
RT_HD float rt_len(const RtVec3& v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }              // normf(3, v)
RT_HD RtVec3 rt_normalized(RtVec3 v) {                                                                  // vector.cu:110-112
  const float m = rt_len(v);
  v.x /= m; v.y /= m; v.z /= m;
  return v;
}

// Solution of [c0 c1 c2] x = b by cofactors, operation order of triangle.cu:121-133 and
// octtree.cu:301-313 (the reference writes the same twenty-odd lines in both places).
RT_HD void rt_cofactor_solve(const RtVec3& c0, const RtVec3& c1, const RtVec3& c2, const RtVec3& b,
                             float& x0, float& x1, float& x2) {
  const float m00 = c1.y * c2.z - c2.y * c1.z;
  const float m01 = -c0.y * c2.z + c0.z * c2.y;
  const float m02 = c0.y * c1.z - c0.z * c1.y;
  const float m10 = -c1.x * c2.z + c1.z * c2.x;
  const float m11 = c0.x * c2.z - c0.z * c2.x;
  const float m12 = -c0.x * c1.z + c0.z * c1.x;
  const float m20 = c1.x * c2.y - c1.y * c2.x;
  const float m21 = -c0.x * c2.y + c0.y * c2.x;
  const float m22 = c0.x * c1.y - c0.y * c1.x;
  const float det = c0.x * m00 + c0.y * m10 + c0.z * m20;
  x0 = (m00 * b.x + m10 * b.y + m20 * b.z) / det;
  x1 = (m01 * b.x + m11 * b.y + m21 * b.z) / det;
  x2 = (m02 * b.x + m12 * b.y + m22 * b.z) / det;
}

// Ray / triangle test of triangle.cu:107-152 on the record (v0, e1 = v1-v0, e2 = v2-v0).
// Returns true and the accepted (t, beta, gamma) iff the reference would accept with closest hit t_max.
RT_HD bool rt_triangle_test(const RtVec3& v0, const RtVec3& e1, const RtVec3& e2, const RtVec3& eye,
                            const RtVec3& dir, float t_max, float& t, float& bet, float& gam) {
  const RtVec3 b = {eye.x - v0.x, eye.y - v0.y, eye.z - v0.z};
  const RtVec3 nd = {-dir.x, -dir.y, -dir.z};
  float x0, x1, x2;
  rt_cofactor_solve(e1, e2, nd, b, x0, x1, x2);
  const float alp = 1 - x0 - x1;
  if (0 < x2 && x2 < t_max && 0 <= x0 && x0 <= 1 && 0 <= x1 && x1 <= 1 && 0 <= alp) {
    t = x2; bet = x0; gam = x1;
    return true;
  }
  return false;
}

// One face of the node test: quad v0 + a*(v1-v0) + b*(v3-v0).
RT_HD bool rt_face_test(const RtVec3& v0, const RtVec3& v1, const RtVec3& v3, const RtVec3& eye, const RtVec3& dir,
                        float t_max) {
  const RtVec3 c0 = {v1.x - v0.x, v1.y - v0.y, v1.z - v0.z};
  const RtVec3 c1 = {v3.x - v0.x, v3.y - v0.y, v3.z - v0.z};
  const RtVec3 c2 = {-dir.x, -dir.y, -dir.z};
  const RtVec3 b = {eye.x - v0.x, eye.y - v0.y, eye.z - v0.z};
  float alp, bet, t1;
  rt_cofactor_solve(c0, c1, c2, b, alp, bet, t1);
  return 0 < t1 && t1 < t_max && 0 <= bet && bet <= 1 && 0 <= alp && alp <= 1;
}

// Node acceptance of octtree.cu:276-321, evaluated literally: six quads in the reference's vertex
// order, accepted when any of them is crossed at 0 < t < t_max inside its unit square.
RT_HD bool rt_node_accept_literal(const RtVec3& ini, const RtVec3& end, const RtVec3& eye, const RtVec3& dir, float t_max) {
  if (rt_face_test({end.x, ini.y, ini.z}, {ini.x, ini.y, ini.z}, {end.x, end.y, ini.z}, eye, dir, t_max)) return true;
  if (rt_face_test({ini.x, ini.y, ini.z}, {ini.x, ini.y, end.z}, {ini.x, end.y, ini.z}, eye, dir, t_max)) return true;
  if (rt_face_test({ini.x, ini.y, end.z}, {end.x, ini.y, end.z}, {ini.x, end.y, end.z}, eye, dir, t_max)) return true;
  if (rt_face_test({end.x, ini.y, end.z}, {end.x, ini.y, ini.z}, {end.x, end.y, end.z}, eye, dir, t_max)) return true;
  if (rt_face_test({ini.x, end.y, end.z}, {end.x, end.y, end.z}, {ini.x, end.y, ini.z}, eye, dir, t_max)) return true;
  if (rt_face_test({ini.x, ini.y, ini.z}, {end.x, ini.y, ini.z}, {ini.x, ini.y, end.z}, eye, dir, t_max)) return true;
  return false;
}

// Ray / sphere test of spheres.cu:73-100 (assumes |dir| = 1). Updates t when accepted.
RT_HD bool rt_sphere_test(const RtVec3& c, float r, const RtVec3& eye, const RtVec3& dir, float& t) {
  const RtVec3 ec = {eye.x - c.x, eye.y - c.y, eye.z - c.z};
  const float B = 2 * rt_dot(ec, dir);
  const float C = rt_dot(ec, ec) - r * r;
  float det = B * B - 4 * C;
  if (det >= 0) {
    det = sqrtf(det);
    float t2 = -B + det, t1 = -B - det;
    if (t2 >= 0) {
      if (t1 > 0 && t1 < 2 * t) { t = t1 / 2; return true; }
      if (t1 < 0 && t2 < 2 * t) { t = t2 / 2; return true; }
    }
  }
  return false;
}

// ------------------------------------------------------------------------------------------------
// Division-free evaluation of the same accept / reject decisions.
//
// The reference decides with correctly rounded quotients q = fl(N / D) compared against 0, 1 and the
// running closest t.  For finite floats, barring underflow of a quotient to zero:
//   q >= 0      <=>  N and D agree in sign (or N is zero)
//   q <= 1      <=>  |N| <= |D|          (no float lies in (|D|, |D|(1 + 2^-24)], the interval that rounds to 1)
//   q <  T      <=>  N' < Tm * D'        (exact in double: Tm = midpoint of T and its predecessor, 25 x 24 bits;
//                                          on a tie the quotient rounds to the neighbour with the even mantissa)
// Octree boxes are cubes whose side is a power of two (root +-16 split at midpoints, octtree.cu:95), so
// the extent factors of octtree.cu:295-313 scale numerators and denominators exactly and cancel:
// a face on axis k crossed at t = fl(b_k / -d_k) has in-plane offset fl(P_i / d_k) with
// P_i = fl(fl(d_k b_i) - fl(d_i b_k)), to be compared with 0 and the (signed) side.
// ------------------------------------------------------------------------------------------------
RT_HD unsigned rt_bits(float x) { union { float f; unsigned u; } c; c.f = x; return c.u; }
RT_HD float rt_from_bits(unsigned u) { union { float f; unsigned u; } c; c.u = u; return c.f; }

struct RtRayConst {        // per ray
  float ad[3];             // |d_k|
  unsigned sd[3];          // sign bit of d_k
};
struct RtTConst {          // per value of the running closest t
  double td[3];            // Tm * |d_k|, one ulp higher when a tie counts as "less than T"
  double tm;               // Tm itself (nudged the same way), for the triangle test
};

RT_HD RtRayConst rt_ray_const(const RtVec3& d) {
  RtRayConst r;
  r.ad[0] = fabsf(d.x); r.ad[1] = fabsf(d.y); r.ad[2] = fabsf(d.z);
  r.sd[0] = rt_bits(d.x) & 0x80000000u; r.sd[1] = rt_bits(d.y) & 0x80000000u; r.sd[2] = rt_bits(d.z) & 0x80000000u;
  return r;
}

RT_HD double rt_next_up(double p) {   // p >= 0 finite -> next double above; NaN stays NaN
  union { double f; unsigned long long u; } c; c.f = p; c.u += 1ull; return c.f;
}

// T > 0.
RT_HD RtTConst rt_t_const(const RtRayConst& r, float T) {
  const unsigned pb = rt_bits(T) - 1u;                  // predecessor of T
  const double tm = 0.5 * ((double)T + (double)rt_from_bits(pb));
  const bool tie_is_less = (pb & 1u) == 0u;            // the midpoint rounds to the predecessor
  RtTConst k;
  for (int a = 0; a < 3; a++) {
    const double p = tm * (double)r.ad[a];
    k.td[a] = tie_is_less ? rt_next_up(p) : p;
  }
  k.tm = tm;   // ties for the triangle test are resolved at the use site (the product there is not exact in general)
  return k;
}

// One face: plane axis k with b_k = e_k - plane; in-plane axes i, j with b measured from the face's
// first vertex; neg_i / neg_j = 0x80000000 when the side runs in the negative direction from that vertex.
RT_HD bool rt_face_exact(float dk_bi, float di_bk, float dk_bj, float dj_bk, float bk, unsigned sdk, unsigned neg_i,
                         unsigned neg_j, float side_adk, double tdk) {
  const float bpp = rt_from_bits(rt_bits(bk) ^ sdk ^ 0x80000000u);      // b_k * sign(-d_k): t > 0 <=> bpp > 0
  const float qi = rt_from_bits(rt_bits(dk_bi - di_bk) ^ sdk ^ neg_i);
  const float qj = rt_from_bits(rt_bits(dk_bj - dj_bk) ^ sdk ^ neg_j);
  return (bpp > 0.f) & ((double)bpp < tdk) & (qi >= 0.f) & (qi <= side_adk) & (qj >= 0.f) & (qj <= side_adk);
}

// Node acceptance, same decision as rt_node_accept_literal for a cube with ini = (lo.x, lo.y, hi.z),
// end = (hi.x, hi.y, lo.z) and power-of-two side.
RT_HD bool rt_node_accept_exact(const RtVec3& ini, const RtVec3& end, const RtVec3& e, const RtVec3& d, const RtRayConst& rc,
                                const RtTConst& tc) {
  const float side = end.x - ini.x;
  const float bxi = e.x - ini.x, bxe = e.x - end.x, byi = e.y - ini.y, bye = e.y - end.y, bzi = e.z - ini.z, bze = e.z - end.z;
  const float sx = side * rc.ad[0], sy = side * rc.ad[1], sz = side * rc.ad[2];
  const unsigned NEG = 0x80000000u;
  const float dz_bxe = d.z * bxe, dz_bxi = d.z * bxi, dz_byi = d.z * byi, dz_bye = d.z * bye;
  const float dx_bzi = d.x * bzi, dx_bze = d.x * bze, dx_byi = d.x * byi, dx_bye = d.x * bye;
  const float dy_bzi = d.y * bzi, dy_bze = d.y * bze, dy_bxi = d.y * bxi, dy_bxe = d.y * bxe;
  bool acc;
  acc  = rt_face_exact(dz_bxe, dx_bzi, dz_byi, dy_bzi, bzi, rc.sd[2], NEG, 0u, sz, tc.td[2]);   // z = ini.z, from (end.x, ini.y)
  acc |= rt_face_exact(dx_bzi, dz_bxi, dx_byi, dy_bxi, bxi, rc.sd[0], NEG, 0u, sx, tc.td[0]);   // x = ini.x, from (ini.y, ini.z)
  acc |= rt_face_exact(dz_bxi, dx_bze, dz_byi, dy_bze, bze, rc.sd[2], 0u, 0u, sz, tc.td[2]);    // z = end.z, from (ini.x, ini.y)
  acc |= rt_face_exact(dx_bze, dz_bxe, dx_byi, dy_bxe, bxe, rc.sd[0], 0u, 0u, sx, tc.td[0]);    // x = end.x, from (ini.y, end.z)
  acc |= rt_face_exact(dy_bxi, dx_bye, dy_bze, dz_bye, bye, rc.sd[1], 0u, 0u, sy, tc.td[1]);    // y = end.y, from (ini.x, end.z)
  acc |= rt_face_exact(dy_bxi, dx_byi, dy_bzi, dz_byi, byi, rc.sd[1], 0u, NEG, sy, tc.td[1]);   // y = ini.y, from (ini.x, ini.z)
  return acc;
}

// Triangle numerators and determinant with the reference's rounding (triangle.cu:117-133, before the divisions).
struct RtTriNum { float det, nb, ng, nt; };

RT_HD RtTriNum rt_triangle_numerators(const RtVec3& v0, const RtVec3& e1, const RtVec3& e2, const RtVec3& eye, const RtVec3& dir) {
  const RtVec3 b = {eye.x - v0.x, eye.y - v0.y, eye.z - v0.z};
  const RtVec3 c2 = {-dir.x, -dir.y, -dir.z};
  const float m00 = e2.y * c2.z - c2.y * e2.z;
  const float m10 = -e2.x * c2.z + e2.z * c2.x;
  const float m20 = e2.x * c2.y - e2.y * c2.x;
  RtTriNum n;
  n.det = e1.x * m00 + e1.y * m10 + e1.z * m20;
  n.nb = m00 * b.x + m10 * b.y + m20 * b.z;
  const float m01 = -e1.y * c2.z + e1.z * c2.y;
  const float m11 = e1.x * c2.z - e1.z * c2.x;
  const float m21 = -e1.x * c2.y + e1.y * c2.x;
  n.ng = m01 * b.x + m11 * b.y + m21 * b.z;
  const float m02 = e1.y * e2.z - e1.z * e2.y;
  const float m12 = -e1.x * e2.z + e1.z * e2.x;
  const float m22 = e1.x * e2.y - e1.y * e2.x;
  n.nt = m02 * b.x + m12 * b.y + m22 * b.z;
  return n;
}

// Everything of the triangle accept rule that needs no division: 0 <= beta <= 1, 0 <= gamma <= 1, 0 < t.
RT_HD bool rt_triangle_candidate(const RtTriNum& n) {
  const unsigned s = rt_bits(n.det) & 0x80000000u;
  const float ad = fabsf(n.det);
  const float nb = rt_from_bits(rt_bits(n.nb) ^ s), ng = rt_from_bits(rt_bits(n.ng) ^ s), nt = rt_from_bits(rt_bits(n.nt) ^ s);
  return (ad > 0.f) & (nb >= 0.f) & (nb <= ad) & (ng >= 0.f) & (ng <= ad) & (nt > 0.f);
}

// Final decision for a candidate, with the reference's quotients (triangle.cu:131-137).
RT_HD bool rt_triangle_finish(const RtTriNum& n, float t_max, float& t, float& bet, float& gam) {
  const float x0 = n.nb / n.det, x1 = n.ng / n.det, x2 = n.nt / n.det;
  const float alp = 1 - x0 - x1;
  if (0 < x2 && x2 < t_max && 0 <= x0 && x0 <= 1 && 0 <= x1 && x1 <= 1 && 0 <= alp) { t = x2; bet = x0; gam = x1; return true; }
  return false;
}


// ------------------------------------------------------------------------------------------------
// The same node decision as one number.  Every condition of a face is a sign condition on a float:
//   t > 0        <=> bpp > 0                 t < T   <=> tdf_k - bpp > 0   (tdf_k = smallest float >= Tm |d_k|)
//   offset >= 0  <=> q >= 0                  offset <= side <=> side|d_k| - q >= 0
// (the sign of a float difference is the sign of the exact difference).  score(face) = min of its six
// terms, S = max over the faces:  S > 0 => some face satisfies everything strictly => accept;
// S < 0 (or NaN) => every face violates something => reject;  S == 0 => a boundary case, to be settled
// with rt_node_accept_exact.
// ------------------------------------------------------------------------------------------------
struct RtTFloat { float tdf[3]; };

RT_HD float rt_round_up_to_float(double y) {   // smallest float >= y, for y >= 0 (NaN stays NaN)
  float f = (float)y;
  if ((double)f < y) f = rt_from_bits(rt_bits(f) + 1u);
  return f;
}
RT_HD RtTFloat rt_t_float(const RtRayConst& r, float T) {
  const RtTConst c = rt_t_const(r, T);
  RtTFloat f;
  f.tdf[0] = rt_round_up_to_float(c.td[0]); f.tdf[1] = rt_round_up_to_float(c.td[1]); f.tdf[2] = rt_round_up_to_float(c.td[2]);
  return f;
}
RT_HD float rt_min3(float a, float b, float c) { return fminf(fminf(a, b), c); }
RT_HD float rt_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// Median of three.  On the device this is one v_med3_f32; the two forms agree whenever no input is a NaN, and the
// only caller makes its decision "undecided" in every case where a NaN can occur.
RT_HD float rt_med3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fmed3f(a, b, c);
#else
  return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
#endif
}

RT_HD float rt_face_score(float dk_bi, float di_bk, float dk_bj, float dj_bk, float bk, unsigned sdk, unsigned neg_i,
                          unsigned neg_j, float side_adk, float tdfk) {
  const float bpp = rt_from_bits(rt_bits(bk) ^ sdk ^ 0x80000000u);
  const float qi = rt_from_bits(rt_bits(dk_bi - di_bk) ^ sdk ^ neg_i);
  const float qj = rt_from_bits(rt_bits(dk_bj - dj_bk) ^ sdk ^ neg_j);
  return fminf(rt_min3(bpp, tdfk - bpp, qi), rt_min3(side_adk - qi, qj, side_adk - qj));
}

RT_HD float rt_node_score(const RtVec3& ini, const RtVec3& end, const RtVec3& e, const RtVec3& d, const RtRayConst& rc,
                          const RtTFloat& tf) {
  const float side = end.x - ini.x;
  const float bxi = e.x - ini.x, bxe = e.x - end.x, byi = e.y - ini.y, bye = e.y - end.y, bzi = e.z - ini.z, bze = e.z - end.z;
  const float sx = side * rc.ad[0], sy = side * rc.ad[1], sz = side * rc.ad[2];
  const unsigned NEG = 0x80000000u;
  const float dz_bxe = d.z * bxe, dz_bxi = d.z * bxi, dz_byi = d.z * byi, dz_bye = d.z * bye;
  const float dx_bzi = d.x * bzi, dx_bze = d.x * bze, dx_byi = d.x * byi, dx_bye = d.x * bye;
  const float dy_bzi = d.y * bzi, dy_bze = d.y * bze, dy_bxi = d.y * bxi, dy_bxe = d.y * bxe;
  const float s0 = rt_face_score(dz_bxe, dx_bzi, dz_byi, dy_bzi, bzi, rc.sd[2], NEG, 0u, sz, tf.tdf[2]);
  const float s1 = rt_face_score(dx_bzi, dz_bxi, dx_byi, dy_bxi, bxi, rc.sd[0], NEG, 0u, sx, tf.tdf[0]);
  const float s2 = rt_face_score(dz_bxi, dx_bze, dz_byi, dy_bze, bze, rc.sd[2], 0u, 0u, sz, tf.tdf[2]);
  const float s3 = rt_face_score(dx_bze, dz_bxe, dx_byi, dy_bxe, bxe, rc.sd[0], 0u, 0u, sx, tf.tdf[0]);
  const float s4 = rt_face_score(dy_bxi, dx_bye, dy_bze, dz_bye, bye, rc.sd[1], 0u, 0u, sy, tf.tdf[1]);
  const float s5 = rt_face_score(dy_bxi, dx_byi, dy_bzi, dz_byi, byi, rc.sd[1], 0u, NEG, sy, tf.tdf[1]);
  return fmaxf(rt_max3(s0, s1, s2), rt_max3(s3, s4, s5));
}

// Complete decision from the score, falling back to the compare form on a boundary.
RT_HD bool rt_node_accept_scored(const RtVec3& ini, const RtVec3& end, const RtVec3& e, const RtVec3& d, const RtRayConst& rc,
                                 const RtTFloat& tf, float T) {
  const float S = rt_node_score(ini, end, e, d, rc, tf);
  if (S == 0.f) return rt_node_accept_exact(ini, end, e, d, rc, rt_t_const(rc, T));
  return S > 0.f;
}


// ------------------------------------------------------------------------------------------------
// Conservative slab classification in front of the exact node test.
//
// With t_p = (plane - e_k) / d_k for the six planes, tN = max of the three near values and tF = min
// of the three far values, the box is crossed on [tN, tF].  The reference's six face tests can only
// succeed at a crossing that lies within rounding distance of tN or tF, and every one of their
// comparisons is, in t, a comparison of two of these plane distances with an error below
// 2^-22 (|a| + |b|), a and b being plane distances of the axes involved (including the far plane of an
// axis, because in-plane offsets are measured from a face corner).  So whenever every comparison that
// matters is decided by more than E = 2^-17 (max |t_p| + T) the outcome of the reference test is known
// without evaluating it:
//   +1  accepted   (the entry face when tN > 0, else the exit face, passes with margin: t in (0,T), the
//                   next near / far plane is further than E away, and the box is crossed for longer than E)
//   -1  rejected   (missed by more than E, or entirely behind by more than E, or first reachable at t > T + E)
//    0  undecided  -> rt_node_score / rt_node_accept_exact
// Directions with a zero (or denormal-small) component give infinite or NaN distances; E is then
// infinite or NaN, every comparison is false and the answer is 0.
// ------------------------------------------------------------------------------------------------
#define RT_SLAB_EPS 7.62939453125e-06f   /* 2^-17 */

// A direction component that is exactly zero.  Faces perpendicular to that axis can never pass (their solve
// divides by zero), and for the other faces the in-plane condition on that axis degenerates to
// 0 <= fl(e - lo) <= side for faces measured from the low corner and 0 <= fl(hi - e) <= side for faces measured
// from the high corner (fl(|d| b) <= side |d| <=> b <= side when side is a power of two).  Strictly inside
// for both conventions: the axis imposes nothing.  Outside for both: nothing can pass.  Otherwise: undecided.
//   returns +1 neutral, -1 rejects the node, 0 undecided
RT_HD int rt_zero_axis(float lo, float hi, float e, float side) {
  const float A = e - lo, B = hi - e;
  const float in = fminf(fminf(A, side - A), fminf(B, side - B));
  const float out = fmaxf(fminf(-A, B - side), fminf(-B, A - side));
  return in > 0.f ? 1 : (out > 0.f ? -1 : 0);
}

// lo / hi: min and max corner of the cube; inv = 1 / d per component; zx, zy, zz: that component of d is zero;
// any_zero: some lane of the wave has a zero component (wave-uniform: skips the extra work otherwise).
RT_HD int rt_node_slab_class(const RtVec3& lo, const RtVec3& hi, const RtVec3& e, const RtVec3& inv, float T, bool any_zero,
                             bool zx, bool zy, bool zz) {
  float x0 = (lo.x - e.x) * inv.x, x1 = (hi.x - e.x) * inv.x;
  float y0 = (lo.y - e.y) * inv.y, y1 = (hi.y - e.y) * inv.y;
  float z0 = (lo.z - e.z) * inv.z, z1 = (hi.z - e.z) * inv.z;
  float nx = fminf(x0, x1), fx = fmaxf(x0, x1), ny = fminf(y0, y1), fy = fmaxf(y0, y1), nz = fminf(z0, z1), fz = fmaxf(z0, z1);
  bool z_rej = false, z_und = false;
  if (any_zero) {
    const float side = hi.x - lo.x;
    const float NINF = -__builtin_inff(), PINF = __builtin_inff();
    if (zx) { const int c = rt_zero_axis(lo.x, hi.x, e.x, side); z_rej |= c < 0; z_und |= c == 0; nx = NINF; fx = PINF; x0 = 0.f; x1 = 0.f; }
    if (zy) { const int c = rt_zero_axis(lo.y, hi.y, e.y, side); z_rej |= c < 0; z_und |= c == 0; ny = NINF; fy = PINF; y0 = 0.f; y1 = 0.f; }
    if (zz) { const int c = rt_zero_axis(lo.z, hi.z, e.z, side); z_rej |= c < 0; z_und |= c == 0; nz = NINF; fz = PINF; z0 = 0.f; z1 = 0.f; }
  }
  const float tN = rt_max3(nx, ny, nz), tN2 = rt_med3(nx, ny, nz);
  const float tF = rt_min3(fx, fy, fz), tF2 = rt_med3(fx, fy, fz);
  // every in-plane comparison of the reference carries the magnitude of an offset measured from a corner of the
  // face, i.e. up to the larger of the two plane distances of that axis: the margin scales with the largest of all six
  const float M = fmaxf(rt_max3(fabsf(x0), fabsf(x1), fabsf(y0)), rt_max3(fabsf(y1), fabsf(z0), fabsf(z1)));
  const float E = RT_SLAB_EPS * (M + T);
  const float span = tF - tN;
  const bool front = tN > 0.f;
  const float tc = front ? tN : tF;                       // first reachable crossing
  const float gap = front ? tN - tN2 : tF2 - tF;          // distance to the next plane of the same kind
  const float a = rt_min3(T - tc, gap, span);
  const bool acc = (a > E) & (tc > E);
  const float r = rt_max3(-span, -tF, tc - T);
  const bool rej = (r > E) | z_rej;
  return rej ? -1 : ((acc & !z_und) ? 1 : 0);
}

// The same classification when no direction component is zero, on the interleaved (lo, hi) layout of the wave
// kernel's node records.  acc / rej equal "class > 0" / "class < 0" of rt_node_slab_class; neither = undecided.
// (tc - T = -(T - tc) exactly, and max3(-a,-b,-c) > E <=> min3(a,b,c) < -E; a NaN can only come from an infinite
// plane distance, which makes E infinite and both comparisons false.)
// The classification comes in two steps.  The head holds what a certain rejection needs; the tail adds the margin and
// the acceptance.  A caller that knows an upper bound Eb >= E for its ray (rt_slab_margin_bound) may conclude
// "rejected" from head.r < -Eb alone and skip the tail: that is the tail's own rejection test with a larger margin.
struct RtSlabHead { float tN, tF, span, tc, x, r; bool front; };
RT_HD RtSlabHead rt_node_slab_head(float nx, float fx, float ny, float fy, float nz, float fz, float T) {
  RtSlabHead h;
  h.tN = rt_max3(nx, ny, nz);
  h.tF = rt_min3(fx, fy, fz);
  h.span = h.tF - h.tN;
  h.front = h.tN > 0.f;
  h.tc = h.front ? h.tN : h.tF;
  h.x = T - h.tc;
  h.r = rt_min3(h.span, h.tF, h.x);
  return h;
}
RT_HD void rt_node_slab_tail(const RtSlabHead& h, float nx, float fx, float ny, float fy, float nz, float fz, float T, bool& acc, bool& rej) {
  const float tN2 = rt_med3(nx, ny, nz), tF2 = rt_med3(fx, fy, fz);
  const float M = fmaxf(rt_max3(fx, fy, fz), -rt_min3(nx, ny, nz));   // = max |t| over the six planes, as near <= far per axis
  const float E = RT_SLAB_EPS * (M + T);
  const float gap = h.front ? h.tN - tN2 : tF2 - h.tF;
  acc = fminf(rt_min3(h.x, gap, h.span), h.tc) > E;
  rej = h.r < -E;
}
RT_HD void rt_node_slab_sorted(float nx, float fx, float ny, float fy, float nz, float fz, float T, bool& acc, bool& rej) {
  const RtSlabHead h = rt_node_slab_head(nx, fx, ny, fy, nz, fz, T);
  rt_node_slab_tail(h, nx, fx, ny, fy, nz, fz, T, acc, rej);
}
// Upper bound of the margin E of the classification for every node inside the cube whose planes are (lo, hi) per axis
// and every T <= T0: a plane p of such a node has |fl(p - e)| <= max(|fl(lo - e)|, |fl(hi - e)|) (rounding is monotone),
// products and sums with the same partners keep the order, so M <= the largest of the three bounds and
// fl(EPS fl(M + T)) <= fl(EPS fl(bound + T0)).
RT_HD float rt_slab_margin_bound(float lox, float hix, float loy, float hiy, float loz, float hiz, const RtVec3& e, const RtVec3& inv, float T0) {
  const float mx = fmaxf(fabsf(lox - e.x), fabsf(hix - e.x)) * fabsf(inv.x);
  const float my = fmaxf(fabsf(loy - e.y), fabsf(hiy - e.y)) * fabsf(inv.y);
  const float mz = fmaxf(fabsf(loz - e.z), fabsf(hiz - e.z)) * fabsf(inv.z);
  return RT_SLAB_EPS * (rt_max3(mx, my, mz) + T0);
}
RT_HD void rt_node_slab_fast(float lox, float hix, float loy, float hiy, float loz, float hiz, const RtVec3& e, const RtVec3& inv,
                             float T, bool& acc, bool& rej) {
  const float x0 = (lox - e.x) * inv.x, x1 = (hix - e.x) * inv.x;
  const float y0 = (loy - e.y) * inv.y, y1 = (hiy - e.y) * inv.y;
  const float z0 = (loz - e.z) * inv.z, z1 = (hiz - e.z) * inv.z;
  rt_node_slab_sorted(fminf(x0, x1), fmaxf(x0, x1), fminf(y0, y1), fmaxf(y0, y1), fminf(z0, z1), fmaxf(z0, z1), T, acc, rej);
}
// The classification on plane distances formed by ONE fused operation each, t' = fl(p * inv - c) with c = fl(e * inv) held
// per ray, instead of fl(fl(p - e) * inv): half the instructions per node in the wave kernel's scan loop.  Both forms
// approximate tau = (p - e) * inv: |t - tau| <= 2^-23 |tau| for the two-step form, |t' - tau| <= 2^-24 (|c| + |t'|) for
// the fused one, so |t' - t| <= 2^-22 max(|t|, |c|).  The classification stays conservative when its margin also
// scales with C = max |c_k|: M becomes max(M, C), E = 2^-17 (max(M, C) + T), which leaves the same factor of 32 over
// the reference's own comparison error plus this evaluation error (<= 2^-20 max(M, C) in all).  Requires finite inv
// (a direction component so small that 1 / d overflows makes c infinite or NaN: such rays take the two-step form).
RT_HD float rt_fused_mul_sub(float a, float b, float c) { return __builtin_fmaf(a, b, -c); }
struct RtSlabFused { RtVec3 c; float cmax; };   // c = fl(e * inv) per axis, cmax = max |c_k|
RT_HD RtSlabFused rt_slab_fused(const RtVec3& e, const RtVec3& inv) {
  RtSlabFused f;
  f.c = {e.x * inv.x, e.y * inv.y, e.z * inv.z};
  f.cmax = rt_max3(fabsf(f.c.x), fabsf(f.c.y), fabsf(f.c.z));
  return f;
}
RT_HD void rt_node_slab_tail_fused(const RtSlabHead& h, float nx, float fx, float ny, float fy, float nz, float fz, float T, float cmax,
                                   bool& acc, bool& rej) {
  const float tN2 = rt_med3(nx, ny, nz), tF2 = rt_med3(fx, fy, fz);
  const float M = rt_max3(rt_max3(fx, fy, fz), -rt_min3(nx, ny, nz), cmax);
  const float E = RT_SLAB_EPS * (M + T);
  const float gap = h.front ? h.tN - tN2 : tF2 - h.tF;
  acc = fminf(rt_min3(h.x, gap, h.span), h.tc) > E;
  rej = h.r < -E;
}
// (near, far) planes per axis already in the order the ray meets them
RT_HD void rt_node_slab_presorted_fused(float nxp, float fxp, float nyp, float fyp, float nzp, float fzp, const RtVec3& inv,
                                        const RtSlabFused& f, float T, bool& acc, bool& rej) {
  const float nx = rt_fused_mul_sub(nxp, inv.x, f.c.x), fx = rt_fused_mul_sub(fxp, inv.x, f.c.x);
  const float ny = rt_fused_mul_sub(nyp, inv.y, f.c.y), fy = rt_fused_mul_sub(fyp, inv.y, f.c.y);
  const float nz = rt_fused_mul_sub(nzp, inv.z, f.c.z), fz = rt_fused_mul_sub(fzp, inv.z, f.c.z);
  const RtSlabHead h = rt_node_slab_head(nx, fx, ny, fy, nz, fz, T);
  rt_node_slab_tail_fused(h, nx, fx, ny, fy, nz, fz, T, f.cmax, acc, rej);
}
// Bound of that margin over every node inside the cube (lo, hi) and every T <= T0 (see rt_slab_margin_bound).
RT_HD float rt_slab_margin_bound_fused(float lox, float hix, float loy, float hiy, float loz, float hiz, const RtVec3& e, const RtVec3& inv,
                                       const RtSlabFused& f, float T0) {
  // |fl(p * inv - c)| <= the larger of the two-step bound and ... the fused distances of the enclosing cube's planes bound
  // those of every plane between them (rounding is monotone), and cmax is part of every node's margin
  const float mx = fmaxf(fabsf(rt_fused_mul_sub(lox, inv.x, f.c.x)), fabsf(rt_fused_mul_sub(hix, inv.x, f.c.x)));
  const float my = fmaxf(fabsf(rt_fused_mul_sub(loy, inv.y, f.c.y)), fabsf(rt_fused_mul_sub(hiy, inv.y, f.c.y)));
  const float mz = fmaxf(fabsf(rt_fused_mul_sub(loz, inv.z, f.c.z)), fabsf(rt_fused_mul_sub(hiz, inv.z, f.c.z)));
  (void)e;
  return RT_SLAB_EPS * (fmaxf(rt_max3(mx, my, mz), f.cmax) + T0);
}
// The same again when the caller already knows which plane of each axis is the near one (n*: the plane the ray meets
// first, i.e. the low one for a positive direction component): (plane - e) * inv is monotone in the plane, so the two
// products come out ordered and min / max are not needed.  With an infinite inv a product can be NaN where min / max
// would have dropped it; E is infinite then and both answers are "no" either way.
RT_HD void rt_node_slab_presorted(float nxp, float fxp, float nyp, float fyp, float nzp, float fzp, const RtVec3& e, const RtVec3& inv,
                                  float T, bool& acc, bool& rej) {
  rt_node_slab_sorted((nxp - e.x) * inv.x, (fxp - e.x) * inv.x, (nyp - e.y) * inv.y, (fyp - e.y) * inv.y,
                      (nzp - e.z) * inv.z, (fzp - e.z) * inv.z, T, acc, rej);
}

// The same classification for a ray whose zero direction components are known at compile time (Z* = that component
// is exactly zero; at least one is not).  Such an axis takes no part in the slab arithmetic; it only asks where the
// origin lies between the two planes (rt_zero_axis, written as comparisons: with A = e - lo and B = hi - e,
// strictly inside <=> 0 < min(A,B) and max(A,B) < side; outside for both conventions <=> (A < 0 and B > side) or
// (B < 0 and A > side)).  acc / rej equal "class > 0" / "class < 0" of rt_node_slab_class for such a ray.
// dlo / dhi: the differences fl(lo - e) and fl(hi - e) per axis (the wave kernel forms them while the record is in its
// scalar registers); A = e - lo is the exact negation of dlo.
template <bool ZX, bool ZY, bool ZZ>
RT_HD void rt_node_slab_zero_d(const float* dlo, const float* dhi, float side, const RtVec3& inv, float T, bool& acc, bool& rej) {
  static_assert(!(ZX && ZY && ZZ), "a direction has a non-zero component");
  const float iv[3] = {inv.x, inv.y, inv.z};
  const bool Z[3] = {ZX, ZY, ZZ};
  float n[2] = {0.f, 0.f}, f[2] = {0.f, 0.f};
  int K = 0;
  bool in_all = true, out_any = false;
  for (int k = 0; k < 3; k++) {
    if (Z[k]) {
      const float A = -dlo[k], B = dhi[k];
      const float mn = fminf(A, B), mx = fmaxf(A, B);
      in_all = (int)in_all & (int)(mn > 0.f) & (int)(mx < side);
      // (A < 0 and B > side) or (B < 0 and A > side): with side > 0 the negative one is the smaller, the other the larger
      out_any = (int)out_any | ((int)(mn < 0.f) & (int)(mx > side));
    } else {
      const float t0 = dlo[k] * iv[k], t1 = dhi[k] * iv[k];
      n[K] = fminf(t0, t1); f[K] = fmaxf(t0, t1); K++;
    }
  }
  const bool two = K == 2;
  const float tN = two ? fmaxf(n[0], n[1]) : n[0], tN2 = fminf(n[0], n[1]);
  const float tF = two ? fminf(f[0], f[1]) : f[0], tF2 = fmaxf(f[0], f[1]);
  const float M = two ? fmaxf(tF2, -tN2) : fmaxf(f[0], -n[0]);
  const float E = RT_SLAB_EPS * (M + T);
  const float span = tF - tN;
  const bool front = tN > 0.f;
  const float tc = front ? tN : tF;
  const float x = T - tc;
  float a = fminf(x, span);
  if (two) a = fminf(a, front ? tN - tN2 : tF2 - tF);
  rej = (rt_min3(span, tF, x) < -E) | out_any;
  acc = (fminf(a, tc) > E) & in_all & !rej;
}
// The part of that rejection which needs no slab arithmetic at all: the origin lies outside the cube's extent on a zero
// axis (for a ray parallel to an axis: outside the cube's footprint).  Implies rej of rt_node_slab_zero_d.
template <bool ZX, bool ZY, bool ZZ>
RT_HD bool rt_node_zero_outside(const float* dlo, const float* dhi, float side) {
  const bool Z[3] = {ZX, ZY, ZZ};
  bool out_any = false;
  for (int k = 0; k < 3; k++) {
    if (!Z[k]) continue;
    const float A = -dlo[k], B = dhi[k];
    out_any = (int)out_any | ((int)(fminf(A, B) < 0.f) & (int)(fmaxf(A, B) > side));
  }
  return out_any;
}
template <bool ZX, bool ZY, bool ZZ>
RT_HD void rt_node_slab_zero(float lox, float hix, float loy, float hiy, float loz, float hiz, const RtVec3& e, const RtVec3& inv,
                             float T, bool& acc, bool& rej) {
  const float dlo[3] = {lox - e.x, loy - e.y, loz - e.z}, dhi[3] = {hix - e.x, hiy - e.y, hiz - e.z};
  rt_node_slab_zero_d<ZX, ZY, ZZ>(dlo, dhi, hix - lox, inv, T, acc, rej);
}

// Same classification as two predicates (accepted / undecided), the form the wave kernel ballots.
RT_HD void rt_node_slab_class2(const RtVec3& lo, const RtVec3& hi, const RtVec3& e, const RtVec3& inv, float T, bool any_zero,
                               unsigned zmask, bool& accepted, bool& undecided) {
  const int c = rt_node_slab_class(lo, hi, e, inv, T, any_zero, (zmask & 1u) != 0u, (zmask & 2u) != 0u, (zmask & 4u) != 0u);
  accepted = c > 0; undecided = c == 0;
}

// ------------------------------------------------------------------------------------------------
// Leaf triangles a whole wave can skip.
//
// A leaf of the octree holds up to 50 triangles that merely touch its cube, and the reference tests a ray that enters
// the cube against all of them (octtree.cu:264-274); nine tests in ten are on triangles nowhere near the ray.  A test
// that the reference is certain to reject changes nothing (triangle.cu:137 only acts on acceptance), so the wave kernel
// leaves such tests out when it can prove the rejection for all of its rays at once: on entering a leaf, lane k takes
// triangle k and asks whether any ray of the wave's BUNDLE can come near it.
//
// The bundle (RtBundle) bounds the wave's ray lines: every point X = q + s d of every line (q: the point of the line nearest
// o) lies within r0 + k |s| of the axis (o, a) - o is a point all the lines pass close to (the eye, a point light) or
// just a point of one of them, a a unit vector between their directions, r0 the largest distance of a line from o, k
// the largest |d - a|.  A line that enters the ball (c, R') does so where |s||d| <= |c - o| + R', and |d| >= 1 - k, so
// with ksp = k / (1 - k) it does so within r0 + ksp (|c - o| + R') of the axis, and cannot if the centre is further from
// the axis than that plus R'.
//
// What R' has to be for "the line misses the ball" to imply "triangle.cu:107-152 rejects", in float arithmetic:
// with b = eye - v0, n = e1 x e2 and det = -d.n, the reference accepts only if its rounded numerators satisfy
// 0 <= nb / det, 0 <= ng / det and 1 - nb / det - ng / det >= 0.  The rounded numerators differ from the exact ones by at
// most E_b = c u |b||e2|, E_g = c u |b||e1|, E_d = c u |e1||e2| (u = 2^-24; c <= 6 sqrt(2) < 10, operation by operation:
// a numerator of rt_cofactor_solve is (p x q) . r with p, q, r two of e1, e2, -d and b.  Each component of the cross product is
// two rounded products and a rounded difference: off by at most 2u (|p_j q_k| + |p_k q_j|) (1 + O(u)), and the vector A of those
// sums has |A| <= sqrt(2) |p||q|.  The dot product with r is three rounded products and two rounded sums: every term carries
// at most three more roundings, b's components one of their own (eye - v0); together |computed - exact| <=
// (2u + 3u + u) sum_i |r_i| A_i (1 + O(u)) <= 6 sqrt(2) u |p||q||r|, with |d| <= 1 + 2^-22.  No operand is near the underflow
// threshold in a scene that fits the +-16 root cube unless the triangle is degenerate, and those carry Q = +inf below).  If |det| >= tau > 2 E_d the computed det has the exact one's sign, and acceptance implies for the
// exact solution beta >= -E_b / tau, gamma >= -E_g / tau, beta + gamma <= 1 + 2u + (E_b + E_g + 2 E_d) / tau: the
// point where the line meets the triangle's plane lies in the triangle scaled about its centroid by 1 + 3 mu,
// mu = c u (|b| (|e1| + |e2|) + 2 |e1||e2|) / tau, hence in the ball of radius R (1 + 3 mu) about the centroid (R: the
// largest distance of a corner).  With tau = cosn sinphi |e1||e2| (cosn: a lower bound of |d.n| / |n| over the wave's
// rays, sinphi = |n| / (|e1||e2|)) that is R' = R + Q (|b| + 2R) / cosn with Q = 3 R c u (1/|e1| + 1/|e2|) / sinphi
// stored per triangle (c = 64 there, for slack; the last term uses 2|e1||e2| / (|e1| + |e2|) <= 2R).
// Rays nearly in the triangle's plane (small |det|) are where the reference's quotients are noise and may accept
// anything: the test therefore never skips a triangle unless cosn > 2^-10 (for slivers, sinphi < 1/64: unless
// cosn sinphi > 2^-16, through a shortened stored normal); degenerate triangles carry Q = +inf, which never passes.
struct RtCullRec { float cx, cy, cz, R, nx, ny, nz, Q; };   // 32 B per leaf reference, in leaf order behind the triangle records
struct RtBundle { float ox, oy, oz, ax, ay, az, r0, ksp, k1, bw; };   // k1 = 1 + ksp; bw >= |eye - o| for every ray
#define RT_CULL_GUARD 0x1p-10f
#define RT_CULL_P2_SLACK (-0x1p-19f)

#if !defined(__HIPCC__) || !defined(__HIP_DEVICE_COMPILE__)
// Host: the record of a triangle (double arithmetic, rounded away from the safe side with room to spare).
inline RtCullRec rt_cull_record(const float* v0, const float* e1, const float* e2) {
  RtCullRec r;
  double c[3], n[3], R = 0, l1 = 0, l2 = 0, nl = 0;
  for (int k = 0; k < 3; k++) { c[k] = (double)v0[k] + ((double)e1[k] + (double)e2[k]) / 3.0; l1 += (double)e1[k] * e1[k]; l2 += (double)e2[k] * e2[k]; }
  n[0] = (double)e1[1] * e2[2] - (double)e1[2] * e2[1]; n[1] = (double)e1[2] * e2[0] - (double)e1[0] * e2[2]; n[2] = (double)e1[0] * e2[1] - (double)e1[1] * e2[0];
  for (int k = 0; k < 3; k++) nl += n[k] * n[k];
  l1 = sqrt(l1); l2 = sqrt(l2); nl = sqrt(nl);
  for (int q = 0; q < 3; q++) {
    double s = 0;
    for (int k = 0; k < 3; k++) { const double p = (double)v0[k] + (q == 1 ? (double)e1[k] : q == 2 ? (double)e2[k] : 0.0) - c[k]; s += p * p; }
    if (sqrt(s) > R) R = sqrt(s);
  }
  r.cx = (float)c[0]; r.cy = (float)c[1]; r.cz = (float)c[2];
  double shift = 0;   // the centre moved by its rounding to float: the radius grows by as much
  for (int k = 0; k < 3; k++) { const double dk = (double)(k == 0 ? r.cx : k == 1 ? r.cy : r.cz) - c[k]; shift += dk * dk; }
  r.R = (float)((R + sqrt(shift)) * 1.001 + 1e-6);
  const float inf = __builtin_inff();
  r.nx = r.ny = r.nz = 0.f; r.Q = inf;
  if (nl > 0 && l1 > 0 && l2 > 0 && std::isfinite(nl) && std::isfinite(R)) {
    const double sinphi = nl / (l1 * l2);
    // (a sliver's normal is stored shortened, by 64 sinphi: the test's cosn then is a lower bound of 64 sinphi |d.n| / |n|, so that
    //  passing its guard still means |det| >= 2^-16 |e1||e2|, far above what rounding can do to det)
    const double shorten = sinphi < 1.0 / 64 ? 64 * sinphi : 1.0;
    r.nx = (float)(n[0] / nl * shorten); r.ny = (float)(n[1] / nl * shorten); r.nz = (float)(n[2] / nl * shorten);
    const double Q = 3.0 * R * 64.0 * 0x1p-24 * (1.0 / l1 + 1.0 / l2) / sinphi * 1.01;
    if (std::isfinite(Q) && sinphi > 0x1p-40) r.Q = (float)(Q * 1.001);
  }
  return r;
}
#endif

// A ray's share of the bundle around axis (o, a): its line's distance from o (for a unit direction; rt_bundle_finish allows for
// the rest), its origin's distance from o, |d - a|.
// (The distance comes from the cross product: a difference of squares would lose it for a far-away o.)
RT_HD void rt_bundle_lane(const RtVec3& o, const RtVec3& a, const RtVec3& e, const RtVec3& d, float& r0, float& bw, float& k) {
  const RtVec3 w = {o.x - e.x, o.y - e.y, o.z - e.z};
  const RtVec3 c = {w.y * d.z - w.z * d.y, w.z * d.x - w.x * d.z, w.x * d.y - w.y * d.x};
  const RtVec3 q = {d.x - a.x, d.y - a.y, d.z - a.z};
  r0 = sqrtf(rt_dot(c, c)); bw = sqrtf(rt_dot(w, w)); k = sqrtf(rt_dot(q, q));
}
// The bundle from the largest of those over the wave's rays, rounded up with room for the rounding of everything above
// and of the lane test (directions and axis are unit vectors to within 2^-22).
RT_HD RtBundle rt_bundle_finish(const RtVec3& o, const RtVec3& a, float r0, float bw, float k) {
  RtBundle B;
  B.ox = o.x; B.oy = o.y; B.oz = o.z; B.ax = a.x; B.ay = a.y; B.az = a.z;
  // (a direction need not be a unit vector: |d| >= 1 - k, so a line reaches the ball (c, R') within (|c - o| + R') / (1 - k) of
  //  its point nearest o, and its distance from o is at most |(o - eye) x d| / (1 - k); that factor goes into spread and r0)
  const float ks = fmaf(k, 1.001f, 0x1p-18f);
  const float grow = ks < 0.9f ? 1.000001f / (1.f - ks) : __builtin_inff();
  B.r0 = fmaf(r0, 1.001f, fmaf(bw, 0x1p-18f, 1e-6f)) * grow;
  B.ksp = ks * grow;
  B.k1 = (1.f + B.ksp) * 1.000001f;
  B.bw = fmaf(bw, 1.001f, 1e-4f);
  return B;
}

// Lane k's question about triangle k, as the assembly asks it (fused multiply-adds are fine here: this is not the
// reference's arithmetic, only a bound).  true: no ray of the bundle can be accepted by this triangle.
RT_HD bool rt_cull_test(const RtCullRec& t, const RtBundle& B) {
  const float vx = t.cx - B.ox, vy = t.cy - B.oy, vz = t.cz - B.oz;
  const float vv = fmaf(vz, vz, fmaf(vy, vy, vx * vx));
  const float tc = fmaf(B.az, vz, fmaf(B.ay, vy, B.ax * vx));
  float p2 = fmaf(-tc, tc, vv);
  p2 = fmaf(RT_CULL_P2_SLACK, vv, p2);   // what rounding can have added to vv - tc^2
  const float p = sqrtf(fmaxf(p2, 0.f)), vl = sqrtf(vv);
  const float cosn = fabsf(fmaf(B.az, t.nz, fmaf(B.ay, t.ny, B.ax * t.nx))) - B.ksp;
  const float reach = fmaf(B.ksp, vl, fmaf(t.R, B.k1, B.r0));          // r0 + R k1 + ksp |v|
  const float lhs = (p - reach) * cosn;
  const float rhs = t.Q * fmaf(4.f, t.R, B.bw + vl) * B.k1;            // margin (1 + ksp) cosn, with |b| <= bw + |v| + R
#ifdef RT_CULL_TEST_WITHOUT_MARGIN   // (only ever defined by tests/test_host_cull.py: the harness has to notice when margin and guard are missing)
  return p - reach > 0.f;
#else
  return lhs > rhs && cosn > RT_CULL_GUARD;
#endif
}

#endif
