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

#endif
