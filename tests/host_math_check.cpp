// host_math_check.cpp -- CPU bit-comparison of the division-free predicates in rt_math.h against the
// literal restatement of the reference arithmetic (same header, compiled as plain C++).
//
//   host_math_check scene <scene.json> <resources.json> <base_dir> <W> <H> <stride>
//       replays the reference traversal for every stride-th pixel (primary ray + shadow rays towards
//       a point above the scene and straight up) and checks EVERY node test and triangle test.
//   host_math_check random <count> <seed>
//       adversarial random cubes / rays: zero direction components, eye on face planes, running t set
//       exactly to (and one ulp around) a face's crossing distance, grazing edges.
// Prints "checked <n_node> <n_tri> mismatches <m>" and exits non-zero when m > 0.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../include/rt_amd.h"
#include "../raytracer_amd/csrc/rt_math.h"

static uint64_t n_node = 0, n_tri = 0, n_bad = 0, n_cand = 0, n_acc = 0, n_zero = 0, n_amb = 0, n_zero_axis = 0, n_quick = 0, n_fused = 0, n_fused_und = 0;

static bool check_node(const RtVec3& ini, const RtVec3& end, const RtVec3& e, const RtVec3& d, float T) {
  const bool lit = rt_node_accept_literal(ini, end, e, d, T);
  const RtRayConst rc = rt_ray_const(d);
  const RtTConst tc = rt_t_const(rc, T);
  bool ex = rt_node_accept_exact(ini, end, e, d, rc, tc);
  const RtTFloat tf = rt_t_float(rc, T);
  const float S = rt_node_score(ini, end, e, d, rc, tf);
  if (S == 0.f) n_zero++;
  if (rt_node_accept_scored(ini, end, e, d, rc, tf, T) != ex) ex = !lit;   // a score mismatch is reported like any other
  {   // the slab classification may abstain, never contradict
    const RtVec3 lo = {ini.x, ini.y, end.z}, hi = {end.x, end.y, ini.z};
    const RtVec3 inv = {1.f / d.x, 1.f / d.y, 1.f / d.z};
    const bool zx = d.x == 0.f, zy = d.y == 0.f, zz = d.z == 0.f;
    const int cls = rt_node_slab_class(lo, hi, e, inv, T, zx | zy | zz, zx, zy, zz);
    if (cls == 0) n_amb++;
    if ((zx | zy | zz) && !(zx & zy & zz)) {   // the forms specialised on the zero pattern
      bool za = false, zr = false;
      const int zp = (zx ? 1 : 0) | (zy ? 2 : 0) | (zz ? 4 : 0);
      switch (zp) {
        case 1: rt_node_slab_zero<true, false, false>(lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, e, inv, T, za, zr); break;
        case 2: rt_node_slab_zero<false, true, false>(lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, e, inv, T, za, zr); break;
        case 3: rt_node_slab_zero<true, true, false>(lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, e, inv, T, za, zr); break;
        case 4: rt_node_slab_zero<false, false, true>(lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, e, inv, T, za, zr); break;
        case 5: rt_node_slab_zero<true, false, true>(lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, e, inv, T, za, zr); break;
        case 6: rt_node_slab_zero<false, true, true>(lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, e, inv, T, za, zr); break;
      }
      n_zero_axis++;
      {   // the footprint part of the rejection on its own may only fire where the whole form rejects
        const float dlo[3] = {lo.x - e.x, lo.y - e.y, lo.z - e.z}, dhi[3] = {hi.x - e.x, hi.y - e.y, hi.z - e.z};
        const float side = hi.x - lo.x;
        bool o = false;
        switch (zp) {
          case 1: o = rt_node_zero_outside<true, false, false>(dlo, dhi, side); break;
          case 2: o = rt_node_zero_outside<false, true, false>(dlo, dhi, side); break;
          case 3: o = rt_node_zero_outside<true, true, false>(dlo, dhi, side); break;
          case 4: o = rt_node_zero_outside<false, false, true>(dlo, dhi, side); break;
          case 5: o = rt_node_zero_outside<true, false, true>(dlo, dhi, side); break;
          case 6: o = rt_node_zero_outside<false, true, true>(dlo, dhi, side); break;
        }
        if (o && !zr) { if (n_bad < 10) printf("ZERO-AXIS OUTSIDE WITHOUT REJECTION pattern %d\n", zp); n_bad++; }
      }
      if (za != (cls > 0) || zr != (cls < 0)) {
        if (n_bad < 10) printf("SLAB ZERO-PATTERN/GENERAL DIFFER pattern %d cls %d special %d %d\n", zp, cls, (int)za, (int)zr);
        n_bad++;
      }
    }
    if (!(zx | zy | zz)) {
      bool fa, fr;
      rt_node_slab_fast(lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, e, inv, T, fa, fr);
      if (fa != (cls > 0) || fr != (cls < 0)) {
        if (n_bad < 10) printf("SLAB FAST/GENERAL DIFFER cls %d fast %d %d\n", cls, (int)fa, (int)fr);
        n_bad++;
      }
      bool pa, pr;   // planes handed over in near / far order by the direction signs
      rt_node_slab_presorted(d.x < 0 ? hi.x : lo.x, d.x < 0 ? lo.x : hi.x, d.y < 0 ? hi.y : lo.y, d.y < 0 ? lo.y : hi.y,
                             d.z < 0 ? hi.z : lo.z, d.z < 0 ? lo.z : hi.z, e, inv, T, pa, pr);
      if (pa != fa || pr != fr) {
        if (n_bad < 10) printf("SLAB PRESORTED/FAST DIFFER %d %d vs %d %d\n", (int)pa, (int)pr, (int)fa, (int)fr);
        n_bad++;
      }
      // the early rejection against a margin bound taken from an enclosing cube may only fire where the full form rejects
      const float nx = ((d.x < 0 ? hi.x : lo.x) - e.x) * inv.x, fx = ((d.x < 0 ? lo.x : hi.x) - e.x) * inv.x;
      const float ny = ((d.y < 0 ? hi.y : lo.y) - e.y) * inv.y, fy = ((d.y < 0 ? lo.y : hi.y) - e.y) * inv.y;
      const float nz = ((d.z < 0 ? hi.z : lo.z) - e.z) * inv.z, fz = ((d.z < 0 ? lo.z : hi.z) - e.z) * inv.z;
      const RtSlabHead hd = rt_node_slab_head(nx, fx, ny, fy, nz, fz, T);
      const bool in_root = lo.x >= -16.f && hi.x <= 16.f && lo.y >= -16.f && hi.y <= 16.f && lo.z >= -16.f && hi.z <= 16.f;
      const float b_self = rt_slab_margin_bound(lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, e, inv, 64.f);
      const float b_root = rt_slab_margin_bound(-16.f, 16.f, -16.f, 16.f, -16.f, 16.f, e, inv, 64.f);
      const bool q_self = T <= 64.f && hd.r < -b_self, q_root = T <= 64.f && in_root && hd.r < -b_root;
      n_quick += q_root;
      if ((q_self || q_root) && !pr) {
        if (n_bad < 10) printf("SLAB EARLY REJECTION WITHOUT REJECTION self %d root %d\n", (int)q_self, (int)q_root);
        n_bad++;
      }
      if (in_root && b_root < b_self) { if (n_bad < 10) printf("ROOT MARGIN BELOW NODE MARGIN\n"); n_bad++; }
      // the form on fused plane distances (one multiply-add per plane): may abstain, never contradict; its early rejection
      // against the root's bound may only fire where its full form rejects
      if (std::isfinite(inv.x) && std::isfinite(inv.y) && std::isfinite(inv.z)) {
        const RtSlabFused fu = rt_slab_fused(e, inv);
        bool ua, ur;
        rt_node_slab_presorted_fused(d.x < 0 ? hi.x : lo.x, d.x < 0 ? lo.x : hi.x, d.y < 0 ? hi.y : lo.y, d.y < 0 ? lo.y : hi.y,
                                     d.z < 0 ? hi.z : lo.z, d.z < 0 ? lo.z : hi.z, inv, fu, T, ua, ur);
        n_fused++; n_fused_und += !ua && !ur;
        if ((ua && !lit) || (ur && lit)) {
          if (n_bad < 10) printf("FUSED SLAB CONTRADICTION acc %d rej %d lit %d e %.9g %.9g %.9g d %.9g %.9g %.9g T %.9g\n", (int)ua, (int)ur, (int)lit, e.x, e.y, e.z, d.x, d.y, d.z, T);
          n_bad++;
        }
        const float un = rt_fused_mul_sub(d.x < 0 ? hi.x : lo.x, inv.x, fu.c.x), uf = rt_fused_mul_sub(d.x < 0 ? lo.x : hi.x, inv.x, fu.c.x);
        const float vn = rt_fused_mul_sub(d.y < 0 ? hi.y : lo.y, inv.y, fu.c.y), vf = rt_fused_mul_sub(d.y < 0 ? lo.y : hi.y, inv.y, fu.c.y);
        const float wn = rt_fused_mul_sub(d.z < 0 ? hi.z : lo.z, inv.z, fu.c.z), wf = rt_fused_mul_sub(d.z < 0 ? lo.z : hi.z, inv.z, fu.c.z);
        const RtSlabHead hf = rt_node_slab_head(un, uf, vn, vf, wn, wf, T);
        const float fb_root = rt_slab_margin_bound_fused(-16.f, 16.f, -16.f, 16.f, -16.f, 16.f, e, inv, fu, 64.f);
        if (T <= 64.f && in_root && hf.r < -fb_root && !ur) { if (n_bad < 10) printf("FUSED EARLY REJECTION WITHOUT REJECTION\n"); n_bad++; }
      }
    }
    if ((cls > 0 && !lit) || (cls < 0 && lit)) {
      if (n_bad < 10) printf("SLAB CONTRADICTION cls %d lit %d ini %.9g %.9g %.9g end %.9g %.9g %.9g e %.9g %.9g %.9g d %.9g %.9g %.9g T %.9g\n", cls, lit,
             ini.x, ini.y, ini.z, end.x, end.y, end.z, e.x, e.y, e.z, d.x, d.y, d.z, T);
      n_bad++;
    }
  }
  n_node++; n_acc += lit;
  if (lit != ex) {
    if (n_bad < 10)
      printf("NODE MISMATCH lit %d exact %d ini %.9g %.9g %.9g end %.9g %.9g %.9g e %.9g %.9g %.9g d %.9g %.9g %.9g T %.9g\n", lit, ex,
             ini.x, ini.y, ini.z, end.x, end.y, end.z, e.x, e.y, e.z, d.x, d.y, d.z, T);
    n_bad++;
  }
  return lit;
}

static bool check_tri(const RtVec3& v0, const RtVec3& e1, const RtVec3& e2, const RtVec3& e, const RtVec3& d, float T, float& t,
                      float& bet, float& gam) {
  const bool lit = rt_triangle_test(v0, e1, e2, e, d, T, t, bet, gam);
  const RtTriNum n = rt_triangle_numerators(v0, e1, e2, e, d);
  float t2 = 0, b2 = 0, g2 = 0;
  bool ex = false;
  if (rt_triangle_candidate(n)) { n_cand++; ex = rt_triangle_finish(n, T, t2, b2, g2); }
  n_tri++;
  if (lit != ex || (lit && (rt_bits(t) != rt_bits(t2) || rt_bits(bet) != rt_bits(b2) || rt_bits(gam) != rt_bits(g2)))) {
    if (n_bad < 10) printf("TRI MISMATCH lit %d exact %d\n", lit, ex);
    n_bad++;
  }
  return lit;
}

struct Tree {
  std::vector<int32_t> kind, refs, ref_first; std::vector<float> bounds, verts; std::vector<int32_t> child[8];
};
static void link(Tree& T, int& next, int& ro) {
  int me = next++;
  if (T.kind[me] >= 0) { T.ref_first[me] = ro; ro += T.kind[me]; return; }
  for (int c = 0; c < 8; c++) { T.child[c][me] = next; link(T, next, ro); }
}

struct Hit { float t; int prim; float bet, gam; };
static void trace(const Tree& T, const RtVec3& e, const RtVec3& d, Hit& h) {
  int node[24], cur[24], sp = 1; node[0] = 0; cur[0] = 0;
  while (sp) {
    int parent = node[sp - 1], c = cur[sp - 1];
    if (c == 7) sp--; else cur[sp - 1] = c + 1;
    int ch = T.child[c][parent];
    const float* b = &T.bounds[6 * ch];
    if (!check_node({b[0], b[1], b[2]}, {b[3], b[4], b[5]}, e, d, h.t)) continue;
    if (T.kind[ch] < 0) { node[sp] = ch; cur[sp] = 0; sp++; continue; }
    for (int k = 0; k < T.kind[ch]; k++) {
      int id = T.refs[T.ref_first[ch] + k];
      const float* v = &T.verts[9 * id];
      float t, bb, gg;
      if (check_tri({v[0], v[1], v[2]}, {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, {v[6] - v[0], v[7] - v[1], v[8] - v[2]}, e, d, h.t, t, bb, gg)) {
        h.t = t; h.prim = id; h.bet = bb; h.gam = gg;
      }
    }
  }
}

static int run_scene(int ac, char** av) {
  if (ac < 8) return 2;
  int W = atoi(av[5]), H = atoi(av[6]), stride = atoi(av[7]);
  rt_scene* sc = nullptr;
  if (rt_scene_load(av[2], av[3], av[4], W, H, 3, &sc)) { printf("%s\n", rt_last_error()); return 2; }
  rt_scene_info I; rt_scene_get_info(sc, &I);
  if (!I.n_nodes) { printf("checked 0 0 mismatches 0\n"); return 0; }
  Tree T; T.kind.resize(I.n_nodes); T.bounds.resize(6 * (size_t)I.n_nodes); T.refs.resize(I.n_leaf_refs + 1);
  T.verts.resize(9 * (size_t)I.n_triangles); std::vector<float> nrm(9 * (size_t)I.n_triangles); std::vector<int32_t> mesh(I.n_triangles);
  rt_scene_dump_tree(sc, T.kind.data(), T.bounds.data(), T.refs.data());
  rt_scene_dump_triangles(sc, T.verts.data(), nrm.data(), mesh.data());
  float cam[19]; rt_scene_dump_camera(sc, cam);
  T.ref_first.assign(I.n_nodes, 0); for (int c = 0; c < 8; c++) T.child[c].assign(I.n_nodes, -1);
  int next = 0, ro = 0; link(T, next, ro);
  for (int row = 0; row < H; row += stride) for (int i = (row / stride) % stride; i < W; i += stride) {
    int j = H - 1 - row;
    float ic = (float)((double)cam[12] + (double)((cam[13] - cam[12]) / (float)W) * (i + 0.5));
    float jc = (float)((double)cam[14] + (double)((cam[15] - cam[14]) / (float)H) * (j + 0.5));
    RtVec3 d = rt_normalized({ic * cam[0] + jc * cam[3] + -0.1f * cam[6], ic * cam[1] + jc * cam[4] + -0.1f * cam[7], ic * cam[2] + jc * cam[5] + -0.1f * cam[8]});
    RtVec3 e = {cam[9], cam[10], cam[11]};
    Hit h = {64.f, -1, 0, 0};
    trace(T, e, d, h);
    if (h.prim < 0) continue;
    const float* v = &T.verts[9 * h.prim];
    float alp = 1 - h.bet - h.gam;
    RtVec3 P = {(v[0] * alp + v[3] * h.bet) + v[6] * h.gam, (v[1] * alp + v[4] * h.bet) + v[7] * h.gam, (v[2] * alp + v[5] * h.bet) + v[8] * h.gam};
    const float* n0 = &nrm[9 * h.prim];
    RtVec3 N = rt_normalized({(n0[0] * alp + n0[3] * h.bet) + n0[6] * h.gam, (n0[1] * alp + n0[4] * h.bet) + n0[7] * h.gam, (n0[2] * alp + n0[5] * h.bet) + n0[8] * h.gam});
    RtVec3 ls[3];
    ls[0] = rt_normalized({0.f - P.x, 10.f - P.y, 20.f - P.z});                       // towards a point light
    ls[1] = {0.f * -1, -1.f * -1, 0.f * -1};                                          // axis-aligned (directional light from above)
    { float s2 = 2 * rt_dot(d, N); ls[2] = rt_normalized({d.x - N.x * s2, d.y - N.y * s2, d.z - N.z * s2}); }   // mirror bounce
    for (int k = 0; k < 3; k++) {
      Hit sh = {64.f, -1, 0, 0};
      trace(T, {P.x + ls[k].x * 0.001f, P.y + ls[k].y * 0.001f, P.z + ls[k].z * 0.001f}, ls[k], sh);
    }
  }
  printf("checked %llu %llu mismatches %llu candidates %llu score_zero %llu slab_undecided %llu zero_axis_tests %llu early_rejections %llu fused %llu fused_undecided %llu\n", (unsigned long long)n_node, (unsigned long long)n_tri, (unsigned long long)n_bad, (unsigned long long)n_cand, (unsigned long long)n_zero, (unsigned long long)n_amb, (unsigned long long)n_zero_axis, (unsigned long long)n_quick, (unsigned long long)n_fused, (unsigned long long)n_fused_und);
  return n_bad ? 1 : 0;
}

static int run_random(int ac, char** av) {
  uint64_t count = ac > 2 ? strtoull(av[2], nullptr, 10) : 1000000; unsigned seed = ac > 3 ? atoi(av[3]) : 1;
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  auto pick = [&](int n) { return (int)(rng() % n); };
  for (uint64_t it = 0; it < count; it++) {
    // dyadic cube inside the root box, orientation ini = (lo, lo, hi), end = (hi, hi, lo)
    int depth = 1 + pick(12);
    float side = 32.f / (float)(1 << depth);
    int cells = 1 << depth;
    float lo[3]; for (int k = 0; k < 3; k++) lo[k] = -16.f + side * (float)pick(cells);
    RtVec3 ini = {lo[0], lo[1], lo[2] + side}, end = {lo[0] + side, lo[1] + side, lo[2]};
    RtVec3 e, d;
    int mode = pick(8);
    e = {20.f * U(rng), 20.f * U(rng), 20.f * U(rng)};
    if (mode == 1) { e.x = lo[0] + side * (float)pick(2); }                      // eye on an x plane
    if (mode == 2) { e.y = lo[1] + side * (float)pick(2); e.z = lo[2] + side * (float)pick(2); }
    if (mode == 3) { e = {lo[0] + side * 0.5f * (1 + U(rng)), lo[1] + side * 0.5f * (1 + U(rng)), lo[2] + side * 0.5f * (1 + U(rng))}; }   // inside
    // aim at a point of the cube (often a corner / edge / face centre) so that hits and grazes are common
    float tx = lo[0] + side * (pick(3) * 0.5f), ty = lo[1] + side * (pick(3) * 0.5f), tz = lo[2] + side * (pick(3) * 0.5f);
    if (pick(3) == 0) { tx += side * 0.7f * U(rng); ty += side * 0.7f * U(rng); tz += side * 0.7f * U(rng); }
    d = {tx - e.x, ty - e.y, tz - e.z};
    if (mode == 4) d.x = 0.f; if (mode == 5) { d.x = 0.f; d.z = -0.f; } if (mode == 6) { d.y = 0.f; }
    if (d.x == 0 && d.y == 0 && d.z == 0) d.y = 1.f;
    d = rt_normalized(d);
    if (mode == 7) { d = {U(rng), U(rng), U(rng)}; d = rt_normalized(d); }
    if (pick(4) == 0) { float tiny = ldexpf(U(rng), -pick(40)); int k = pick(3); (k == 0 ? d.x : k == 1 ? d.y : d.z) = tiny; d = rt_normalized(d); }   // one nearly-zero component
    // running closest t: far, random, or exactly the crossing distance of one of the six faces (+- 1 ulp)
    float T = 64.f;
    int tm = pick(4);
    if (tm == 1) T = 40.f * fabsf(U(rng)) + 1e-3f;
    if (tm >= 2) {
      int k = pick(3); float plane = (&lo[0])[k] + side * (float)pick(2);
      float ek = k == 0 ? e.x : k == 1 ? e.y : e.z, dk = k == 0 ? d.x : k == 1 ? d.y : d.z;
      float t = (ek - plane) / (-dk);
      if (t > 0 && t < 1e6f) { T = t; int u = pick(5) - 2; T = rt_from_bits(rt_bits(T) + (unsigned)u); if (!(T > 0)) T = t; }
    }
    check_node(ini, end, e, d, T);
  }
  printf("checked %llu %llu mismatches %llu accepted %llu score_zero %llu slab_undecided %llu zero_axis_tests %llu early_rejections %llu fused %llu fused_undecided %llu\n", (unsigned long long)n_node, (unsigned long long)n_tri, (unsigned long long)n_bad, (unsigned long long)n_acc, (unsigned long long)n_zero, (unsigned long long)n_amb, (unsigned long long)n_zero_axis, (unsigned long long)n_quick, (unsigned long long)n_fused, (unsigned long long)n_fused_und);
  return n_bad ? 1 : 0;
}

int main(int ac, char** av) {
  if (ac > 1 && !strcmp(av[1], "scene")) return run_scene(ac, av);
  if (ac > 1 && !strcmp(av[1], "random")) return run_random(ac, av);
  return 2;
}
