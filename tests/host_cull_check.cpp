// host_cull_check.cpp -- CPU check of the wave-level triangle skip of rt_math.h (rt_cull_record, rt_bundle_*, rt_cull_test):
// whenever the test says "no ray of this bundle can be accepted by this triangle", the literal restatement of
// triangle.cu:107-152 must reject every ray of the bundle, whatever the running closest hit.
//
//   host_cull_check scene <scene.json> <resources.json> <base_dir> <W> <H> <stride>
//       8x8 pixel tiles (every stride-th): the primary rays, and the shadow rays of the hit points towards a point
//       above the scene and straight up, as bundles; every triangle of every leaf some ray of the bundle enters.
//   host_cull_check random <count> <seed>
//       adversarial pairs: single rays and narrow bundles that miss a random triangle (slivers, specks, large ones) by
//       10^-7 .. 10^-1, rays in the triangle's plane and parallel to its edges, origins inside the plane, directions that are
//       not quite unit vectors.
// Prints "checked <pairs> skipped <n> (<pct>%) accepted <a> mismatches <m>" (scene mode: also the leaf visits left without a
// triangle) and exits non-zero when m > 0.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../include/rt_amd.h"
#include "../raytracer_amd/csrc/rt_math.h"

static uint64_t n_pairs = 0, n_skip = 0, n_bad = 0, n_accept = 0, n_leaves = 0, n_leaves_emptied = 0;

struct Ray { RtVec3 e, d; };

static RtBundle bundle_of(const std::vector<Ray>& rays, const RtVec3& o) {
  // (the kernel's axis: between the first and the last ray)
  const RtVec3 s = {rays.front().d.x + rays.back().d.x, rays.front().d.y + rays.back().d.y, rays.front().d.z + rays.back().d.z};
  const RtVec3 a = rt_normalized(s);
  float r0 = 0, bw = 0, k = 0;
  for (const Ray& r : rays) { float x, y, z; rt_bundle_lane(o, a, r.e, r.d, x, y, z); r0 = fmaxf(r0, x); bw = fmaxf(bw, y); k = fmaxf(k, z); }
  return rt_bundle_finish(o, a, r0, bw, k);
}

// One triangle against one bundle; returns whether it was skipped.
static bool check(const float* v0, const float* e1, const float* e2, const std::vector<Ray>& rays, const RtBundle& B) {
  const RtCullRec rec = rt_cull_record(v0, e1, e2);
  const bool skip = rt_cull_test(rec, B);
  n_pairs++;
  const RtVec3 V0 = {v0[0], v0[1], v0[2]}, E1 = {e1[0], e1[1], e1[2]}, E2 = {e2[0], e2[1], e2[2]};
  bool any = false;
  for (const Ray& r : rays) { float t, b, g; if (rt_triangle_test(V0, E1, E2, r.e, r.d, __builtin_inff(), t, b, g)) any = true; }
  if (any) n_accept++;
  if (skip) {
    n_skip++;
    if (any) {
      if (n_bad < 10) fprintf(stderr, "skipped a triangle that the reference accepts: v0 %.9g %.9g %.9g e1 %.9g %.9g %.9g e2 %.9g %.9g %.9g\n",
                              v0[0], v0[1], v0[2], e1[0], e1[1], e1[2], e2[0], e2[1], e2[2]);
      n_bad++;
    }
  }
  return skip;
}

struct Tree {
  std::vector<int32_t> kind, refs, ref_first; std::vector<float> bounds, verts; std::vector<int32_t> child[8];
};
static void link(Tree& T, int& next, int& ro) {
  const int me = next++;
  if (T.kind[me] >= 0) { T.ref_first[me] = ro; ro += T.kind[me]; return; }
  for (int c = 0; c < 8; c++) { T.child[c][me] = next; link(T, next, ro); }
}
struct Hit { float t; int prim; float bet, gam; };

// The union walk of a bundle: every leaf that some ray enters (reference order, each ray with its own closest hit).
static void walk(const Tree& T, int node, const std::vector<Ray>& rays, std::vector<Hit>& hits, const std::vector<char>& mask, const RtBundle& B) {
  for (int c = 0; c < 8; c++) {
    const int ch = T.child[c][node];
    const float* b = &T.bounds[6 * ch];
    std::vector<char> m(rays.size(), 0); int cnt = 0;
    for (size_t i = 0; i < rays.size(); i++)
      if (mask[i] && rt_node_accept_literal({b[0], b[1], b[2]}, {b[3], b[4], b[5]}, rays[i].e, rays[i].d, hits[i].t)) { m[i] = 1; cnt++; }
    if (!cnt) continue;
    if (T.kind[ch] < 0) { walk(T, ch, rays, hits, m, B); continue; }
    int left = 0;
    for (int k = 0; k < T.kind[ch]; k++) {
      const int id = T.refs[T.ref_first[ch] + k];
      const float* v = &T.verts[9 * id];
      const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
      if (!check(v, e1, e2, rays, B)) left++;
      for (size_t i = 0; i < rays.size(); i++) if (m[i]) {
        float t, bb, gg;
        if (rt_triangle_test({v[0], v[1], v[2]}, {e1[0], e1[1], e1[2]}, {e2[0], e2[1], e2[2]}, rays[i].e, rays[i].d, hits[i].t, t, bb, gg)) hits[i] = {t, id, bb, gg};
      }
    }
    n_leaves++; if (!left) n_leaves_emptied++;
  }
}

static int run_scene(int ac, char** av) {
  if (ac < 8) return 2;
  const int W = atoi(av[5]), H = atoi(av[6]), stride = atoi(av[7]);
  rt_scene* sc = nullptr;
  if (rt_scene_load(av[2], av[3], av[4], W, H, 3, &sc)) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
  rt_scene_info I; rt_scene_get_info(sc, &I);
  Tree T; T.kind.resize(I.n_nodes); T.bounds.resize(6 * (size_t)I.n_nodes); T.refs.resize(I.n_leaf_refs + 1); T.verts.resize(9 * (size_t)I.n_triangles);
  std::vector<float> nrm(9 * (size_t)I.n_triangles); std::vector<int32_t> mesh(I.n_triangles);
  rt_scene_dump_tree(sc, T.kind.data(), T.bounds.data(), T.refs.data());
  rt_scene_dump_triangles(sc, T.verts.data(), nrm.data(), mesh.data());
  float cam[19]; rt_scene_dump_camera(sc, cam);
  T.ref_first.assign(I.n_nodes, 0); for (int c = 0; c < 8; c++) T.child[c].assign(I.n_nodes, -1);
  int next = 0, ro = 0; link(T, next, ro);
  const RtVec3 eye = {cam[9], cam[10], cam[11]}, light = {0.f, 10.f, 20.f};
  for (int ty = 0; ty < H / 8; ty += stride) for (int tx = (ty / stride) % stride; tx < W / 8; tx += stride) {
    std::vector<Ray> rays;
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
      const int i = tx * 8 + x, j = H - 1 - (ty * 8 + y);
      const float ic = (float)((double)cam[12] + (double)((cam[13] - cam[12]) / (float)W) * (i + 0.5));
      const float jc = (float)((double)cam[14] + (double)((cam[15] - cam[14]) / (float)H) * (j + 0.5));
      const RtVec3 d = {ic * cam[0] + jc * cam[3] + -0.1f * cam[6], ic * cam[1] + jc * cam[4] + -0.1f * cam[7], ic * cam[2] + jc * cam[5] + -0.1f * cam[8]};
      rays.push_back({eye, rt_normalized(d)});
    }
    std::vector<Hit> hits(rays.size(), Hit{64.f, -1, 0, 0});
    walk(T, 0, rays, hits, std::vector<char>(rays.size(), 1), bundle_of(rays, eye));
    for (int li = 0; li < 2; li++) {
      std::vector<Ray> sh;
      for (size_t i = 0; i < rays.size(); i++) if (hits[i].prim >= 0) {
        const float* v = &T.verts[9 * hits[i].prim];
        const float alp = 1 - hits[i].bet - hits[i].gam, bet = hits[i].bet, gam = hits[i].gam;
        const RtVec3 P = {(v[0] * alp + v[3] * bet) + v[6] * gam, (v[1] * alp + v[4] * bet) + v[7] * gam, (v[2] * alp + v[5] * bet) + v[8] * gam};
        const RtVec3 l = li == 0 ? rt_normalized({light.x - P.x, light.y - P.y, light.z - P.z}) : RtVec3{0.f, 1.f, 0.f};
        sh.push_back({{P.x + l.x * 0.001f, P.y + l.y * 0.001f, P.z + l.z * 0.001f}, l});
      }
      if (sh.empty()) continue;
      // (the kernel's choice of o: a point of the first ray's line, at the light's distance for a shadow ray)
      const float h0 = li == 0 ? sqrtf((light.x - sh[0].e.x) * (light.x - sh[0].e.x) + (light.y - sh[0].e.y) * (light.y - sh[0].e.y) + (light.z - sh[0].e.z) * (light.z - sh[0].e.z)) : 64.f;
      const RtVec3 o = {fmaf(h0, sh[0].d.x, sh[0].e.x), fmaf(h0, sh[0].d.y, sh[0].e.y), fmaf(h0, sh[0].d.z, sh[0].e.z)};
      std::vector<Hit> sh_hits(sh.size(), Hit{64.f, -1, 0, 0});
      walk(T, 0, sh, sh_hits, std::vector<char>(sh.size(), 1), bundle_of(sh, o));
    }
  }
  return 0;
}

static int run_random(int ac, char** av) {
  if (ac < 4) return 2;
  const uint64_t count = strtoull(av[2], nullptr, 10);
  std::mt19937_64 rng(strtoull(av[3], nullptr, 10));
  auto U = [&](double a, double b) { return a + (b - a) * (double)(rng() >> 11) * 0x1p-53; };
  auto LOGU = [&](double a, double b) { return exp(U(log(a), log(b))); };
  for (uint64_t it = 0; it < count; it++) {
    // a triangle: size 1e-3 .. 10, now and then a sliver or nearly degenerate, somewhere in the scene's cube
    const double size = LOGU(1e-3, 10.0);
    double p[3][3];
    for (int k = 0; k < 3; k++) p[0][k] = U(-16, 16);
    for (int q = 1; q < 3; q++) for (int k = 0; k < 3; k++) p[q][k] = p[0][k] + U(-size, size);
    const int shape = (int)(rng() % 8);
    if (shape == 0) for (int k = 0; k < 3; k++) p[2][k] = p[0][k] + (p[1][k] - p[0][k]) * U(-1, 2) + U(-1, 1) * size * LOGU(1e-7, 1e-2);   // sliver
    if (shape == 1) for (int k = 0; k < 3; k++) p[2][k] = p[1][k];                                                                      // degenerate
    float v0[3], e1[3], e2[3];
    for (int k = 0; k < 3; k++) { v0[k] = (float)p[0][k]; e1[k] = (float)p[1][k] - v0[k]; e2[k] = (float)p[2][k] - v0[k]; }
    double n[3] = {(double)e1[1] * e2[2] - (double)e1[2] * e2[1], (double)e1[2] * e2[0] - (double)e1[0] * e2[2], (double)e1[0] * e2[1] - (double)e1[1] * e2[0]};
    const double nl = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    // a target point in the triangle's plane: outside it by a small distance (relative to its size), or far away
    double be = U(-0.5, 1.5), ga = U(-0.5, 1.5);
    const int where = (int)(rng() % 4);
    const double off = LOGU(1e-7, 1e-1);
    if (where == 0) { be = -off; ga = U(0, 1); } else if (where == 1) { ga = -off; be = U(0, 1); } else if (where == 2) { be = U(0, 1); ga = 1 - be + off; }
    else { be = U(-30, 30); ga = U(-30, 30); }
    double X[3]; for (int k = 0; k < 3; k++) X[k] = v0[k] + be * e1[k] + ga * e2[k];
    // a direction: anything, or nearly in the plane, or in the plane along an edge
    double d[3] = {U(-1, 1), U(-1, 1), U(-1, 1)};
    const int dir = (int)(rng() % 4);
    if (dir >= 1 && nl > 0) {
      const double* ed = dir == 1 ? nullptr : nullptr; (void)ed;
      double a = U(-1, 1), b = U(-1, 1); if (dir == 2) { a = 1; b = 0; } if (dir == 3) { a = 0; b = 1; }
      const double tilt = (rng() % 3 == 0) ? 0.0 : LOGU(1e-9, 1e-2) * (rng() % 2 ? 1 : -1);
      for (int k = 0; k < 3; k++) d[k] = a * e1[k] + b * e2[k] + tilt * size * n[k] / nl;
    }
    const double dl = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); if (!(dl > 0)) continue;
    RtVec3 df = rt_normalized({(float)d[0], (float)d[1], (float)d[2]});
    if (rng() % 8 == 0) { const float sc = (float)U(0.9, 1.1); df = {df.x * sc, df.y * sc, df.z * sc}; }   // a direction that is not a unit vector
    const double back = U(0.0, 40.0);
    // a bundle: the ray alone, or with neighbours through a common point / parallel to it
    std::vector<Ray> rays;
    const RtVec3 e0 = {(float)(X[0] - back * df.x), (float)(X[1] - back * df.y), (float)(X[2] - back * df.z)};
    rays.push_back({e0, df});
    const int extra = (int)(rng() % 3) * 3;
    const double spread = LOGU(1e-6, 1e-2);
    const bool parallel = rng() % 2;
    for (int q = 0; q < extra; q++) {
      if (parallel) rays.push_back({{(float)(e0.x + U(-1, 1) * spread * 10), (float)(e0.y + U(-1, 1) * spread * 10), (float)(e0.z + U(-1, 1) * spread * 10)}, df});
      else rays.push_back({e0, rt_normalized({(float)(df.x + U(-1, 1) * spread), (float)(df.y + U(-1, 1) * spread), (float)(df.z + U(-1, 1) * spread)})});
    }
    check(v0, e1, e2, rays, bundle_of(rays, e0));
  }
  return 0;
}

int main(int ac, char** av) {
  int rc = 2;
  if (ac > 1 && !strcmp(av[1], "scene")) rc = run_scene(ac, av);
  else if (ac > 1 && !strcmp(av[1], "random")) rc = run_random(ac, av);
  if (rc) return rc;
  if (n_leaves) printf("leaf visits %llu, without a triangle left %llu (%.1f%%)\n", (unsigned long long)n_leaves, (unsigned long long)n_leaves_emptied, 100.0 * n_leaves_emptied / n_leaves);
  printf("checked %llu skipped %llu (%.1f%%) accepted %llu mismatches %llu \n", (unsigned long long)n_pairs, (unsigned long long)n_skip,
         n_pairs ? 100.0 * n_skip / n_pairs : 0.0, (unsigned long long)n_accept, (unsigned long long)n_bad);
  return n_bad ? 1 : 0;
}
