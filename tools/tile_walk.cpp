// tile_walk.cpp -- design study, not part of the product or the tests.
// Replays the wave-synchronous walk of single 8x8 tiles of a frame on the CPU (reference order, each ray with its own closest
// hit) and prints, per trace (primary rays, shadow rays towards a point light and straight up): the bundle of the rays
// (rt_math.h), node iterations of the wave, leaves entered, triangle iterations with and without the bundle test.
// Build: g++ -O2 -ffp-contract=off -Iinclude tools/tile_walk.cpp -Lraytracer_amd -lrt_amd -Wl,-rpath,$PWD/raytracer_amd -o /tmp/tile_walk
// Run:   /tmp/tile_walk assets/scenes/meshDragon.json 3840 2160 143,203 [more tiles x,y ...]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/rt_amd.h"
#include "../raytracer_amd/csrc/rt_math.h"

struct Ray { RtVec3 e, d; };
struct Hit { float t; int prim; float bet, gam; };
struct Tree { std::vector<int32_t> kind, refs, ref_first; std::vector<float> bounds, verts; std::vector<int32_t> child[8]; };
static void link(Tree& T, int& next, int& ro) {
  const int me = next++;
  if (T.kind[me] >= 0) { T.ref_first[me] = ro; ro += T.kind[me]; return; }
  for (int c = 0; c < 8; c++) { T.child[c][me] = next; link(T, next, ro); }
}
struct Stats { long node_it = 0, node_empty = 0, leaves = 0, tri_it = 0, tri_left = 0, leaves_left = 0, cand = 0; };

static RtBundle bundle_of(const std::vector<Ray>& rays, const RtVec3& o, float& r0, float& k) {
  const RtVec3 s = {rays.front().d.x + rays.back().d.x, rays.front().d.y + rays.back().d.y, rays.front().d.z + rays.back().d.z};
  const RtVec3 a = rt_normalized(s);
  float bw = 0; r0 = 0; k = 0;
  for (const Ray& r : rays) { float x, y, z; rt_bundle_lane(o, a, r.e, r.d, x, y, z); r0 = fmaxf(r0, x); bw = fmaxf(bw, y); k = fmaxf(k, z); }
  return rt_bundle_finish(o, a, r0, bw, k);
}

static void walk(const Tree& T, int node, const std::vector<Ray>& rays, std::vector<Hit>& hits, const std::vector<char>& mask, const RtBundle& B, bool cull, Stats& S) {
  for (int c = 0; c < 8; c++) {
    const int ch = T.child[c][node];
    if (T.kind[ch] == 0) continue;   // empty leaves are not visited
    const float* b = &T.bounds[6 * ch];
    std::vector<char> m(rays.size(), 0); int cnt = 0;
    for (size_t i = 0; i < rays.size(); i++)
      if (mask[i] && rt_node_accept_literal({b[0], b[1], b[2]}, {b[3], b[4], b[5]}, rays[i].e, rays[i].d, hits[i].t)) { m[i] = 1; cnt++; }
    S.node_it++;
    if (!cnt) { S.node_empty++; continue; }
    if (T.kind[ch] < 0) { walk(T, ch, rays, hits, m, B, cull, S); continue; }
    S.leaves++;
    int left = 0;
    for (int k = 0; k < T.kind[ch]; k++) {
      const int id = T.refs[T.ref_first[ch] + k];
      const float* v = &T.verts[9 * id];
      const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
      S.tri_it++;
      if (!(cull && rt_cull_test(rt_cull_record(v, e1, e2), B))) left++;
      for (size_t i = 0; i < rays.size(); i++) if (m[i]) {
        float t, bb, gg;
        if (rt_triangle_test({v[0], v[1], v[2]}, {e1[0], e1[1], e1[2]}, {e2[0], e2[1], e2[2]}, rays[i].e, rays[i].d, hits[i].t, t, bb, gg)) { hits[i] = {t, id, bb, gg}; S.cand++; }
      }
    }
    S.tri_left += left; if (left) S.leaves_left++;
  }
}

int main(int ac, char** av) {
  if (ac < 5) return 2;
  const int W = atoi(av[2]), H = atoi(av[3]);
  rt_scene* sc = nullptr;
  if (rt_scene_load(av[1], "assets/resources.json", "assets", W, H, 3, &sc)) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
  rt_scene_info I; rt_scene_get_info(sc, &I);
  Tree T; T.kind.resize(I.n_nodes); T.bounds.resize(6 * (size_t)I.n_nodes); T.refs.resize(I.n_leaf_refs + 1); T.verts.resize(9 * (size_t)I.n_triangles);
  std::vector<float> nrm(9 * (size_t)I.n_triangles); std::vector<int32_t> mesh(I.n_triangles);
  rt_scene_dump_tree(sc, T.kind.data(), T.bounds.data(), T.refs.data());
  rt_scene_dump_triangles(sc, T.verts.data(), nrm.data(), mesh.data());
  float cam[19]; rt_scene_dump_camera(sc, cam);
  T.ref_first.assign(I.n_nodes, 0); for (int c = 0; c < 8; c++) T.child[c].assign(I.n_nodes, -1);
  int next = 0, ro = 0; link(T, next, ro);
  const RtVec3 eye = {cam[9], cam[10], cam[11]}, light = {0.f, 10.f, 20.f};
  for (int a = 4; a < ac; a++) {
    int tx = 0, ty = 0; sscanf(av[a], "%d,%d", &tx, &ty);
    std::vector<Ray> rays;
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
      const int i = tx * 8 + x, j = H - 1 - (ty * 8 + y);
      const float ic = (float)((double)cam[12] + (double)((cam[13] - cam[12]) / (float)W) * (i + 0.5));
      const float jc = (float)((double)cam[14] + (double)((cam[15] - cam[14]) / (float)H) * (j + 0.5));
      const RtVec3 d = {ic * cam[0] + jc * cam[3] + -0.1f * cam[6], ic * cam[1] + jc * cam[4] + -0.1f * cam[7], ic * cam[2] + jc * cam[5] + -0.1f * cam[8]};
      rays.push_back({eye, rt_normalized(d)});
    }
    // the same rays walked cluster by cluster: the rays whose line passes within rs of the first remaining ray's point o and whose
    // direction is within ks of its direction go together, the others wait for a later walk (exact: a ray's result does not
    // depend on which rays walk with it)
    auto split_report = [&](const char* what, const std::vector<Ray>& rs, const std::vector<float>& hs, float rsplit, float ksplit) {
      std::vector<char> left(rs.size(), 1); size_t remaining = rs.size(); Stats S; int walks = 0;
      while (remaining) {
        size_t f = 0; while (!left[f]) f++;
        const RtVec3 o = {fmaf(hs[f], rs[f].d.x, rs[f].e.x), fmaf(hs[f], rs[f].d.y, rs[f].e.y), fmaf(hs[f], rs[f].d.z, rs[f].e.z)};
        std::vector<Ray> cl;
        for (size_t i = f; i < rs.size(); i++) if (left[i]) { float x, y, z; rt_bundle_lane(o, rs[f].d, rs[i].e, rs[i].d, x, y, z); if (x <= rsplit && z <= ksplit) { cl.push_back(rs[i]); left[i] = 0; remaining--; } }
        float r0, k; const RtBundle B = bundle_of(cl, o, r0, k);
        std::vector<Hit> hits(cl.size(), Hit{64.f, -1, 0, 0});
        walk(T, 0, cl, hits, std::vector<char>(cl.size(), 1), B, true, S); walks++;
      }
      printf("  %-10s split at r %.2f k %.3f: %d walks | node iterations %ld leaves %ld (%ld) | triangle iterations %ld -> %ld\n", what, rsplit, ksplit, walks, S.node_it, S.leaves, S.leaves_left, S.tri_it, S.tri_left);
    };
    auto report = [&](const char* what, const std::vector<Ray>& rs, const RtVec3& o, std::vector<Hit>& hits) {
      float r0, k; const RtBundle B = bundle_of(rs, o, r0, k);
      const bool on = r0 <= 1.0f && k <= 0.25f;
      Stats S; walk(T, 0, rs, hits, std::vector<char>(rs.size(), 1), B, on, S);
      printf("  %-10s rays %2zu bundle r0 %.4f ksp %.5f %s | node iterations %ld (%ld with no ray) leaves %ld (%ld with a triangle left) | triangle iterations %ld -> %ld | accepted %ld\n",
             what, rs.size(), r0, k, on ? "on " : "OFF", S.node_it, S.node_empty, S.leaves, S.leaves_left, S.tri_it, S.tri_left, S.cand);
    };
    printf("tile %d,%d\n", tx, ty);
    std::vector<Hit> hits(rays.size(), Hit{64.f, -1, 0, 0});
    report("primary", rays, eye, hits);
    int nhit = 0; float tmin = 1e9, tmax = 0; for (auto& h : hits) if (h.prim >= 0) { nhit++; tmin = fminf(tmin, h.t); tmax = fmaxf(tmax, h.t); }
    printf("  hits %d, t %.3f .. %.3f\n", nhit, tmin, tmax);
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
      const float h0 = li == 0 ? sqrtf((light.x - sh[0].e.x) * (light.x - sh[0].e.x) + (light.y - sh[0].e.y) * (light.y - sh[0].e.y) + (light.z - sh[0].e.z) * (light.z - sh[0].e.z)) : 64.f;
      const RtVec3 o = {fmaf(h0, sh[0].d.x, sh[0].e.x), fmaf(h0, sh[0].d.y, sh[0].e.y), fmaf(h0, sh[0].d.z, sh[0].e.z)};
      std::vector<Hit> sh_hits(sh.size(), Hit{64.f, -1, 0, 0});
      report(li == 0 ? "shadow-pt" : "shadow-dir", sh, o, sh_hits);
      std::vector<float> hs(sh.size());
      for (size_t i = 0; i < sh.size(); i++) hs[i] = li == 0 ? sqrtf((light.x - sh[i].e.x) * (light.x - sh[i].e.x) + (light.y - sh[i].e.y) * (light.y - sh[i].e.y) + (light.z - sh[i].e.z) * (light.z - sh[i].e.z)) : 64.f;
      for (float rs : {0.1f, 0.2f, 0.4f}) for (float ks : {0.004f, 0.01f}) split_report(li == 0 ? "shadow-pt" : "shadow-dir", sh, hs, rs, ks);
    }
  }
  return 0;
}
