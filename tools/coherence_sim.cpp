// coherence_sim.cpp -- design study, not part of the product or the tests.
//
// Replays the reference traversal (depth-first, children 0..7, running closest t) for the rays of
// 8x8 pixel tiles and prices two execution models for a 64-lane wave:
//   SIMT  : every lane walks its own stack; an iteration costs one node test for the wave plus the
//           longest leaf loop among the lanes that reached a leaf in that iteration.
//   SYNC  : the wave walks the union of the lanes' subtrees once, in the common child order, with a
//           lane mask per stack level; every node / triangle of the union costs one test for the wave.
// Build: g++ -O2 -ffp-contract=off -Iinclude tools/coherence_sim.cpp -Lraytracer_amd -lrt_amd -Wl,-rpath,$PWD/raytracer_amd -o /tmp/coherence_sim
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/rt_amd.h"
#include "../raytracer_amd/csrc/rt_math.h"

struct Tree {
  std::vector<int32_t> kind; std::vector<float> bounds; std::vector<int32_t> refs;
  std::vector<int32_t> ref_first;      // per node (preorder index): offset into refs for leaves
  std::vector<int32_t> child[8];       // per node: preorder index of child c (branches)
  std::vector<float> verts;
};

static int build_links(Tree& T, int id, int& next, int& ref_off) {
  // nodes are in preorder; returns index after this subtree
  int me = next++;
  if (T.kind[me] >= 0) { T.ref_first[me] = ref_off; ref_off += T.kind[me]; return me; }
  for (int c = 0; c < 8; c++) { T.child[c][me] = next; build_links(T, 0, next, ref_off); }
  return me;
}

struct Ray { RtVec3 e, d; float t; int prim; float bet, gam; bool alive; };

static bool accept(const Tree& T, int node, const Ray& r) {
  const float* b = &T.bounds[6 * node];
  return rt_node_accept_literal({b[0], b[1], b[2]}, {b[3], b[4], b[5]}, r.e, r.d, r.t);
}
static void leaf(const Tree& T, int node, Ray& r) {
  for (int k = 0; k < T.kind[node]; k++) {
    int id = T.refs[T.ref_first[node] + k];
    const float* v = &T.verts[9 * id];
    RtVec3 v0 = {v[0], v[1], v[2]}, e1 = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2 = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
    float t, bb, gg;
    if (rt_triangle_test(v0, e1, e2, r.e, r.d, r.t, t, bb, gg)) { r.t = t; r.prim = id; r.bet = bb; r.gam = gg; }
  }
}

#include <unordered_map>
static std::unordered_map<int, unsigned long long> g_tested;   // per trace: lanes that have tested a triangle (mailbox study)
static double g_skippable = 0, g_total_iter = 0, g_last8 = 0, g_last16 = 0;
static std::vector<int> g_recent;
struct Cost { double simt_node = 0, simt_tri = 0, sync_node = 0, sync_tri = 0, lane_node = 0, lane_tri = 0, waves = 0, sync_tri_active = 0, sync_node_active = 0; };

// SIMT replay of up to 64 rays in lockstep.
static void run_simt(const Tree& T, std::vector<Ray>& rays, Cost& C) {
  const int n = (int)rays.size();
  struct St { int node[24], cur[24], sp; };
  std::vector<St> st(n);
  for (int i = 0; i < n; i++) { st[i].sp = rays[i].alive ? 1 : 0; st[i].node[0] = 0; st[i].cur[0] = 0; }
  for (;;) {
    bool any = false; int maxleaf = 0;
    for (int i = 0; i < n; i++) {
      St& s = st[i];
      if (!s.sp) continue;
      any = true;
      int parent = s.node[s.sp - 1], c = s.cur[s.sp - 1];
      if (c == 7) s.sp--; else s.cur[s.sp - 1] = c + 1;
      int ch = T.child[c][parent];
      C.lane_node++;
      if (!accept(T, ch, rays[i])) continue;
      if (T.kind[ch] < 0) { s.node[s.sp] = ch; s.cur[s.sp] = 0; s.sp++; }
      else { leaf(T, ch, rays[i]); C.lane_tri += T.kind[ch]; if (T.kind[ch] > maxleaf) maxleaf = T.kind[ch]; }
    }
    if (!any) break;
    C.simt_node += 1; C.simt_tri += maxleaf;
  }
}

// SYNC replay: one shared walk, lane masks per level.
static void sync_rec(const Tree& T, int node, std::vector<Ray>& rays, const std::vector<char>& mask, Cost& C) {
  for (int c = 0; c < 8; c++) {
    int ch = T.child[c][node];
    std::vector<char> m(rays.size(), 0); int cnt = 0, act = 0;
    for (size_t i = 0; i < rays.size(); i++) if (mask[i]) { act++; if (accept(T, ch, rays[i])) { m[i] = 1; cnt++; } }
    C.sync_node += 1; C.sync_node_active += act;
    if (!cnt) continue;
    if (T.kind[ch] < 0) sync_rec(T, ch, rays, m, C);
    else {
      for (size_t i = 0; i < rays.size(); i++) if (m[i]) leaf(T, ch, rays[i]);
      {
        unsigned long long mm = 0; for (size_t i = 0; i < rays.size(); i++) if (m[i]) mm |= 1ull << i;
        for (int k = 0; k < T.kind[ch]; k++) {
          int id = T.refs[T.ref_first[ch] + k];
          unsigned long long& t = g_tested[id];
          g_total_iter++;
          if ((mm & ~t) == 0) {
            g_skippable++;
            // would a small "recently tested" window have caught it?
            int pos = -1; for (int q = (int)g_recent.size() - 1, n = 0; q >= 0 && n < 64; q--, n++) if (g_recent[q] == id) { pos = n; break; }
            if (pos >= 0 && pos < 32) g_last8++;
            if (pos >= 0 && pos < 64) g_last16++;
          }
          t |= mm;
          g_recent.push_back(id);
        }
      }
      C.sync_tri += T.kind[ch]; C.sync_tri_active += (double)T.kind[ch] * cnt;
    }
  }
}

int main(int ac, char** av) {
  const char* scene = ac > 1 ? av[1] : "assets/scenes/meshDragon.json";
  int W = ac > 2 ? atoi(av[2]) : 3840, H = ac > 3 ? atoi(av[3]) : 2160, stride = ac > 4 ? atoi(av[4]) : 7;
  int tw = ac > 5 ? atoi(av[5]) : 8, th = ac > 6 ? atoi(av[6]) : 8;
  rt_scene* sc = nullptr;
  if (rt_scene_load(scene, "assets/resources.json", "assets", W, H, 3, &sc)) { printf("%s\n", rt_last_error()); return 1; }
  rt_scene_info I; rt_scene_get_info(sc, &I);
  Tree T; T.kind.resize(I.n_nodes); T.bounds.resize(6 * I.n_nodes); T.refs.resize(I.n_leaf_refs + 1);
  T.verts.resize(9 * I.n_triangles); std::vector<float> nrm(9 * I.n_triangles); std::vector<int32_t> mesh(I.n_triangles);
  rt_scene_dump_tree(sc, T.kind.data(), T.bounds.data(), T.refs.data());
  rt_scene_dump_triangles(sc, T.verts.data(), nrm.data(), mesh.data());
  float cam[19]; rt_scene_dump_camera(sc, cam);
  T.ref_first.assign(I.n_nodes, 0); for (int c = 0; c < 8; c++) T.child[c].assign(I.n_nodes, -1);
  int next = 0, ro = 0; build_links(T, 0, next, ro);
  const float lightpos[3] = {0, 10, 20};   // meshDragon's point light; directional light is (0,-1,0)
  Cost P, S1, S2;
  int tiles = 0;
  for (int ty = 0; ty < H / th; ty += stride) for (int tx = (ty / stride) % stride; tx < W / tw; tx += stride) {
    std::vector<Ray> rays;
    for (int y = 0; y < th; y++) for (int x = 0; x < tw; x++) {
      int i = tx * tw + x, row = ty * th + y, j = H - 1 - row;
      float ic = (float)((double)cam[12] + (double)((cam[13] - cam[12]) / (float)W) * (i + 0.5));
      float jc = (float)((double)cam[14] + (double)((cam[15] - cam[14]) / (float)H) * (j + 0.5));
      RtVec3 d = {ic * cam[0] + jc * cam[3] + -0.1f * cam[6], ic * cam[1] + jc * cam[4] + -0.1f * cam[7], ic * cam[2] + jc * cam[5] + -0.1f * cam[8]};
      d = rt_normalized(d);
      rays.push_back({{cam[9], cam[10], cam[11]}, d, 64.f, -1, 0, 0, true});
    }
    std::vector<Ray> a = rays, b = rays;
    run_simt(T, a, P); P.waves++;
    std::vector<char> all(rays.size(), 1);
    g_tested.clear(); g_recent.clear();
    sync_rec(T, 0, b, all, P);
    for (size_t i = 0; i < a.size(); i++) if (a[i].prim != b[i].prim || a[i].t != b[i].t) { printf("MISMATCH simt/sync\n"); return 2; }
    // shadow rays of the hit pixels (no early exit here: full reference traversal)
    for (int li = 0; li < 2; li++) {
      std::vector<Ray> sh;
      for (size_t i = 0; i < a.size(); i++) {
        Ray r = a[i]; r.alive = a[i].prim >= 0;
        if (r.alive) {
          const float* v = &T.verts[9 * a[i].prim];
          float alp = 1 - a[i].bet - a[i].gam, bet = a[i].bet, gam = a[i].gam;
          RtVec3 Pp = {(v[0] * alp + v[3] * bet) + v[6] * gam, (v[1] * alp + v[4] * bet) + v[7] * gam, (v[2] * alp + v[5] * bet) + v[8] * gam};
          RtVec3 l = li == 0 ? rt_normalized({lightpos[0] - Pp.x, lightpos[1] - Pp.y, lightpos[2] - Pp.z}) : RtVec3{0.f * -1, -1.f * -1, 0.f * -1};
          r.e = {Pp.x + l.x * 0.001f, Pp.y + l.y * 0.001f, Pp.z + l.z * 0.001f}; r.d = l; r.t = 64.f; r.prim = -1;
        }
        sh.push_back(r);
      }
      std::vector<Ray> sa = sh, sb = sh;
      Cost& S = li == 0 ? S1 : S2;
      bool anyalive = false; std::vector<char> m(sh.size(), 0);
      for (size_t i = 0; i < sh.size(); i++) if (sh[i].alive) { m[i] = 1; anyalive = true; }
      if (!anyalive) continue;
      run_simt(T, sa, S); S.waves++;
      g_tested.clear(); g_recent.clear();
      sync_rec(T, 0, sb, m, S);
    }
    tiles++;
  }
  auto rep = [&](const char* name, const Cost& C) {
    printf("%-10s waves %6.0f | per wave: SIMT node-iters %7.1f tri-iters %7.1f | SYNC node %7.1f tri %7.1f | lane-sum/64: node %7.1f tri %7.1f | SYNC active lanes: node %.1f tri %.1f\n",
           name, C.waves, C.simt_node / C.waves, C.simt_tri / C.waves, C.sync_node / C.waves, C.sync_tri / C.waves,
           C.lane_node / C.waves / 64, C.lane_tri / C.waves / 64, C.sync_node_active / C.sync_node, C.sync_tri_active / (C.sync_tri > 0 ? C.sync_tri : 1));
  };
  printf("mailbox study: triangle iterations %.0f, skippable (all active lanes already tested it in this trace) %.0f = %.1f%%, of which within last 32 iterations %.1f%%, last 64 %.1f%%\n",
         g_total_iter, g_skippable, 100 * g_skippable / g_total_iter, 100 * g_last8 / g_total_iter, 100 * g_last16 / g_total_iter);
  printf("scene %s %dx%d, %d tiles of %dx%d sampled\n", scene, W, H, tiles, tw, th);
  rep("primary", P); rep("shadow-pt", S1); rep("shadow-dir", S2);
  return 0;
}
