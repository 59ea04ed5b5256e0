// bundle_study.cpp -- design study, not part of the product or the tests.
// Replays the wave-synchronous walk of 8x8 tiles of the meshDragon frame on the CPU (reference order, each ray with its own
// closest hit, shadow rays leaving at their first blocker) and counts, per kind of walk (primary rays, shadow rays towards
// the point light, shadow rays of the directional light), what several ways of putting the wave's rays in front of a leaf's
// triangles leave to be tested:
//   one    : one bundle around all rays of the walk (what the kernel does)
//   quad   : four bundles, one per 4x4 quadrant of the tile; a triangle is left out only if every quadrant that has a ray in
//            the leaf rules it out
//   strip  : the same with four 8x2 strips (sixteen consecutive lanes each)
//   depthK : K bundles of rays grouped by the distance of their primary hit (K equal parts of the tile's range)
// Build: g++ -O2 -ffp-contract=off -Iinclude tools/bundle_study.cpp -Lraytracer_amd -lrt_amd -Wl,-rpath,$PWD/raytracer_amd -o /tmp/bundle_study
// Run:   /tmp/bundle_study assets/scenes/meshDragon.json 3840 2160 [stride N | x,y ...]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/rt_amd.h"
#include "../raytracer_amd/csrc/rt_math.h"

struct Ray { RtVec3 e, d; float stop; int lane; };
struct Hit { float t; int prim; float bet, gam; };
struct Tree { std::vector<int32_t> kind, refs, ref_first; std::vector<float> bounds, verts; std::vector<int32_t> child[8]; std::vector<RtCullRec> cull; };
static void link(Tree& T, int& next, int& ro) {
  const int me = next++;
  if (T.kind[me] >= 0) { T.ref_first[me] = ro; ro += T.kind[me]; return; }
  for (int c = 0; c < 8; c++) { T.child[c][me] = next; link(T, next, ro); }
}

struct Sub { RtBundle B; bool on; uint64_t lanes; };          // one bundle and the rays (indices into the walk's ray list) it covers
struct Strategy { const char* name; std::vector<Sub> subs; long tri_left = 0, leaves_left = 0, passes = 0; };
struct Stats { long node_it = 0, leaves = 0, tri_it = 0, hit_it = 0, hit_lanes = 0, exact_nodes = 0; };

static Sub make_sub(const std::vector<Ray>& rays, uint64_t lanes, bool shadow) {
  Sub s; s.lanes = lanes; s.on = false; memset(&s.B, 0, sizeof s.B);
  if (!lanes) return s;
  int first = __builtin_ctzll(lanes), last = 63 - __builtin_clzll(lanes);
  const Ray& f = rays[first];
  const float h = shadow ? fminf(f.stop, 64.f) : 0.f;
  const RtVec3 o = {fmaf(h, f.d.x, f.e.x), fmaf(h, f.d.y, f.e.y), fmaf(h, f.d.z, f.e.z)};
  const RtVec3 sum = {f.d.x + rays[last].d.x, f.d.y + rays[last].d.y, f.d.z + rays[last].d.z};
  if (!(rt_dot(sum, sum) > 1.f)) return s;
  const RtVec3 a = rt_normalized(sum);
  float r0 = 0, bw = 0, k = 0;
  for (size_t i = 0; i < rays.size(); i++) if ((lanes >> i) & 1) { float x, y, z; rt_bundle_lane(o, a, rays[i].e, rays[i].d, x, y, z); r0 = fmaxf(r0, x); bw = fmaxf(bw, y); k = fmaxf(k, z); }
  s.on = r0 <= 1.0f && k <= 0.25f;
  s.B = rt_bundle_finish(o, a, r0, bw, k);
  return s;
}

static void walk(const Tree& T, int node, const std::vector<Ray>& rays, std::vector<Hit>& hits, uint64_t mask, uint64_t& live, std::vector<Strategy>& st, Stats& S) {
  for (int c = 0; c < 8; c++) {
    mask &= live;
    if (!mask) return;
    const int ch = T.child[c][node];
    if (T.kind[ch] == 0) continue;   // empty leaves are not visited
    const float* b = &T.bounds[6 * ch];
    uint64_t m = 0;
    for (size_t i = 0; i < rays.size(); i++)
      if (((mask >> i) & 1) && rt_node_accept_literal({b[0], b[1], b[2]}, {b[3], b[4], b[5]}, rays[i].e, rays[i].d, hits[i].t)) m |= 1ull << i;
    S.node_it++;
    if (!m) continue;
    if (T.kind[ch] < 0) { walk(T, ch, rays, hits, m, live, st, S); continue; }
    S.leaves++;
    const int n = T.kind[ch];
    const bool ask = n >= 3;
    for (Strategy& s : st) {
      int left = 0;
      for (int k = 0; k < n; k++) {
        const RtCullRec& cr = T.cull[T.ref_first[ch] + k];
        bool culled = ask && !s.subs.empty();
        for (const Sub& sb : s.subs) if (sb.lanes & m) { if (!sb.on || !rt_cull_test(cr, sb.B)) { culled = false; break; } }
        if (!culled) left++;
      }
      s.tri_left += left; if (left) s.leaves_left++;
      if (ask && !s.subs.empty()) s.passes += s.subs.size() > 1 ? (n + 15) / 16 : 1;
    }
    for (int k = 0; k < n; k++) {
      const int id = T.refs[T.ref_first[ch] + k];
      const float* v = &T.verts[9 * id];
      const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
      S.tri_it++;
      int hl = 0;
      for (size_t i = 0; i < rays.size(); i++) if ((m >> i) & 1) {
        float t, bb, gg;
        if (rt_triangle_test({v[0], v[1], v[2]}, {e1[0], e1[1], e1[2]}, {e2[0], e2[1], e2[2]}, rays[i].e, rays[i].d, hits[i].t, t, bb, gg)) { hits[i] = {t, id, bb, gg}; hl++; }
      }
      if (hl) { S.hit_it++; S.hit_lanes += hl; }
    }
    for (size_t i = 0; i < rays.size(); i++) if (((m >> i) & 1) && rays[i].stop >= 0.f && hits[i].t < rays[i].stop) live &= ~(1ull << i);
  }
}

struct Totals { Stats S; std::vector<Strategy> st; long walks = 0; };

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
  T.cull.resize(I.n_leaf_refs);
  for (int r = 0; r < I.n_leaf_refs; r++) {
    const float* v = &T.verts[9 * T.refs[r]];
    const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
    T.cull[r] = rt_cull_record(v, e1, e2);
  }
  const RtVec3 eye = {cam[9], cam[10], cam[11]}, light = {0.f, 10.f, 20.f};
  std::vector<std::pair<int, int>> tiles;
  bool verbose = true;
  if (!strcmp(av[4], "stride")) { const int s = atoi(av[5]); verbose = false; for (int y = s / 2; y < H / 8; y += s) for (int x = s / 2; x < W / 8; x += s) tiles.push_back({x, y}); }
  else for (int a = 4; a < ac; a++) { int tx = 0, ty = 0; sscanf(av[a], "%d,%d", &tx, &ty); tiles.push_back({tx, ty}); }
  const char* kinds[3] = {"primary", "shadow-pt", "shadow-dir"};
  Totals tot[3];
  auto strategies = [&](const std::vector<Ray>& rays, const std::vector<float>& depth, bool shadow) {
    std::vector<Strategy> st;
    const uint64_t all = rays.size() == 64 ? ~0ull : ((1ull << rays.size()) - 1);
    st.push_back({"none", {}});
    st.push_back({"one", {make_sub(rays, all, shadow)}});
    auto by = [&](const char* name, int K, auto key) {
      Strategy s{name, {}};
      for (int q = 0; q < K; q++) { uint64_t m = 0; for (size_t i = 0; i < rays.size(); i++) if (key(i) == q) m |= 1ull << i; if (m) s.subs.push_back(make_sub(rays, m, shadow)); }
      st.push_back(s);
    };
    by("quad", 4, [&](size_t i) { const int l = rays[i].lane; return ((l & 7) >> 2) + 2 * ((l >> 3) >> 2); });
    by("strip", 4, [&](size_t i) { return rays[i].lane >> 4; });
    float dmin = 1e30f, dmax = -1e30f; for (float d : depth) { dmin = fminf(dmin, d); dmax = fmaxf(dmax, d); }
    for (int K : {2, 4}) by(K == 2 ? "depth2" : "depth4", K, [&](size_t i) { const float f = dmax > dmin ? (depth[i] - dmin) / (dmax - dmin) : 0.f; int q = (int)(f * K); return q >= K ? K - 1 : q; });
    // quadrants, each split once more by depth (eight bundles)
    by("quad-d2", 8, [&](size_t i) { const int l = rays[i].lane; const int q = ((l & 7) >> 2) + 2 * ((l >> 3) >> 2); const float f = dmax > dmin ? (depth[i] - dmin) / (dmax - dmin) : 0.f; return q * 2 + (f >= 0.5f ? 1 : 0); });
    return st;
  };
  auto add = [&](Totals& t, const Stats& S, const std::vector<Strategy>& st) {
    t.S.node_it += S.node_it; t.S.leaves += S.leaves; t.S.tri_it += S.tri_it; t.S.hit_it += S.hit_it; t.S.hit_lanes += S.hit_lanes; t.walks++;
    if (t.st.empty()) { t.st = st; for (auto& s : t.st) { s.subs.clear(); s.tri_left = s.leaves_left = s.passes = 0; } }
    for (size_t i = 0; i < st.size(); i++) { t.st[i].tri_left += st[i].tri_left; t.st[i].leaves_left += st[i].leaves_left; t.st[i].passes += st[i].passes; }
  };
  auto show = [&](const char* what, const Stats& S, const std::vector<Strategy>& st, const std::vector<Strategy>* detail) {
    printf("  %-10s node it %ld leaves %ld tri it %ld (with an accepted hit: %ld, lanes %ld) |", what, S.node_it, S.leaves, S.tri_it, S.hit_it, S.hit_lanes);
    for (const auto& s : st) printf(" %s %ld/%ld(p%ld)", s.name, s.tri_left, s.leaves_left, s.passes);
    // crude cost in issued instructions: 20 per node iteration, 40 per pass of the bundle test, 25 per triangle iteration
    printf("\n  %-10s cost k-instr:", "");
    for (const auto& s : st) printf(" %s %.1f", s.name, (20.0 * S.node_it + 40.0 * s.passes + 25.0 * s.tri_left) / 1e3);
    printf("\n");
    if (detail) for (const auto& s : *detail) if (s.subs.size() >= 1 && strcmp(s.name, "none")) { printf("    %-8s", s.name); for (const auto& sb : s.subs) printf(" [%d rays r0 %.3f ksp %.4f %s]", __builtin_popcountll(sb.lanes), sb.B.r0, sb.B.ksp, sb.on ? "on" : "OFF"); printf("\n"); }
  };
  for (auto [tx, ty] : tiles) {
    std::vector<Ray> rays;
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
      const int i = tx * 8 + x, j = H - 1 - (ty * 8 + y);
      const float ic = (float)((double)cam[12] + (double)((cam[13] - cam[12]) / (float)W) * (i + 0.5));
      const float jc = (float)((double)cam[14] + (double)((cam[15] - cam[14]) / (float)H) * (j + 0.5));
      const RtVec3 d = {ic * cam[0] + jc * cam[3] + -0.1f * cam[6], ic * cam[1] + jc * cam[4] + -0.1f * cam[7], ic * cam[2] + jc * cam[5] + -0.1f * cam[8]};
      rays.push_back({eye, rt_normalized(d), -1.f, y * 8 + x});
    }
    if (verbose) printf("tile %d,%d\n", tx, ty);
    std::vector<Hit> hits(rays.size(), Hit{64.f, -1, 0, 0});
    {
      std::vector<float> depth(rays.size(), 0.f);
      auto st = strategies(rays, depth, false); Stats S; uint64_t live = ~0ull;
      walk(T, 0, rays, hits, ~0ull, live, st, S);
      add(tot[0], S, st); if (verbose) show(kinds[0], S, st, &st);
    }
    for (int li = 0; li < 2; li++) {
      std::vector<Ray> sh; std::vector<float> depth;
      for (size_t i = 0; i < rays.size(); i++) if (hits[i].prim >= 0) {
        const float* v = &T.verts[9 * hits[i].prim];
        const float alp = 1 - hits[i].bet - hits[i].gam, bet = hits[i].bet, gam = hits[i].gam;
        const RtVec3 P = {(v[0] * alp + v[3] * bet) + v[6] * gam, (v[1] * alp + v[4] * bet) + v[7] * gam, (v[2] * alp + v[5] * bet) + v[8] * gam};
        const RtVec3 l = li == 0 ? rt_normalized({light.x - P.x, light.y - P.y, light.z - P.z}) : RtVec3{0.f, 1.f, 0.f};
        const RtVec3 sP = {P.x + l.x * 0.001f, P.y + l.y * 0.001f, P.z + l.z * 0.001f};
        const float ld = li == 0 ? rt_len({light.x - sP.x, light.y - sP.y, light.z - sP.z}) : 64.f;
        sh.push_back({sP, l, ld, rays[i].lane}); depth.push_back(hits[i].t);
      }
      if (sh.empty()) continue;
      std::vector<Hit> sh_hits(sh.size(), Hit{64.f, -1, 0, 0});
      auto st = strategies(sh, depth, true); Stats S; uint64_t live = sh.size() == 64 ? ~0ull : ((1ull << sh.size()) - 1);
      walk(T, 0, sh, sh_hits, live, live, st, S);
      add(tot[1 + li], S, st); if (verbose) show(kinds[1 + li], S, st, &st);
    }
  }
  printf("TOTAL over %zu tiles\n", tiles.size());
  for (int k = 0; k < 3; k++) if (tot[k].walks) { printf(" walks %ld\n", tot[k].walks); show(kinds[k], tot[k].S, tot[k].st, nullptr); }
  return 0;
}
