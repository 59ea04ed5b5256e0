// rt_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Scalar CPU restatement of the per-pixel render path of masotrix/raytracer,
// used as the parity checker for the HIP kernels and as the timed CPU baseline.
// Nothing under raytracer_amd/ may include, link or call this file.
//
// It restates, operation for operation (IEEE float32, no FMA contraction:
// build with -ffp-contract=off), the reference's device path and the host
// feeder arithmetic that path depends on:
//   camera basis / frustum ......... camera/camera.cu:45-74
//   per-pixel loop, quantisation ... camera/camera.cu:84-124
//   closest hit + shadow + dispatch  ray/ray.cu:29-87
//   octree build (host) ............ octtree/octtree.cu:10-20,90-112,162-174, main.cu:25-31
//   node face test, leaf loop ...... octtree/octtree.cu:264-321
//   triangle SAT + Cramer test ..... triangle/triangle.cu:11-14,61-152
//   sphere test .................... spheres/spheres.cu:73-100
//   materials ...................... material/material.cu:10-101
//   lights ......................... lights/lights.cu:260-367
//   vector helpers, ray stack ...... vector/vector.cu:48-63,90-216
//   mesh load / transforms ......... parsing/parsing.cu:21-130, mesh/mesh.cu:22-48,65-97, vertex/vertex.cu:8-12
//   scene + resource parsing ....... parsing/parsing.cu:145-473
//   constants ...................... model/model.cuh:21-25, camera/camera.cuh:5
//
// The reference itself (CUDA + Boost.PropertyTree + Boost.GIL + libpng) cannot
// be built in the authoring image, so this restatement is pinned by (a) the
// reference's committed render of scenes/sphere.json (image.png, byte-exact)
// and (b) the ray / node-test / triangle-test counters and the 4K meshDragon
// frame digest that were recorded from a run of the reference's own routines
// (tests/golden/reference_counters.json).
//
// Defined behaviour chosen where the reference has undefined behaviour:
//   * refraction-level index (ray.cuh:10, material.cu:40,61,87) is clamped to
//     [0,2]: leaving at level 0 stays at level 0, entering at level 2
//     overwrites level[2].  Events are counted.
//   * a push onto a full 10-entry ray stack (ray.cuh:15-18) is dropped and counted.
//   * float -> uint8: after min(c,1)*255 (NaN compares false, so NaN -> 1 -> 255
//     exactly as vector.cu:90 does) negative values are stored as 0.
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace {

// ---------------------------------------------------------------- constants
const float kInf = 64.0f;       // model.cuh:21,24
const float kEps = 0.001f;      // model.cuh:22,25 (double literal narrowed to float at every use)
const float kNear = 0.1f;       // camera.cuh:5
const int kLeafCap = 50;        // octtree.cu:17
const int kRayStack = 10;       // ray.cuh:16
const int kOctStack = 20;       // octtree.cuh:23

// ------------------------------------------------------ host vector helpers
// vector.cu:48-50: products in float, accumulation in double, result float.
inline float hdot(const float* a, const float* b) {
  double acc = 0.0;
  acc = acc + (a[0] * b[0]);
  acc = acc + (a[1] * b[1]);
  acc = acc + (a[2] * b[2]);
  return (float)acc;
}
inline float hmag(const float* v) { return sqrtf(hdot(v, v)); }   // vector.cu:57-59
inline void hnormalize(float* v) {                                   // vector.cu:61-63
  float m = hmag(v);
  v[0] = v[0] / m; v[1] = v[1] / m; v[2] = v[2] / m;
}
inline void hcross(float* r, const float* u, const float* v) {       // vector.cu:52-55
  float x = u[1] * v[2] - u[2] * v[1];
  float y = u[2] * v[0] - u[0] * v[2];
  float z = u[0] * v[1] - u[1] * v[0];
  r[0] = x; r[1] = y; r[2] = z;
}

// ---------------------------------------------------- device-side helpers
inline float ddot(const float* a, const float* b) {                  // vector.cu:106-108
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
inline float dlen(const float* v) {                                  // normf(3,v)
  return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
}
inline void dnormalize(float* v) {                                   // vector.cu:110-112
  float m = dlen(v);
  v[0] /= m; v[1] /= m; v[2] /= m;
}
inline float fminr(float a, float b) { return a < b ? a : b; }       // vector.cu:90
inline float fmaxr(float a, float b) { return a > b ? a : b; }       // vector.cu:92

// Cofactor solve of [c0 c1 c2] x = b with the operation order of
// triangle.cu:121-133 / octtree.cu:301-313.
inline void cofactor_solve(const float* c0, const float* c1, const float* c2,
                           const float* b, float* x) {
  const float A00 = c0[0], A10 = c0[1], A20 = c0[2];
  const float A01 = c1[0], A11 = c1[1], A21 = c1[2];
  const float A02 = c2[0], A12 = c2[1], A22 = c2[2];
  float m00 = A11 * A22 - A12 * A21;
  float m01 = -A10 * A22 + A20 * A12;
  float m02 = A10 * A21 - A20 * A11;
  float m10 = -A01 * A22 + A21 * A02;
  float m11 = A00 * A22 - A20 * A02;
  float m12 = -A00 * A21 + A20 * A01;
  float m20 = A01 * A12 - A11 * A02;
  float m21 = -A00 * A12 + A10 * A02;
  float m22 = A00 * A11 - A10 * A01;
  float det = A00 * m00 + A10 * m10 + A20 * m20;
  x[0] = (m00 * b[0] + m10 * b[1] + m20 * b[2]) / det;
  x[1] = (m01 * b[0] + m11 * b[1] + m21 * b[2]) / det;
  x[2] = (m02 * b[0] + m12 * b[1] + m22 * b[2]) / det;
}

// ------------------------------------------------------------------ JSON
// Ordered tree; every scalar keeps its literal text (numbers are re-parsed
// with strtof/atoi by the consumer, as parsing.cu does through stof/stoi).
struct JNode {
  std::string text;                                      // scalar text
  std::vector<std::pair<std::string, JNode>> kids;       // object members / array items (key "")
};

struct JParser {
  const std::string& s; size_t p = 0; bool ok = true;
  explicit JParser(const std::string& src) : s(src) {}
  void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) p++; }
  std::string str() {
    std::string o; p++;
    while (p < s.size() && s[p] != '"') {
      if (s[p] == '\\' && p + 1 < s.size()) {
        char c = s[++p];
        if (c == 'n') o += '\n'; else if (c == 't') o += '\t'; else o += c;
      } else o += s[p];
      p++;
    }
    p++; return o;
  }
  JNode value() {
    JNode n; ws();
    if (p >= s.size()) { ok = false; return n; }
    if (s[p] == '{') {
      p++; ws();
      if (s[p] == '}') { p++; return n; }
      while (ok) {
        ws(); if (s[p] != '"') { ok = false; break; }
        std::string k = str(); ws();
        if (s[p] != ':') { ok = false; break; }
        p++;
        JNode v = value();
        n.kids.emplace_back(k, std::move(v)); ws();
        if (s[p] == ',') { p++; continue; }
        if (s[p] == '}') { p++; break; }
        ok = false;
      }
    } else if (s[p] == '[') {
      p++; ws();
      if (s[p] == ']') { p++; return n; }
      while (ok) {
        JNode v = value();
        n.kids.emplace_back("", std::move(v)); ws();
        if (s[p] == ',') { p++; continue; }
        if (s[p] == ']') { p++; break; }
        ok = false;
      }
    } else if (s[p] == '"') {
      n.text = str();
    } else {
      size_t b = p;
      while (p < s.size() && s[p] != ',' && s[p] != '}' && s[p] != ']' &&
             s[p] != ' ' && s[p] != '\n' && s[p] != '\r' && s[p] != '\t') p++;
      n.text = s.substr(b, p - b);
    }
    return n;
  }
};

bool read_json(const std::string& path, JNode& out) {
  std::ifstream f(path);
  if (!f.is_open()) return false;
  std::stringstream ss; ss << f.rdbuf();
  std::string src = ss.str();
  JParser P(src);
  out = P.value();
  return P.ok;
}
inline float jf(const JNode& n) { return strtof(n.text.c_str(), nullptr); }
inline void jvec(const JNode& n, std::vector<float>& v) { for (auto& k : n.kids) v.push_back(jf(k.second)); }

// ------------------------------------------------------------ scene model
enum MatType { LAMBERT = 0, PHONG = 1, REFLECTIVE = 2, DIELECTRIC = 3 };
enum LightType { PUNCTUAL = 0, SPOT = 1, AMBIENT = 2, DIRECTIONAL = 3 };

struct Material { int type; float col[3]; float att[3]; float param; int tex; bool bilinear, for_ambient; };   // tex >= 0: colour from that texture
struct Texture { int w, h; std::vector<uint8_t> rgb; };   // RGB8, row 0 on top
struct Light { int type; float col[3], pos[3], dir[3], angle; };
struct Sphere { float c[3], r; std::vector<int> mats; };
struct Tri { float v[3][3]; float n[3][3]; int mesh; float uv[3][2]; };

struct ONode {            // pointer-free restatement of OctNode/OctBranch/OctLeave
  float ini[3], end[3];
  bool leaf;
  int child[8];           // branch
  std::vector<int> objs;  // leaf: triangle ids in insertion order
};

struct Camera {
  float u[3], v[3], w[3], pos[3], left, right, bottom, top, scene_refr, bg[3];
  int W, H, depth; bool shadows;
};

struct Scene {
  Camera cam;
  std::vector<Material> mats;
  std::vector<Light> lights;
  std::vector<Sphere> spheres;
  std::vector<Tri> tris;
  std::vector<std::vector<int>> mesh_mats;
  std::vector<Texture> textures;
  std::vector<ONode> nodes;   // nodes[0] = root when has_oct
  bool has_oct = false;
  std::string error;
};

struct Counters {
  uint64_t rays, shadow_rays, node_tests, leaf_visits, tri_tests, sphere_tests;
  uint64_t ub_leave_level0, ub_level_overflow, ub_stack_overflow;
};

// ----------------------------------------------------------- OBJ loading
// Emulates std::getline(stream, token, delim) on one line (parsing.cu:35-118).
struct LineTok {
  std::string s; size_t p = 0; bool fail = false;
  void reset(const std::string& line) { s = line; p = 0; fail = false; }
  std::string next(char d) {
    if (fail || p >= s.size()) { fail = true; return std::string(); }
    size_t e = s.find(d, p);
    std::string t;
    if (e == std::string::npos) { t = s.substr(p); p = s.size(); }
    else { t = s.substr(p, e - p); p = e + 1; }
    return t;
  }
};

struct MeshData {
  struct V { float p[3]; float n[3]; bool has_n; std::vector<int> adj; float uv[2]; };
  std::map<int, V> verts;                 // keyed by position index, first-seen attributes win
  std::vector<std::array<int, 3>> faces;  // vertex keys
  std::vector<std::array<float, 3>> fn;   // face normals
};

void face_normal(MeshData& m, size_t f) {                            // triangle.cu:11-14
  const float* p0 = m.verts[m.faces[f][0]].p;
  const float* p1 = m.verts[m.faces[f][1]].p;
  const float* p2 = m.verts[m.faces[f][2]].p;
  float a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
  float b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
  float n[3]; hcross(n, a, b); hnormalize(n);
  m.fn[f] = {n[0], n[1], n[2]};
}
void vertex_normal(MeshData& m, MeshData::V& v) {                    // vertex.cu:8-12
  float n[3] = {0, 0, 0};
  for (int f : v.adj) { n[0] = n[0] + m.fn[f][0]; n[1] = n[1] + m.fn[f][1]; n[2] = n[2] + m.fn[f][2]; }
  hnormalize(n);
  v.n[0] = n[0]; v.n[1] = n[1]; v.n[2] = n[2]; v.has_n = true;
}

bool load_obj(const std::string& path, MeshData& m, std::string& err) {
  std::ifstream file(path);
  if (!file.is_open()) { err = "No " + path + " has been found"; return false; }
  std::vector<std::array<float, 3>> v, n; std::vector<std::array<float, 2>> uv;
  std::string s, token; LineTok lt;
  auto next_line = [&]() {
    if (!std::getline(file, s)) s.clear();
    lt.reset(s); token = lt.next(' ');
  };
  int guard = 0;
  do { next_line(); if (file.eof() && ++guard > 4) { err = "no vertices in " + path; return false; } } while (token != "v");
  while (token == "v") {
    std::array<float, 3> p;
    for (int i = 0; i < 3; i++) p[i] = strtof(lt.next(' ').c_str(), nullptr);
    v.push_back(p); next_line();
  }
  while (token == "vt") {
    std::array<float, 2> p;
    for (int i = 0; i < 2; i++) p[i] = strtof(lt.next(' ').c_str(), nullptr);
    uv.push_back(p); next_line();
  }
  while (token == "vn") {
    std::array<float, 3> p;
    for (int i = 0; i < 3; i++) p[i] = strtof(lt.next(' ').c_str(), nullptr);
    n.push_back(p); next_line();
  }
  guard = 0;
  while (token != "f") { next_line(); if (file.eof() && ++guard > 4) { err = "no faces in " + path; return false; } }
  while (token == "f") {
    std::array<int, 3> fk;
    for (int i = 0; i < 3; i++) {
      int vin, nin = -1, uvin = -1;
      if (n.size()) vin = (int)strtoul(lt.next('/').c_str(), nullptr, 10) - 1;
      else vin = (int)strtoul(lt.next(' ').c_str(), nullptr, 10) - 1;
      if (uv.size()) uvin = (int)strtoul(lt.next('/').c_str(), nullptr, 10) - 1;
      else if (n.size()) (void)lt.next('/');
      if (n.size()) nin = (int)strtoul(lt.next(' ').c_str(), nullptr, 10) - 1;
      if (vin < 0 || vin >= (int)v.size()) { err = "bad vertex index in " + path; return false; }
      if (m.verts.find(vin) == m.verts.end()) {
        MeshData::V nv; nv.p[0] = v[vin][0]; nv.p[1] = v[vin][1]; nv.p[2] = v[vin][2]; nv.has_n = false;
        nv.uv[0] = nv.uv[1] = 0.f;
        if (uvin >= 0 && uvin < (int)uv.size()) { nv.uv[0] = uv[uvin][0]; nv.uv[1] = uv[uvin][1]; }   // parsing.cu:106-107
        if (nin != -1) {
          if (nin < 0 || nin >= (int)n.size()) { err = "bad normal index in " + path; return false; }
          nv.n[0] = n[nin][0]; nv.n[1] = n[nin][1]; nv.n[2] = n[nin][2]; nv.has_n = true;
        }
        m.verts[vin] = nv;
      }
      fk[i] = vin;
    }
    m.faces.push_back(fk); m.fn.push_back({0, 0, 0});
    face_normal(m, m.faces.size() - 1);                 // Triangle ctor calls setN (triangle.cu:9)
    for (int i = 0; i < 3; i++) m.verts[fk[i]].adj.push_back((int)m.faces.size() - 1);
    next_line();
  }
  if (!n.size()) for (auto& kv : m.verts) vertex_normal(m, kv.second);   // parsing.cu:121
  return true;
}

void mesh_scale(MeshData& m, const std::vector<float>& s) {           // mesh.cu:22-26
  for (auto& kv : m.verts) for (int i = 0; i < 3; i++) kv.second.p[i] = kv.second.p[i] * s[i];
}
void mesh_translate(MeshData& m, const std::vector<float>& t) {       // mesh.cu:44-48
  for (auto& kv : m.verts) for (int i = 0; i < 3; i++) kv.second.p[i] = kv.second.p[i] + t[i];
}
inline void matmul3(const float r[3][3], float* v) {                  // vector.cu:35-41
  float o[3];
  for (int i = 0; i < 3; i++) o[i] = r[i][0] * v[0] + r[i][1] * v[1] + r[i][2] * v[2];
  v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
}
void mesh_rotate(MeshData& m, const std::vector<float>& rot) {        // mesh.cu:28-42
  float a = (float)(M_PI * rot[0] / 180), b = (float)(M_PI * rot[1] / 180), c = (float)(M_PI * rot[2] / 180);
  float rx[3][3] = {{1, 0, 0}, {0, cosf(a), -sinf(a)}, {0, sinf(a), cosf(a)}};
  for (auto& kv : m.verts) matmul3(rx, kv.second.p);
  float ry[3][3] = {{cosf(b), 0, sinf(b)}, {0, 1, 0}, {-sinf(b), 0, cosf(b)}};
  for (auto& kv : m.verts) matmul3(ry, kv.second.p);
  float rz[3][3] = {{cosf(c), -sinf(c), 0}, {sinf(c), cosf(c), 0}, {0, 0, 1}};
  for (auto& kv : m.verts) matmul3(rz, kv.second.p);
  for (size_t f = 0; f < m.faces.size(); f++) face_normal(m, f);
  for (auto& kv : m.verts) vertex_normal(m, kv.second);
}

// ------------------------------------------------------------ octree build
bool tri_overlaps_box(const Tri& t, const float* ini, const float* end) {   // triangle.cu:61-105
  static const float bn[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  float e01[3], e12[3], e20[3];
  for (int k = 0; k < 3; k++) {
    e01[k] = t.v[1][k] - t.v[0][k]; e12[k] = t.v[2][k] - t.v[1][k]; e20[k] = t.v[0][k] - t.v[2][k];
  }
  float axes[9][3];
  for (int i = 0; i < 3; i++) {
    hcross(axes[3 * i + 0], e01, bn[i]);
    hcross(axes[3 * i + 1], e12, bn[i]);
    hcross(axes[3 * i + 2], e20, bn[i]);
  }
  for (int i = 0; i < 9; i++) {
    const float* ax = axes[i];
    if (!hdot(ax, ax)) continue;
    float mint = kInf, maxt = -kInf, minb = kInf, maxb = -kInf, pr;
    for (int j = 0; j < 3; j++) {
      pr = hdot(t.v[j], ax);
      maxt = maxt < pr ? pr : maxt; mint = pr < mint ? pr : mint;
    }
    const float cs[8][3] = {
        {ini[0], ini[1], ini[2]}, {end[0], ini[1], ini[2]}, {ini[0], end[1], ini[2]}, {ini[0], ini[1], end[2]},
        {end[0], end[1], ini[2]}, {end[0], ini[1], end[2]}, {ini[0], end[1], end[2]}, {end[0], end[1], end[2]}};
    for (int c = 0; c < 8; c++) {
      pr = hdot(cs[c], ax);
      maxb = maxb < pr ? pr : maxb; minb = pr < minb ? pr : minb;
    }
    if (maxt < minb || mint > maxb) return false;
  }
  return true;
}

struct OctBuilder {
  Scene& sc; int max_depth_seen = 0; bool too_deep = false;
  explicit OctBuilder(Scene& s) : sc(s) {}
  int new_leaf(const float* ini, const float* end) {
    ONode n; n.leaf = true;
    for (int k = 0; k < 3; k++) { n.ini[k] = ini[k]; n.end[k] = end[k]; }
    for (int k = 0; k < 8; k++) n.child[k] = -1;
    sc.nodes.push_back(n); return (int)sc.nodes.size() - 1;
  }
  // octtree.cu:90-112: turn node `id` into a branch with eight empty leaves.
  void make_branch(int id) {
    float ini[3], end[3], mid[3];
    for (int k = 0; k < 3; k++) { ini[k] = sc.nodes[id].ini[k]; end[k] = sc.nodes[id].end[k]; mid[k] = (end[k] + ini[k]) / 2; }
    const float ci[8][3] = {{ini[0], ini[1], ini[2]}, {mid[0], ini[1], ini[2]}, {ini[0], mid[1], ini[2]}, {mid[0], mid[1], ini[2]},
                            {ini[0], ini[1], mid[2]}, {mid[0], ini[1], mid[2]}, {ini[0], mid[1], mid[2]}, {mid[0], mid[1], mid[2]}};
    const float ce[8][3] = {{mid[0], mid[1], mid[2]}, {end[0], mid[1], mid[2]}, {mid[0], end[1], mid[2]}, {end[0], end[1], mid[2]},
                            {mid[0], mid[1], end[2]}, {end[0], mid[1], end[2]}, {mid[0], end[1], end[2]}, {end[0], end[1], end[2]}};
    sc.nodes[id].leaf = false; sc.nodes[id].objs.clear(); sc.nodes[id].objs.shrink_to_fit();
    for (int c = 0; c < 8; c++) { int l = new_leaf(ci[c], ce[c]); sc.nodes[id].child[c] = l; }
  }
  void insert(int id, int tri, int depth) {                          // octtree.cu:14-20,162-174
    if (depth > max_depth_seen) max_depth_seen = depth;
    if (depth >= kOctStack) { too_deep = true; return; }
    if (!tri_overlaps_box(sc.tris[tri], sc.nodes[id].ini, sc.nodes[id].end)) return;
    if (!sc.nodes[id].leaf) {
      for (int c = 0; c < 8; c++) insert(sc.nodes[id].child[c], tri, depth + 1);
      return;
    }
    sc.nodes[id].objs.push_back(tri);
    if ((int)sc.nodes[id].objs.size() > kLeafCap) {
      std::vector<int> objs = sc.nodes[id].objs;
      make_branch(id);
      for (int o : objs) for (int c = 0; c < 8; c++) insert(sc.nodes[id].child[c], o, depth + 1);
    }
  }
};

// Texture images are handed to the checker already decoded (oracle_provide_texture: by another decoder than the
// library's, or by the library's when only the lookup is under test), keyed by the file_path written in the resources.
static std::map<std::string, Texture> g_provided_textures;

// -------------------------------------------------------------- parsing
bool load_scene(Scene& sc, const std::string& scene_path, const std::string& res_path, int W, int H, int depth,
                const std::string& base_dir) {
  JNode res, world;
  if (!read_json(res_path, res)) { sc.error = "cannot read " + res_path; return false; }
  if (!read_json(scene_path, world)) { sc.error = "cannot read " + scene_path; return false; }
  std::map<std::string, int> mat_by_name;
  std::map<std::string, std::string> tex_file;      // texture name -> file_path as written in the resources
  std::vector<std::string> tex_of;                  // per material: its texture's name, "" = none
  for (auto& rp : res.kids) {                                           // parsing.cu:181-269
    if (rp.first == "textures") {
      for (auto& te : rp.second.kids) {
        std::string name, file;
        for (auto& tp : te.second.kids) { if (tp.first == "name") name = tp.second.text; else if (tp.first == "file_path") file = tp.second.text; }
        if (name.size()) tex_file[name] = file;
      }
      continue;
    }
    if (rp.first != "materials") continue;
    for (auto& me : rp.second.kids) {
      std::vector<float> col, att; std::string type, name, texture, filtering; int shin = 0; float ior = 0; bool for_ambient = false;
      for (auto& mp : me.second.kids) {
        if (mp.first == "__type__") type = mp.second.text;
        else if (mp.first == "name") name = mp.second.text;
        else if (mp.first == "color") jvec(mp.second, col);
        else if (mp.first == "brdfParams") { for (auto& bp : mp.second.kids) if (bp.first == "shininess") shin = atoi(bp.second.text.c_str()); }
        else if (mp.first == "attenuation") jvec(mp.second, att);
        else if (mp.first == "refraction_index") ior = jf(mp.second);
        else if (mp.first == "brdf") type = mp.second.text;
        else if (mp.first == "color_texture") texture = mp.second.text;
        else if (mp.first == "texture_filtering") filtering = mp.second.text;
        else if (mp.first == "use_for_ambient") for_ambient = mp.second.text == "true";
      }
      Material m; memset(&m, 0, sizeof m); m.tex = -1;
      if (type == "lambert") m.type = LAMBERT;
      else if (type == "blinnPhong") { m.type = PHONG; m.param = (float)shin; }
      else if (type == "reflective_material") m.type = REFLECTIVE;
      else if (type == "dielectric_material") { m.type = DIELECTRIC; m.param = ior; }
      else continue;
      if (texture.size() && (m.type == LAMBERT || m.type == PHONG)) {
        // `brdf_color_texture_material`: no behaviour in the reference (its TextureMaterial is a stub, material.cu:263-301).
        // Defined here, for the checker's sake, exactly as the library defines it: see texture_colour below.
        m.tex = -2; m.bilinear = filtering == "bilinear"; m.for_ambient = for_ambient;   // (-2: resolved once a scene uses it)
        mat_by_name[name] = (int)sc.mats.size(); sc.mats.push_back(m); tex_of.push_back(texture);
        continue;
      }
      if (col.size() < 3) continue;   // other entries without a colour: the reference reads out of bounds here
      for (int k = 0; k < 3; k++) m.col[k] = col[k];
      if (m.type == DIELECTRIC) { if (att.size() < 3) continue; for (int k = 0; k < 3; k++) m.att[k] = att[k]; }
      mat_by_name[name] = (int)sc.mats.size(); sc.mats.push_back(m); tex_of.push_back("");
    }
  }
  std::vector<float> e, up, target; float fov = 0; bool have_cam = false;
  float bg[3] = {0, 0, 0}; float scene_refr = 1.0f; bool shadows = true;
  for (auto& wp : world.kids) {                                         // parsing.cu:275-470
    if (wp.first == "camera") {
      for (auto& cp : wp.second.kids) {
        if (cp.first == "fov") fov = (float)(M_PI * jf(cp.second) / 180);
        else if (cp.first == "position") jvec(cp.second, e);
        else if (cp.first == "up") jvec(cp.second, up);
        else if (cp.first == "target") jvec(cp.second, target);
      }
      have_cam = true;
    } else if (wp.first == "objects") {
      for (auto& ob : wp.second.kids) {
        std::vector<float> pos, sca, rot, trans; float radius = 0; std::vector<int> mats; std::string type, file; bool bad_mat = false;
        for (auto& op : ob.second.kids) {
          if (op.first == "__type__") type = op.second.text;
          else if (op.first == "file_path") file = op.second.text;
          else if (op.first == "scaling") jvec(op.second, sca);
          else if (op.first == "rotation") jvec(op.second, rot);
          else if (op.first == "translation") jvec(op.second, trans);
          else if (op.first == "radius") radius = jf(op.second);
          else if (op.first == "position") jvec(op.second, pos);
          else if (op.first == "materials") {
            for (auto& mn : op.second.kids) {
              auto it = mat_by_name.find(mn.second.text);
              if (it == mat_by_name.end()) bad_mat = true; else mats.push_back(it->second);
            }
          }
        }
        if (bad_mat) { sc.error = "unknown material in " + scene_path; return false; }
        if (type == "sphere") {
          if (pos.size() < 3) { sc.error = "sphere without position"; return false; }
          Sphere s; s.c[0] = pos[0]; s.c[1] = pos[1]; s.c[2] = pos[2]; s.r = radius; s.mats = mats;
          sc.spheres.push_back(s);
        } else if (type == "mesh") {
          MeshData md; std::string err;
          std::string p = (file.size() && file[0] == '/') ? file : base_dir + file;
          if (!load_obj(p, md, err)) { sc.error = err; return false; }
          if (sca.size()) mesh_scale(md, sca);
          if (rot.size()) mesh_rotate(md, rot);
          if (trans.size()) mesh_translate(md, trans);
          int mesh_id = (int)sc.mesh_mats.size(); sc.mesh_mats.push_back(mats);
          for (size_t f = 0; f < md.faces.size(); f++) {                // mesh.cu:65-97
            Tri t; t.mesh = mesh_id;
            for (int j = 0; j < 3; j++) {
              const MeshData::V& v = md.verts[md.faces[f][j]];
              for (int k = 0; k < 3; k++) { t.v[j][k] = v.p[k]; t.n[j][k] = v.has_n ? v.n[k] : 0.0f; }
              t.uv[j][0] = v.uv[0]; t.uv[j][1] = v.uv[1];
            }
            sc.tris.push_back(t);
          }
        }
      }
    } else if (wp.first == "lights") {
      for (auto& li : wp.second.kids) {
        std::string type; float angle = 0; std::vector<float> pos, col, dir;
        for (auto& lp : li.second.kids) {
          if (lp.first == "__type__") type = lp.second.text;
          else if (lp.first == "position") jvec(lp.second, pos);
          else if (lp.first == "color") jvec(lp.second, col);
          else if (lp.first == "direction") jvec(lp.second, dir);
          else if (lp.first == "angle") angle = jf(lp.second);
        }
        Light L; memset(&L, 0, sizeof L);
        if (col.size() < 3) continue;
        for (int k = 0; k < 3; k++) L.col[k] = col[k];
        auto need = [&](std::vector<float>& v) { return v.size() >= 3; };
        if (type == "spot_light" && need(pos) && need(dir)) {
          L.type = SPOT; L.angle = (float)(M_PI * angle / 180);
          float d[3] = {dir[0], dir[1], dir[2]}; hnormalize(d);
          for (int k = 0; k < 3; k++) { L.pos[k] = pos[k]; L.dir[k] = d[k]; }
        } else if (type == "point_light" && need(pos)) {
          L.type = PUNCTUAL; for (int k = 0; k < 3; k++) L.pos[k] = pos[k];
        } else if (type == "directional_light" && need(dir)) {
          L.type = DIRECTIONAL;
          float d[3] = {dir[0], dir[1], dir[2]}; hnormalize(d);   // parsing.cu:437
          hnormalize(d);                                           // lights.cu:27 normalises again
          for (int k = 0; k < 3; k++) L.dir[k] = d[k];
        } else if (type == "ambient_light") L.type = AMBIENT;
        else continue;
        sc.lights.push_back(L);
      }
    } else if (wp.first == "params") {
      for (auto& pp : wp.second.kids) {
        if (pp.first == "background_color") { int i = 0; for (auto& c : pp.second.kids) if (i < 3) bg[i++] = jf(c.second); }
        else if (pp.first == "enable_shadows") shadows = (pp.second.text == "true");
        else if (pp.first == "refraction_index") scene_refr = jf(pp.second);
      }
    }
  }
  {   // textures of the materials some object uses
    std::map<std::string, int> loaded;
    auto use = [&](int mi) -> bool {
      if (sc.mats[mi].tex != -2) return true;
      const std::string& tname = tex_of[mi];
      auto it = loaded.find(tname);
      if (it == loaded.end()) {
        auto tf = tex_file.find(tname);
        if (tf == tex_file.end()) { sc.error = "unknown texture " + tname; return false; }
        auto pv = g_provided_textures.find(tf->second);
        if (pv == g_provided_textures.end()) { sc.error = "texture " + tf->second + " was not provided to the checker"; return false; }
        it = loaded.emplace(tname, (int)sc.textures.size()).first;
        sc.textures.push_back(pv->second);
      }
      sc.mats[mi].tex = it->second;
      return true;
    };
    for (auto& sp : sc.spheres) for (int mi : sp.mats) if (!use(mi)) return false;
    for (auto& mm : sc.mesh_mats) for (int mi : mm) if (!use(mi)) return false;
  }
  if (!have_cam || e.size() < 3 || up.size() < 3 || target.size() < 3) { sc.error = "scene has no camera"; return false; }
  // camera.cu:45-68
  Camera& c = sc.cam;
  float w[3] = {e[0] - target[0], e[1] - target[1], e[2] - target[2]}; hnormalize(w);
  float upv[3] = {up[0], up[1], up[2]};
  float u[3]; hcross(u, upv, w); hnormalize(u);
  float v[3]; hcross(v, w, u);
  float top = kNear * tanf(fov / 2);
  float right = (float)W * top / (float)H;
  for (int k = 0; k < 3; k++) { c.u[k] = u[k]; c.v[k] = v[k]; c.w[k] = w[k]; c.pos[k] = e[k]; c.bg[k] = bg[k]; }
  c.top = top; c.bottom = -top; c.right = right; c.left = -right;
  c.W = W; c.H = H; c.depth = depth; c.shadows = shadows; c.scene_refr = scene_refr;
  // main.cu:25-31
  if (!sc.mesh_mats.empty()) {
    OctBuilder ob(sc);
    const float ini[3] = {-kInf / 4, -kInf / 4, kInf / 4}, end[3] = {kInf / 4, kInf / 4, -kInf / 4};
    int root = ob.new_leaf(ini, end); ob.make_branch(root);
    for (int t = 0; t < (int)sc.tris.size(); t++) {
      // OctBranch::insert on the root (octtree.cu:162-166)
      if (!tri_overlaps_box(sc.tris[t], sc.nodes[0].ini, sc.nodes[0].end)) continue;
      for (int ch = 0; ch < 8; ch++) ob.insert(sc.nodes[0].child[ch], t, 1);
    }
    if (ob.too_deep) { sc.error = "octree deeper than the 20-entry traversal stack"; return false; }
    sc.has_oct = true;
  }
  return true;
}

// ------------------------------------------------------------- rendering
struct Ray {                    // ray.cuh:9-13
  float eye[3], dir[3], t, ip[3], n[3], col[3], att[3], level[3];
  int sph, tri; bool oct; int depth, nlevel;
  float uv[2];   // texture coordinates of the hit (texture materials; not in the reference)
};
struct RayStack { Ray r[kRayStack]; int size; };

struct Tracer {
  const Scene& sc; Counters& ct;
  Tracer(const Scene& s, Counters& c) : sc(s), ct(c) {}

  // triangle.cu:107-152
  bool hit_triangle(int id, Ray& ray, bool record) const {
    const Tri& t = sc.tris[id];
    const float *v0 = t.v[0], *v1 = t.v[1], *v2 = t.v[2], *e = ray.eye, *d = ray.dir;
    float b[3] = {e[0] - v0[0], e[1] - v0[1], e[2] - v0[2]};
    float c0[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    float c1[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    float c2[3] = {-d[0], -d[1], -d[2]};
    float x[3]; cofactor_solve(c0, c1, c2, b, x);
    float bet = x[0], gam = x[1], alp = 1 - bet - gam, t1 = x[2];
    if (0 < t1 && t1 < ray.t && 0 <= bet && bet <= 1 && 0 <= gam && gam <= 1 && 0 <= alp) {
      ray.t = t1;
      if (record) {
        for (int k = 0; k < 3; k++) {
          float wn0 = t.n[0][k] * alp, wn1 = t.n[1][k] * bet, wn2 = t.n[2][k] * gam;
          ray.n[k] = (wn0 + wn1) + wn2;
        }
        dnormalize(ray.n);
        for (int k = 0; k < 3; k++) {
          float wv0 = v0[k] * alp, wv1 = v1[k] * bet, wv2 = v2[k] * gam;
          ray.ip[k] = (wv0 + wv1) + wv2;
        }
        for (int k = 0; k < 2; k++) ray.uv[k] = (t.uv[0][k] * alp + t.uv[1][k] * bet) + t.uv[2][k] * gam;
        ray.tri = id; ray.sph = -1;
      }
      return true;
    }
    return false;
  }

  // octtree.cu:276-321: true iff the node would be pushed.
  bool node_accept(const ONode& nd, const Ray& ray) const {
    const float *ini = nd.ini, *end = nd.end, *d = ray.dir, *e = ray.eye;
    const float v0[6][3] = {{end[0], ini[1], ini[2]}, {ini[0], ini[1], ini[2]}, {ini[0], ini[1], end[2]},
                            {end[0], ini[1], end[2]}, {ini[0], end[1], end[2]}, {ini[0], ini[1], ini[2]}};
    const float v1[6][3] = {{ini[0], ini[1], ini[2]}, {ini[0], ini[1], end[2]}, {end[0], ini[1], end[2]},
                            {end[0], ini[1], ini[2]}, {end[0], end[1], end[2]}, {end[0], ini[1], ini[2]}};
    const float v3[6][3] = {{end[0], end[1], ini[2]}, {ini[0], end[1], ini[2]}, {ini[0], end[1], end[2]},
                            {end[0], end[1], end[2]}, {ini[0], end[1], ini[2]}, {ini[0], ini[1], end[2]}};
    for (int i = 0; i < 6; i++) {
      float c0[3] = {v1[i][0] - v0[i][0], v1[i][1] - v0[i][1], v1[i][2] - v0[i][2]};
      float c1[3] = {v3[i][0] - v0[i][0], v3[i][1] - v0[i][1], v3[i][2] - v0[i][2]};
      float c2[3] = {-d[0], -d[1], -d[2]};
      float b[3] = {e[0] - v0[i][0], e[1] - v0[i][1], e[2] - v0[i][2]};
      float x[3]; cofactor_solve(c0, c1, c2, b, x);
      float alp = x[0], bet = x[1], t1 = x[2];
      if (0 < t1 && t1 < ray.t && 0 <= bet && bet <= 1 && 0 <= alp && alp <= 1) return true;
    }
    return false;
  }

  // spheres.cu:73-100
  bool hit_sphere(int id, Ray& ray, bool record) const {
    const Sphere& s = sc.spheres[id];
    float ec[3] = {ray.eye[0] - s.c[0], ray.eye[1] - s.c[1], ray.eye[2] - s.c[2]};
    float B = 2 * ddot(ec, ray.dir);
    float C = ddot(ec, ec) - s.r * s.r;
    float det = B * B - 4 * C;
    if (det >= 0) {
      det = sqrtf(det);
      float t2 = -B + det, t1 = -B - det;
      if (t2 >= 0) {
        if (t1 > 0 && t1 < 2 * ray.t) { t1 /= 2; ray.t = t1; }
        else if (t1 < 0 && t2 < 2 * ray.t) { t2 /= 2; ray.t = t2; }
        else return false;
        if (record) {
          for (int k = 0; k < 3; k++) ray.ip[k] = ray.eye[k] + ray.dir[k] * ray.t;
          for (int k = 0; k < 3; k++) ray.n[k] = ray.ip[k] - s.c[k];
          dnormalize(ray.n);
          ray.uv[0] = 0.5f + atan2f(ray.n[2], ray.n[0]) * 0.15915494f;   // longitude and latitude of the unit normal
          ray.uv[1] = 0.5f + asinf(ray.n[1] < -1.f ? -1.f : (ray.n[1] > 1.f ? 1.f : ray.n[1])) * 0.31830987f;
          ray.sph = id; ray.tri = -1;
        }
        return true;
      }
    }
    return false;
  }

  // ray.cu:29-54
  void intersect(Ray& ray, bool record) const {
    ct.rays++;
    int onode[kOctStack], ochild[kOctStack], osize = 0;
    if (ray.oct) { onode[0] = 0; ochild[0] = 0; osize = 1; }
    while (osize) {
      int child = ochild[osize - 1];
      int id = onode[--osize];
      const ONode& nd = sc.nodes[id];
      if (nd.leaf) {
        ct.leaf_visits++;
        for (int tid : nd.objs) { ct.tri_tests++; hit_triangle(tid, ray, record); }
      } else if (child < 8) {
        onode[osize] = id; ochild[osize] = child + 1; osize++;
        ct.node_tests++;
        int cid = nd.child[child];
        if (node_accept(sc.nodes[cid], ray)) { onode[osize] = cid; ochild[osize] = 0; osize++; }
      }
    }
    for (int is = 0; is < (int)sc.spheres.size(); is++) { ct.sphere_tests++; hit_sphere(is, ray, record); }
  }

  // ray.cu:56-72
  bool shadowed(const float* pos, const float* dir, bool oct, float ldist) const {
    Ray r; memset(&r, 0, sizeof r);
    for (int k = 0; k < 3; k++) { r.eye[k] = pos[k]; r.dir[k] = dir[k]; }
    r.sph = -1; r.tri = -1; r.t = kInf; r.oct = oct;
    ct.shadow_rays++;
    intersect(r, false);
    return ldist > r.t;
  }

  void push(RayStack& st, const float* pos, const float* dir, const float* level, int nlevel, int depth,
            const float* att, bool oct) const {                     // vector.cu:188-207
    if (st.size >= kRayStack) { ct.ub_stack_overflow++; return; }
    Ray& r = st.r[st.size];
    for (int k = 0; k < 3; k++) { r.eye[k] = pos[k]; r.dir[k] = dir[k]; r.att[k] = att[k]; r.level[k] = level[k]; r.col[k] = 0; }
    r.sph = -1; r.tri = -1; r.t = kInf; r.nlevel = nlevel; r.depth = depth; r.oct = oct;
    st.size++;
  }

  // Light vector, shadow origin and light distance shared by lights.cu:268-311 and :322-365.
  // Returns false when the light contributes nothing to this point.
  bool light_setup(const Light& li, const Ray& ray, float* l) const {
    float sP[3], ldist;
    if (li.type == PUNCTUAL || li.type == SPOT) {
      for (int k = 0; k < 3; k++) l[k] = li.pos[k] - ray.ip[k];
      dnormalize(l);
      for (int k = 0; k < 3; k++) sP[k] = ray.ip[k] + l[k] * kEps;
      float dt[3] = {li.pos[0] - sP[0], li.pos[1] - sP[1], li.pos[2] - sP[2]};
      ldist = dlen(dt);
    } else {
      for (int k = 0; k < 3; k++) l[k] = li.dir[k] * -1;
      for (int k = 0; k < 3; k++) sP[k] = ray.ip[k] + l[k] * kEps;
      ldist = kInf;
    }
    if (sc.cam.shadows && shadowed(sP, l, ray.oct, ldist)) return false;
    if (li.type == SPOT && !(-ddot(l, li.dir) > cosf(li.angle / 2))) return false;
    return true;
  }

  // Colour of texture `T` at (u, v): u and v repeat, v = 1 is the top row; nearest texel, or the four texels around
  // (u W - 0.5, (1 - v) H - 0.5) blended along x, then along y; a channel is its byte / 255.
  static void texture_colour(const Texture& T, bool bilinear, float u, float v, float* out) {
    const int W = T.w, H = T.h;
    auto texel = [&](int x, int y, float* c) { const uint8_t* p = &T.rgb[3 * ((size_t)y * W + x)]; for (int k = 0; k < 3; k++) c[k] = (float)p[k] / 255.f; };
    if (!(u - u == 0.f) || !(v - v == 0.f)) { u = 0.f; v = 0.f; }
    u = u - floorf(u); v = v - floorf(v);
    const float fx = u * (float)W, fy = (1.f - v) * (float)H;
    if (!bilinear) {
      int x = (int)fx, y = (int)fy;
      x = x < W ? x : W - 1; y = y < H ? y : H - 1;
      texel(x, y, out);
      return;
    }
    const float gx = fx - 0.5f, gy = fy - 0.5f;
    const float x0f = floorf(gx), y0f = floorf(gy);
    const float tx = gx - x0f, ty = gy - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? x0 + W : x0; y0 = y0 < 0 ? y0 + H : y0;
    x1 = x1 >= W ? x1 - W : x1; y1 = y1 >= H ? y1 - H : y1;
    x0 = x0 < W ? x0 : W - 1; y0 = y0 < H ? y0 : H - 1;
    float c00[3], c10[3], c01[3], c11[3];
    texel(x0, y0, c00); texel(x1, y0, c10); texel(x0, y1, c01); texel(x1, y1, c11);
    const float sx = 1.f - tx, sy = 1.f - ty;
    for (int k = 0; k < 3; k++) {
      const float top = c00[k] * sx + c10[k] * tx, bot = c01[k] * sx + c11[k] * tx;
      out[k] = top * sy + bot * ty;
    }
  }

  void lambert(const Light& li, const Material& m, Ray& ray) const {   // lights.cu:260-313
    if (li.type == AMBIENT) {
      if (m.tex >= 0 && !m.for_ambient) return;
      for (int k = 0; k < 3; k++) ray.col[k] = ray.col[k] + m.col[k] * li.col[k];
      return;
    }
    float l[3];
    if (!light_setup(li, ray, l)) return;
    float lam = fmaxr(ddot(ray.n, l), 0.f);
    for (int k = 0; k < 3; k++) ray.col[k] = ray.col[k] + (m.col[k] * li.col[k]) * lam;
  }

  void phong(const Light& li, const Material& m, Ray& ray) const {     // lights.cu:314-367
    if (li.type == AMBIENT) return;
    float l[3];
    if (!light_setup(li, ray, l)) return;
    float h[3] = {ray.dir[0] * -1 + l[0], ray.dir[1] * -1 + l[1], ray.dir[2] * -1 + l[2]};
    dnormalize(h);
    float ph = powf(fmaxr(ddot(ray.n, h), 0.f), m.param);
    for (int k = 0; k < 3; k++) ray.col[k] = ray.col[k] + (m.col[k] * li.col[k]) * ph;
  }

  static void refract(float* rr, const float* d, const float* n, float rori) {   // vector.cu:147-156
    float costhe = -ddot(d, n), sinthe = sqrtf(1 - powf(costhe, 2)), sinfi = rori * sinthe,
          cosfi = sqrtf(1 - powf(sinfi, 2));
    for (int k = 0; k < 3; k++) {
      float N1 = n[k] * costhe; N1 = N1 + d[k];
      float b = N1 / sinthe; b = b * sinfi;
      float N2 = n[k] * cosfi;
      rr[k] = b - N2;
    }
  }

  void colorate(RayStack& st, const Material& m, Ray& ray) const {     // material.cu:10-101
    switch (m.type) {
      case LAMBERT:
      case PHONG: {
        Material mm = m;
        if (m.tex >= 0) texture_colour(sc.textures[m.tex], m.bilinear, ray.uv[0], ray.uv[1], mm.col);
        for (const Light& li : sc.lights) { if (mm.type == LAMBERT) lambert(li, mm, ray); else phong(li, mm, ray); }
        return;
      }
      case REFLECTIVE:
        if (ray.depth < sc.cam.depth) {
          float s = 2 * ddot(ray.dir, ray.n), r[3], sP[3], att[3];
          for (int k = 0; k < 3; k++) r[k] = ray.dir[k] - ray.n[k] * s;
          dnormalize(r);
          for (int k = 0; k < 3; k++) { sP[k] = ray.ip[k] + r[k] * kEps; att[k] = m.col[k] * ray.att[k]; }
          push(st, sP, r, ray.level, ray.nlevel, ray.depth + 1, att, ray.oct);
        }
        return;
      case DIELECTRIC:
        if (ray.depth < sc.cam.depth) {
          int lv = ray.nlevel < 0 ? 0 : (ray.nlevel > 2 ? 2 : ray.nlevel);
          float refr = ray.level[lv];
          float s = 2 * ddot(ray.dir, ray.n), rl[3], rlsP[3], rf[3], rfsP[3], attl[3], attf[3], K[3];
          for (int k = 0; k < 3; k++) rl[k] = ray.dir[k] - ray.n[k] * s;
          dnormalize(rl);
          for (int k = 0; k < 3; k++) rlsP[k] = ray.ip[k] + ray.n[k] * kEps;
          float R0 = powf((refr - m.param) / (refr + m.param), 2);
          bool entering = ddot(ray.dir, ray.n) < 0;
          float rOrI, cs, nn[3];
          if (entering) {
            rOrI = refr / m.param; cs = -ddot(ray.dir, ray.n);
            K[0] = K[1] = K[2] = 1.0f;
            for (int k = 0; k < 3; k++) nn[k] = ray.n[k];
          } else {
            rOrI = m.param / refr; cs = ddot(ray.dir, ray.n);
            for (int k = 0; k < 3; k++) { nn[k] = ray.n[k] * -1; K[k] = expf(m.att[k] * -ray.t); }
          }
          float totRefl = 1 - powf(rOrI, 2) * (1 - powf(cs, 2));
          for (int k = 0; k < 3; k++) { attl[k] = (m.col[k] * K[k]) * ray.att[k]; attf[k] = (m.col[k] * K[k]) * ray.att[k]; }
          if (totRefl >= 0) {
            refract(rf, ray.dir, nn, rOrI);
            for (int k = 0; k < 3; k++) rfsP[k] = ray.ip[k] + rf[k] * kEps;
            float fresnel = (float)(R0 + (1 - R0) * pow(1.0 - cs, 5));
            for (int k = 0; k < 3; k++) attf[k] = attf[k] * (1 - fresnel);
            if (entering) {
              int nl = ray.nlevel + 1;
              if (nl > 2) { nl = 2; ct.ub_level_overflow++; }
              ray.level[nl] = m.param;
              push(st, rfsP, rf, ray.level, nl, ray.depth + 1, attf, ray.oct);
            } else {
              int nl = ray.nlevel - 1;
              if (nl < 0) { nl = 0; ct.ub_leave_level0++; }
              push(st, rfsP, rf, ray.level, nl, ray.depth + 1, attf, ray.oct);
            }
            for (int k = 0; k < 3; k++) attl[k] = attl[k] * fresnel;
            push(st, rlsP, rl, ray.level, ray.nlevel, ray.depth + 1, attl, ray.oct);
          } else {
            push(st, rlsP, rl, ray.level, ray.nlevel, ray.depth + 1, attl, ray.oct);
          }
        }
        return;
    }
  }

  void shade(RayStack& st, Ray& ray) const {                            // ray.cu:74-87
    if (ray.sph >= 0) { for (int mi : sc.spheres[ray.sph].mats) colorate(st, sc.mats[mi], ray); }
    else if (ray.tri >= 0) { for (int mi : sc.mesh_mats[sc.tris[ray.tri].mesh]) colorate(st, sc.mats[mi], ray); }
    else { for (int k = 0; k < 3; k++) ray.col[k] = sc.cam.bg[k]; }
  }

  // camera.cu:94-122 for one pixel (i = column, j = row counted from the bottom).
  void pixel(int i, int j, uint8_t* rgb, int32_t* id_out, float* t_out) const {
    const Camera& c = sc.cam;
    float ic = (float)(c.left + ((c.right - c.left) / (float)c.W) * (i + 0.5));
    float jc = (float)(c.bottom + ((c.top - c.bottom) / (float)c.H) * (j + 0.5));
    float co[3] = {ic, jc, -kNear}, d[3];
    for (int k = 0; k < 3; k++) d[k] = co[0] * c.u[k] + co[1] * c.v[k] + co[2] * c.w[k];
    dnormalize(d);
    float color[3] = {0, 0, 0}, att[3] = {1, 1, 1}, level[3] = {c.scene_refr, 0, 0};
    RayStack st; st.size = 0;
    push(st, c.pos, d, level, 0, 0, att, sc.has_oct);
    bool first = true;
    while (st.size) {
      Ray ray = st.r[--st.size];                                        // copyRay (vector.cu:170-185)
      intersect(ray, true);
      if (first) {
        first = false;
        if (id_out) *id_out = ray.sph >= 0 ? -2 - ray.sph : (ray.tri >= 0 ? ray.tri : -1);
        if (t_out) *t_out = ray.t;
      }
      shade(st, ray);
      for (int k = 0; k < 3; k++) color[k] = color[k] + ray.att[k] * ray.col[k];
    }
    for (int k = 0; k < 3; k++) {
      float v = fminr(color[k], 1.f) * 255.f;
      rgb[k] = v < 0 ? (uint8_t)0 : (uint8_t)v;
    }
  }
};

}  // namespace

// ================================================================== C API
extern "C" {

struct oracle_counters {
  uint64_t rays, shadow_rays, node_tests, leaf_visits, tri_tests, sphere_tests;
  uint64_t ub_leave_level0, ub_level_overflow, ub_stack_overflow;
};
struct oracle_info {
  int32_t width, height, depth, n_tris, n_spheres, n_lights, n_mats, n_nodes, n_leaves, n_empty_leaves, n_refs, max_depth,
      max_leaf;
};

static std::string g_err;
const char* oracle_last_error() { return g_err.c_str(); }

// Decoded pixels (RGB8, row 0 on top) of the texture the resources call `file_path`, for scenes loaded afterwards.
void oracle_provide_texture(const char* file_path, int w, int h, const uint8_t* rgb) {
  Texture t; t.w = w; t.h = h; t.rgb.assign(rgb, rgb + (size_t)w * h * 3);
  g_provided_textures[file_path] = t;
}
void oracle_clear_textures() { g_provided_textures.clear(); }

// base_dir: prefix for relative mesh paths ("" = cwd, as the reference resolves them).
void* oracle_scene_load(const char* scene, const char* resources, int W, int H, int depth, const char* base_dir) {
  auto* sc = new Scene();
  std::string bd = base_dir ? base_dir : "";
  if (bd.size() && bd.back() != '/') bd += '/';
  if (!load_scene(*sc, scene, resources, W, H, depth, bd)) { g_err = sc->error; delete sc; return nullptr; }
  return sc;
}
void oracle_scene_free(void* h) { delete (Scene*)h; }

static void tree_stats(const Scene& sc, int id, int depth, oracle_info& o) {
  const ONode& n = sc.nodes[id];
  if (depth > o.max_depth) o.max_depth = depth;
  if (n.leaf) {
    o.n_leaves++; if (n.objs.empty()) o.n_empty_leaves++;
    o.n_refs += (int)n.objs.size(); if ((int)n.objs.size() > o.max_leaf) o.max_leaf = (int)n.objs.size();
  } else for (int c = 0; c < 8; c++) tree_stats(sc, n.child[c], depth + 1, o);
}
void oracle_scene_info(void* h, oracle_info* o) {
  const Scene& sc = *(Scene*)h; memset(o, 0, sizeof *o);
  o->width = sc.cam.W; o->height = sc.cam.H; o->depth = sc.cam.depth;
  o->n_tris = (int)sc.tris.size(); o->n_spheres = (int)sc.spheres.size(); o->n_lights = (int)sc.lights.size();
  o->n_mats = (int)sc.mats.size(); o->n_nodes = (int)sc.nodes.size();
  if (sc.has_oct) tree_stats(sc, 0, 0, *o);
}

// DFS-preorder dump (children 0..7), the order OctBranch::buildDNodes walks (octtree.cu:64-87).
// kind[i] = -1 for a branch, else the leaf's triangle count; bounds[i] = ini xyz, end xyz;
// refs = concatenated leaf contents in that order.  Pass NULLs to size the buffers from oracle_scene_info.
static void dump_rec(const Scene& sc, int id, int32_t* kind, float* bounds, int32_t* refs, int& ni, int& ri) {
  const ONode& n = sc.nodes[id];
  for (int k = 0; k < 3; k++) { bounds[6 * ni + k] = n.ini[k]; bounds[6 * ni + 3 + k] = n.end[k]; }
  if (n.leaf) {
    kind[ni++] = (int32_t)n.objs.size();
    for (int t : n.objs) refs[ri++] = t;
  } else {
    kind[ni++] = -1;
    for (int c = 0; c < 8; c++) dump_rec(sc, n.child[c], kind, bounds, refs, ni, ri);
  }
}
void oracle_dump_tree(void* h, int32_t* kind, float* bounds, int32_t* refs) {
  const Scene& sc = *(Scene*)h; if (!sc.has_oct) return;
  int ni = 0, ri = 0; dump_rec(sc, 0, kind, bounds, refs, ni, ri);
}
// verts: n_tris*9 positions, normals: n_tris*9, mesh: n_tris.
void oracle_dump_tris(void* h, float* verts, float* normals, int32_t* mesh) {
  const Scene& sc = *(Scene*)h;
  for (size_t i = 0; i < sc.tris.size(); i++) {
    memcpy(verts + 9 * i, sc.tris[i].v, 36); memcpy(normals + 9 * i, sc.tris[i].n, 36); mesh[i] = sc.tris[i].mesh;
  }
}
// 19 floats: u v w pos (12), left right bottom top (4), bg (3)
void oracle_dump_camera(void* h, float* out) {
  const Camera& c = ((Scene*)h)->cam;
  memcpy(out, c.u, 12); memcpy(out + 3, c.v, 12); memcpy(out + 6, c.w, 12); memcpy(out + 9, c.pos, 12);
  out[12] = c.left; out[13] = c.right; out[14] = c.bottom; out[15] = c.top; memcpy(out + 16, c.bg, 12);
}

// Renders image rows [row_begin,row_end) (row 0 = top).  rgb: H*W*3 full-image buffer; ids/ts: H*W or NULL.
// threads > 1 splits rows over OpenMP threads (counters are summed).
void oracle_render_rows(void* h, int row_begin, int row_end, uint8_t* rgb, int32_t* ids, float* ts,
                        oracle_counters* out, int threads) {
  const Scene& sc = *(Scene*)h; const int W = sc.cam.W, H = sc.cam.H;
  Counters total; memset(&total, 0, sizeof total);
  if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
  {
    Counters ct; memset(&ct, 0, sizeof ct);
    Tracer tr(sc, ct);
#pragma omp for schedule(dynamic, 1)
    for (int row = row_begin; row < row_end; row++) {
      int j = H - 1 - row;
      for (int i = 0; i < W; i++) {
        size_t p = (size_t)W * row + i;
        tr.pixel(i, j, rgb + 3 * p, ids ? ids + p : nullptr, ts ? ts + p : nullptr);
      }
    }
#pragma omp critical
    {
      uint64_t* a = (uint64_t*)&total; const uint64_t* b = (const uint64_t*)&ct;
      for (size_t k = 0; k < sizeof(Counters) / 8; k++) a[k] += b[k];
    }
  }
  if (out) memcpy(out, &total, sizeof total);
}

// ---- per-routine entry points for known-answer tests (one call = one reference routine) ----
// triangle.cu:107-152. tri: 9 positions + 9 normals. ray_io: eye3 dir3 t. out: hit flag, t, ip3, n3.
int oracle_intersect_triangle(const float* tri, const float* ray, float* out) {
  Scene sc; Tri t; memset(&t, 0, sizeof t); memcpy(t.v, tri, 36); memcpy(t.n, tri + 9, 36); t.mesh = 0; sc.tris.push_back(t);
  Counters ct; Tracer tr(sc, ct); Ray r; memset(&r, 0, sizeof r);
  memcpy(r.eye, ray, 12); memcpy(r.dir, ray + 3, 12); r.t = ray[6]; r.tri = -1; r.sph = -1;
  bool hit = tr.hit_triangle(0, r, true);
  out[0] = r.t; memcpy(out + 1, r.ip, 12); memcpy(out + 4, r.n, 12);
  return hit;
}
// spheres.cu:73-100. sph: c3 r.
int oracle_intersect_sphere(const float* sph, const float* ray, float* out) {
  Scene sc; Sphere s; memcpy(s.c, sph, 12); s.r = sph[3]; sc.spheres.push_back(s);
  Counters ct; Tracer tr(sc, ct); Ray r; memset(&r, 0, sizeof r);
  memcpy(r.eye, ray, 12); memcpy(r.dir, ray + 3, 12); r.t = ray[6]; r.tri = -1; r.sph = -1;
  bool hit = tr.hit_sphere(0, r, true);
  out[0] = r.t; memcpy(out + 1, r.ip, 12); memcpy(out + 4, r.n, 12);
  return hit;
}
// octtree.cu:276-321. box: ini3 end3.
int oracle_node_accept(const float* box, const float* ray) {
  Scene sc; Counters ct; Tracer tr(sc, ct); ONode n; memcpy(n.ini, box, 12); memcpy(n.end, box + 3, 12);
  Ray r; memset(&r, 0, sizeof r); memcpy(r.eye, ray, 12); memcpy(r.dir, ray + 3, 12); r.t = ray[6];
  return tr.node_accept(n, r);
}
// triangle.cu:61-105.
int oracle_tri_overlaps_box(const float* verts, const float* box) {
  Tri t; memset(&t, 0, sizeof t); memcpy(t.v, verts, 36); return tri_overlaps_box(t, box, box + 3);
}
// vector.cu:147-156.
void oracle_refract(const float* d, const float* n, float rori, float* out) { Tracer::refract(out, d, n, rori); }

}  // extern "C"
