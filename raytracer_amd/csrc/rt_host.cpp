// rt_host.cpp -- host side of the render path: scene + resource parsing, OBJ ingest, mesh
// transforms, octree construction straight into the flat HBM layout, PNG output.
//
// Everything here feeds the HIP kernels with bit-identical inputs to the ones the reference's
// host code feeds its kernel, so the float arithmetic follows the reference operation by operation
// (camera/camera.cu:45-74, mesh/mesh.cu:22-48, vertex/vertex.cu:8-12, triangle/triangle.cu:11-14,61-105,
// vector/vector.cu:35-63, octtree/octtree.cu:14-20,90-112,162-174, parsing/parsing.cu:21-130,145-473).
// Build with -ffp-contract=off.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include <zlib.h>

#include "rt_internal.h"
#include "rt_math.h"

// ------------------------------------------------------------------ errors
static thread_local std::string g_last_error;
void rt_set_error(const std::string& msg) { g_last_error = msg; }

extern "C" const char* rt_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* rt_strerror(int status) {
  switch (status) {
    case RT_OK: return "ok";
    case RT_ERR_ARG: return "invalid argument";
    case RT_ERR_IO: return "file not found or unreadable";
    case RT_ERR_PARSE: return "malformed scene, resource or mesh file";
    case RT_ERR_LIMIT: return "scene exceeds a fixed limit of the render path";
    case RT_ERR_DEVICE: return "HIP device error";
    case RT_ERR_NOMEM: return "out of memory";
    default: return "unknown status";
  }
}

bool rt_decode_image_file(const std::string& path, int& W, int& H, std::vector<uint8_t>& rgb, std::string& err);   // rt_image.cpp

namespace {

using Clock = std::chrono::steady_clock;
inline double seconds_since(Clock::time_point t0) {
  return std::chrono::duration<double>(Clock::now() - t0).count();
}

struct Fail {
  int code; std::string msg;
};

bool slurp(const std::string& path, std::string& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  size_t got = n > 0 ? fread(&out[0], 1, (size_t)n, f) : 0;
  fclose(f);
  out.resize(got);
  return true;
}

// ----------------------------------------------------- host float helpers
// Dot product with float products and a double accumulator (vector.cu:48-50).
inline float dot_dacc(const float* a, const float* b) {
  double s = 0.0;
  s += (double)(a[0] * b[0]);
  s += (double)(a[1] * b[1]);
  s += (double)(a[2] * b[2]);
  return (float)s;
}
inline void normalize_dacc(float* v) {   // vector.cu:57-63
  const float m = sqrtf(dot_dacc(v, v));
  v[0] = v[0] / m; v[1] = v[1] / m; v[2] = v[2] / m;
}
inline void cross3(float* r, const float* u, const float* v) {   // vector.cu:52-55
  const float x = u[1] * v[2] - u[2] * v[1];
  const float y = u[2] * v[0] - u[0] * v[2];
  const float z = u[0] * v[1] - u[1] * v[0];
  r[0] = x; r[1] = y; r[2] = z;
}

// ------------------------------------------------------------------- JSON
// Small DOM; members keep file order and scalars keep their source text, because the reference
// converts every number with stof/stoi of the literal (parsing.cu:207,215,230).
struct Json {
  enum Kind { SCALAR, OBJECT, ARRAY } kind = SCALAR;
  std::string text;
  std::vector<std::string> keys;   // OBJECT only
  std::vector<Json> items;         // OBJECT values / ARRAY elements
  float as_float() const { return strtof(text.c_str(), nullptr); }
  int as_int() const { return atoi(text.c_str()); }
  void floats(std::vector<float>& out) const { for (const Json& j : items) out.push_back(j.as_float()); }
};

class JsonReader {
 public:
  explicit JsonReader(const std::string& s) : s_(s) {}
  Json parse() {
    Json j = value();
    skip();
    if (pos_ != s_.size()) throw Fail{RT_ERR_PARSE, "trailing characters in JSON"};
    return j;
  }

 private:
  const std::string& s_;
  size_t pos_ = 0;
  void skip() { while (pos_ < s_.size() && isspace((unsigned char)s_[pos_])) pos_++; }
  char peek() { skip(); if (pos_ >= s_.size()) throw Fail{RT_ERR_PARSE, "unexpected end of JSON"}; return s_[pos_]; }
  void expect(char c) { if (peek() != c) throw Fail{RT_ERR_PARSE, std::string("JSON: expected '") + c + "'"}; pos_++; }
  std::string quoted() {
    expect('"');
    std::string o;
    while (pos_ < s_.size() && s_[pos_] != '"') {
      char c = s_[pos_++];
      if (c == '\\' && pos_ < s_.size()) {
        char e = s_[pos_++];
        switch (e) { case 'n': c = '\n'; break; case 't': c = '\t'; break; case 'r': c = '\r'; break; default: c = e; }
      }
      o.push_back(c);
    }
    if (pos_ >= s_.size()) throw Fail{RT_ERR_PARSE, "unterminated JSON string"};
    pos_++;
    return o;
  }
  Json value() {
    Json j;
    char c = peek();
    if (c == '{') {
      j.kind = Json::OBJECT; pos_++;
      if (peek() == '}') { pos_++; return j; }
      for (;;) {
        j.keys.push_back(quoted());
        expect(':');
        j.items.push_back(value());
        if (peek() == ',') { pos_++; continue; }
        expect('}');
        break;
      }
    } else if (c == '[') {
      j.kind = Json::ARRAY; pos_++;
      if (peek() == ']') { pos_++; return j; }
      for (;;) {
        j.items.push_back(value());
        if (peek() == ',') { pos_++; continue; }
        expect(']');
        break;
      }
    } else if (c == '"') {
      j.text = quoted();
    } else {
      size_t b = pos_;
      while (pos_ < s_.size() && !isspace((unsigned char)s_[pos_]) && s_[pos_] != ',' && s_[pos_] != '}' && s_[pos_] != ']') pos_++;
      if (pos_ == b) throw Fail{RT_ERR_PARSE, "JSON: empty value"};
      j.text = s_.substr(b, pos_ - b);
    }
    return j;
  }
};

Json read_json_file(const std::string& path) {
  std::string src;
  if (!slurp(path, src)) throw Fail{RT_ERR_IO, "cannot open " + path};
  return JsonReader(src).parse();
}

// ------------------------------------------------------------------ meshes
struct HostMesh {
  std::vector<float> pos;        // 3 per OBJ position index
  std::vector<float> nrm;        // 3 per position index (first normal seen in a face, or computed)
  std::vector<float> uv;         // 2 per position index: the `vt` of the first face corner that uses the position (parsing.cu:106-108), else 0
  std::vector<uint8_t> used;     // position index referenced by a face
  std::vector<int32_t> face;     // 3 position indices per face
  std::vector<float> face_n;     // 3 per face
  std::vector<int32_t> adj_start, adj;   // CSR vertex -> faces, in face order, one entry per corner
  bool file_normals = false;
  size_t n_faces() const { return face.size() / 3; }
};

// Cursor over one line that splits on a single delimiter character the way std::getline(stream,
// token, delim) does (parsing.cu:35-118 tokenises every line like that).
struct Splitter {
  const char* p; const char* end; bool exhausted;
  Splitter(const char* b, const char* e) : p(b), end(e), exhausted(b == e) {}
  // Returns [tb,te) of the next token; empty + exhausted once the line is consumed.
  void next(char delim, const char*& tb, const char*& te) {
    if (exhausted || p >= end) { exhausted = true; tb = te = end; return; }
    tb = p;
    const char* q = (const char*)memchr(p, delim, (size_t)(end - p));
    if (q) { te = q; p = q + 1; } else { te = end; p = end; }
  }
};
inline float tok_float(Splitter& sp) {
  const char *b, *e; sp.next(' ', b, e);
  char buf[64]; size_t n = std::min<size_t>((size_t)(e - b), sizeof(buf) - 1);
  memcpy(buf, b, n); buf[n] = 0;
  return strtof(buf, nullptr);
}
inline long tok_index(Splitter& sp, char delim) {
  const char *b, *e; sp.next(delim, b, e);
  char buf[32]; size_t n = std::min<size_t>((size_t)(e - b), sizeof(buf) - 1);
  memcpy(buf, b, n); buf[n] = 0;
  return (long)strtoul(buf, nullptr, 10);
}

void compute_face_normal(HostMesh& m, size_t f) {   // triangle.cu:11-14
  const float* p0 = &m.pos[3 * m.face[3 * f + 0]];
  const float* p1 = &m.pos[3 * m.face[3 * f + 1]];
  const float* p2 = &m.pos[3 * m.face[3 * f + 2]];
  const float a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
  const float b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
  float n[3]; cross3(n, a, b); normalize_dacc(n);
  m.face_n[3 * f + 0] = n[0]; m.face_n[3 * f + 1] = n[1]; m.face_n[3 * f + 2] = n[2];
}
void compute_vertex_normals(HostMesh& m) {          // vertex.cu:8-12 for every used vertex
  const size_t nv = m.used.size();
  for (size_t v = 0; v < nv; v++) {
    if (!m.used[v]) continue;
    float n[3] = {0.f, 0.f, 0.f};
    for (int32_t k = m.adj_start[v]; k < m.adj_start[v + 1]; k++) {
      const float* fn = &m.face_n[3 * m.adj[k]];
      n[0] = n[0] + fn[0]; n[1] = n[1] + fn[1]; n[2] = n[2] + fn[2];
    }
    normalize_dacc(n);
    m.nrm[3 * v + 0] = n[0]; m.nrm[3 * v + 1] = n[1]; m.nrm[3 * v + 2] = n[2];
  }
}

// OBJ subset of parsing.cu:21-130: skip to the first "v" line; one contiguous block each of v, vt, vn
// (in that order, vt/vn optional); skip to the first "f"; triangles until the first non-"f" line.
HostMesh load_obj(const std::string& path) {
  std::string src;
  if (!slurp(path, src)) throw Fail{RT_ERR_IO, "No " + path + " has been found"};
  HostMesh m;
  const char* cur = src.data();
  const char* const fin = src.data() + src.size();
  const char *lb = nullptr, *le = nullptr;
  bool eof = false;
  auto fetch = [&]() {   // next line without its '\n'; past the end every line is empty
    if (cur >= fin) { lb = le = fin; eof = true; return; }
    lb = cur;
    const char* q = (const char*)memchr(cur, '\n', (size_t)(fin - cur));
    if (q) { le = q; cur = q + 1; } else { le = fin; cur = fin; }
  };
  auto head_is = [&](Splitter& sp, const char* word) {
    const char *b, *e; sp.next(' ', b, e);
    size_t n = strlen(word);
    return (size_t)(e - b) == n && memcmp(b, word, n) == 0;
  };
  std::vector<float> vn, vt;
  // skip to the first vertex
  for (;;) {
    fetch();
    if (eof) throw Fail{RT_ERR_PARSE, "no vertex lines in " + path};
    Splitter sp(lb, le);
    if (head_is(sp, "v")) { for (int k = 0; k < 3; k++) m.pos.push_back(tok_float(sp)); break; }
  }
  enum { V, VT, VN, SEEK_F, F, DONE } state = V;
  std::vector<int32_t> first_normal;   // per position index, -1 = none
  while (state != DONE) {
    fetch();
    Splitter sp(lb, le);
    const char *hb, *he; sp.next(' ', hb, he);
    auto is = [&](const char* w) { size_t n = strlen(w); return (size_t)(he - hb) == n && memcmp(hb, w, n) == 0; };
    if (state == V) { if (is("v")) { for (int k = 0; k < 3; k++) m.pos.push_back(tok_float(sp)); continue; } state = VT; }
    if (state == VT) { if (is("vt")) { vt.push_back(tok_float(sp)); vt.push_back(tok_float(sp)); continue; } state = VN; }
    if (state == VN) { if (is("vn")) { for (int k = 0; k < 3; k++) vn.push_back(tok_float(sp)); continue; } state = SEEK_F; }
    if (state == SEEK_F) {
      if (!is("f")) { if (eof) throw Fail{RT_ERR_PARSE, "no face lines in " + path}; continue; }
      state = F;
      const size_t nv = m.pos.size() / 3;
      m.used.assign(nv, 0); m.nrm.assign(3 * nv, 0.f); m.uv.assign(2 * nv, 0.f); first_normal.assign(nv, -1);
      m.file_normals = !vn.empty();
    }
    if (state == F) {
      if (!is("f")) { state = DONE; break; }
      const size_t nv = m.used.size();
      for (int c = 0; c < 3; c++) {
        long vin, nin = -1, uvin = -1;
        if (m.file_normals) vin = tok_index(sp, '/') - 1; else vin = tok_index(sp, ' ') - 1;
        if (!vt.empty()) uvin = tok_index(sp, '/') - 1;
        else if (m.file_normals) { const char *b, *e; sp.next('/', b, e); }
        if (m.file_normals) nin = tok_index(sp, ' ') - 1;
        if (vin < 0 || (size_t)vin >= nv) throw Fail{RT_ERR_PARSE, "face references a missing vertex in " + path};
        if (!m.used[vin]) {
          m.used[vin] = 1;
          if (uvin >= 0 && (size_t)uvin < vt.size() / 2) { m.uv[2 * vin] = vt[2 * uvin]; m.uv[2 * vin + 1] = vt[2 * uvin + 1]; }
          if (nin != -1) {
            if (nin < 0 || (size_t)nin >= vn.size() / 3) throw Fail{RT_ERR_PARSE, "face references a missing normal in " + path};
            m.nrm[3 * vin + 0] = vn[3 * nin + 0]; m.nrm[3 * vin + 1] = vn[3 * nin + 1]; m.nrm[3 * vin + 2] = vn[3 * nin + 2];
          }
        }
        m.face.push_back((int32_t)vin);
      }
    }
  }
  const size_t nf = m.n_faces(), nv = m.used.size();
  // adjacency (Vertex::addT order = face order, one entry per corner)
  m.adj_start.assign(nv + 1, 0);
  for (int32_t v : m.face) m.adj_start[v + 1]++;
  for (size_t v = 0; v < nv; v++) m.adj_start[v + 1] += m.adj_start[v];
  m.adj.resize(m.face.size());
  { std::vector<int32_t> fill(m.adj_start.begin(), m.adj_start.end() - 1);
    for (size_t f = 0; f < nf; f++) for (int c = 0; c < 3; c++) m.adj[fill[m.face[3 * f + c]]++] = (int32_t)f; }
  m.face_n.resize(3 * nf);
  for (size_t f = 0; f < nf; f++) compute_face_normal(m, f);   // Triangle ctor (triangle.cu:9)
  if (!m.file_normals) compute_vertex_normals(m);              // parsing.cu:121
  return m;
}

void mesh_scale(HostMesh& m, const std::vector<float>& s) {       // mesh.cu:22-26
  for (size_t v = 0; v < m.used.size(); v++) for (int k = 0; k < 3; k++) m.pos[3 * v + k] = m.pos[3 * v + k] * s[k];
}
void mesh_translate(HostMesh& m, const std::vector<float>& t) {   // mesh.cu:44-48
  for (size_t v = 0; v < m.used.size(); v++) for (int k = 0; k < 3; k++) m.pos[3 * v + k] = m.pos[3 * v + k] + t[k];
}
void mesh_rotate(HostMesh& m, const std::vector<float>& r) {      // mesh.cu:28-42, vector.cu:35-41
  const float ang[3] = {(float)(M_PI * r[0] / 180), (float)(M_PI * r[1] / 180), (float)(M_PI * r[2] / 180)};
  for (int axis = 0; axis < 3; axis++) {
    const float c = cosf(ang[axis]), s = sinf(ang[axis]);
    float R[3][3];
    if (axis == 0) { float t[3][3] = {{1, 0, 0}, {0, c, -s}, {0, s, c}}; memcpy(R, t, sizeof R); }
    else if (axis == 1) { float t[3][3] = {{c, 0, s}, {0, 1, 0}, {-s, 0, c}}; memcpy(R, t, sizeof R); }
    else { float t[3][3] = {{c, -s, 0}, {s, c, 0}, {0, 0, 1}}; memcpy(R, t, sizeof R); }
    for (size_t v = 0; v < m.used.size(); v++) {
      float* p = &m.pos[3 * v];
      float o[3];
      for (int i = 0; i < 3; i++) o[i] = R[i][0] * p[0] + R[i][1] * p[1] + R[i][2] * p[2];
      p[0] = o[0]; p[1] = o[1]; p[2] = o[2];
    }
  }
  for (size_t f = 0; f < m.n_faces(); f++) compute_face_normal(m, f);
  compute_vertex_normals(m);
}

// ------------------------------------------------------------------ octree
// Separating-axis test of triangle.cu:61-105: nine edge x box-normal axes only (the three box axes and the
// triangle normal are missing, so leaves also hold triangles that merely come close), projections with the
// double-accumulating dot over the three triangle vertices and the eight box corners, intervals seeded with +-64.
// Evaluated here with everything that does not depend on the box taken out of it.
//   * The axis edge x unit_k has a zero k-component and the other two are +-edge components, so a projection
//     is the double sum of two float products (the third product is +-0 and leaves the sum unchanged).
//   * The triangle's interval on an axis depends on the triangle only: computed once per triangle.
//   * Rounding is monotone, so the box interval over the eight corners is attained at the two corners picked
//     by the signs of the axis components: two projections instead of eight.
// The decisions are those of the literal form bit for bit (the tests compare the trees with the oracle's, which evaluates it literally).
struct TriAxes {
  float a[9], b[9];        // the two non-zero axis components, on coordinates (k+1)%3 and (k+2)%3 of axis 3*k+ei
  float tmin[9], tmax[9];  // triangle interval, seeded with +-64
  uint32_t live;           // bit per axis: 0 = degenerate axis, skipped by the reference
};

inline float proj2(float c1, float a, float c2, float b) {   // dot_dacc restricted to the two live coordinates
  double s = 0.0;
  s += (double)(c1 * a);
  s += (double)(c2 * b);
  return (float)s;
}

TriAxes make_tri_axes(const float* tv) {
  const float* v0 = tv; const float* v1 = tv + 3; const float* v2 = tv + 6;
  const float edge[3][3] = {{v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]},
                            {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]},
                            {v0[0] - v2[0], v0[1] - v2[1], v0[2] - v2[2]}};
  static const float unit[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  TriAxes T; T.live = 0;
  for (int bi = 0; bi < 3; bi++) for (int ei = 0; ei < 3; ei++) {
    const int x = 3 * bi + ei, i1 = (bi + 1) % 3, i2 = (bi + 2) % 3;
    float ax[3]; cross3(ax, edge[ei], unit[bi]);
    T.a[x] = ax[i1]; T.b[x] = ax[i2];
    float tmin = RT_INF, tmax = -RT_INF;
    if (dot_dacc(ax, ax)) {
      T.live |= 1u << x;
      for (int j = 0; j < 3; j++) {
        const float pr = dot_dacc(tv + 3 * j, ax);
        tmax = std::max(tmax, pr); tmin = std::min(tmin, pr);
      }
    }
    T.tmin[x] = tmin; T.tmax[x] = tmax;
  }
  return T;
}

inline bool tri_axes_touch_box(const TriAxes& T, const float* ini, const float* end) {
  for (int x = 0; x < 9; x++) {
    if (!((T.live >> x) & 1u)) continue;
    const int bi = x / 3, i1 = (bi + 1) % 3, i2 = (bi + 2) % 3;
    const float a = T.a[x], b = T.b[x];
    const float lo1 = std::min(ini[i1], end[i1]), hi1 = std::max(ini[i1], end[i1]);
    const float lo2 = std::min(ini[i2], end[i2]), hi2 = std::max(ini[i2], end[i2]);
    // c * a is smallest at lo when a >= 0, at hi otherwise
    const float bmin = std::min(RT_INF, proj2(a >= 0 ? lo1 : hi1, a, b >= 0 ? lo2 : hi2, b));
    const float bmax = std::max(-RT_INF, proj2(a >= 0 ? hi1 : lo1, a, b >= 0 ? hi2 : lo2, b));
    if (T.tmax[x] < bmin || T.tmin[x] > bmax) return false;
  }
  return true;
}

// Top-down construction that yields the tree the reference's incremental insertion yields
// (octtree.cu:14-20,162-174): a node's triangle set is the parent's set filtered by the node's box
// test, kept in ascending triangle order (= insertion order), and a node is a branch exactly when
// that set has more than 50 members, because a leaf splits the moment its 51st member arrives and
// re-inserts its members in order.  Nodes are laid out so the eight children of a branch are
// consecutive and subtrees follow depth-first order, which is the order rays walk them.
//
// Subtrees are independent, so they are built by concurrent tasks into private arrays (indices relative to
// the subtree) and spliced in depth-first order afterwards: the result does not depend on the number of threads.
struct TreeBuilder {
  rt_scene& sc;
  std::vector<TriAxes> axes;
  std::atomic<int> spare_threads{0};
  // A set that no subdivision separates (more than 50 triangles through one point, or triangles as large as the root
  // box) would grow eightfold per level: the first branch that hits the depth limit, or a tree whose leaves list more
  // entries than any real mesh needs, ends the whole build.
  std::atomic<bool> give_up{false};
  std::atomic<int64_t> refs_total{0};
  static constexpr int64_t REFS_LIMIT = int64_t(1) << 28;
  explicit TreeBuilder(rt_scene& s) : sc(s) {}

  struct Sub {                       // descendants of one node, in depth-first block order
    std::vector<RtNode> nodes;       // `first` of branches relative to nodes[0]; `first` of leaves relative to refs[0]
    std::vector<int32_t> refs;
    int max_depth = 0;
    bool too_deep = false;
    bool too_large = false;
  };

  static void child_box(const float* ini, const float* end, int c, float* ci, float* ce) {   // octtree.cu:95-111
    for (int k = 0; k < 3; k++) {
      const float mid = (end[k] + ini[k]) / 2;
      const bool high = (c >> k) & 1;
      ci[k] = high ? mid : ini[k];
      ce[k] = high ? end[k] : mid;
    }
  }
  static void set_bounds(RtNode& n, const float* ini, const float* end) {
    for (int k = 0; k < 3; k++) { n.a[k] = ini[k]; n.b[k] = end[k]; }
  }
  static void set_links(RtNode& n, int32_t first, int32_t count) {
    memcpy(&n.a[3], &first, 4); memcpy(&n.b[3], &count, 4);
  }
  static void get_links(const RtNode& n, int32_t& first, int32_t& count) {
    memcpy(&first, &n.a[3], 4); memcpy(&count, &n.b[3], 4);
  }

  // Fills `self` (bounds already set) and appends its descendants to `out`; depth of the root is 0.
  void fill(RtNode& self, const std::vector<int32_t>& tris, int depth, bool force_branch, Sub& out) {
    if (depth > out.max_depth) out.max_depth = depth;
    if (give_up.load(std::memory_order_relaxed)) { set_links(self, (int32_t)out.refs.size(), 0); return; }
    if (!force_branch && (int)tris.size() <= RT_LEAF_CAP) {
      if (refs_total.fetch_add((int64_t)tris.size(), std::memory_order_relaxed) > REFS_LIMIT) { out.too_large = true; give_up = true; }
      set_links(self, (int32_t)out.refs.size(), (int32_t)tris.size());
      out.refs.insert(out.refs.end(), tris.begin(), tris.end());
      return;
    }
    if (depth + 1 >= RT_OCT_STACK) { out.too_deep = true; give_up = true; set_links(self, (int32_t)out.refs.size(), 0); return; }
    const int32_t first = (int32_t)out.nodes.size();
    set_links(self, first, -1);
    float ini[3], end[3];
    for (int k = 0; k < 3; k++) { ini[k] = self.a[k]; end[k] = self.b[k]; }   // before out.nodes grows: self may live in it
    out.nodes.resize(out.nodes.size() + 8);
    RtNode kids[8];
    for (int c = 0; c < 8; c++) {
      float ci[3], ce[3]; child_box(ini, end, c, ci, ce);
      kids[c] = RtNode{}; set_bounds(kids[c], ci, ce);
    }
    auto members = [&](int c, std::vector<int32_t>& sub) {
      float ci[3], ce[3];
      for (int k = 0; k < 3; k++) { ci[k] = kids[c].a[k]; ce[k] = kids[c].b[k]; }
      sub.clear();
      for (int32_t t : tris) if (tri_axes_touch_box(axes[t], ci, ce)) sub.push_back(t);
    };
    // large sets: the eight children are filtered and built concurrently while threads are to be had
    const bool fork = tris.size() >= 4096;
    Sub part[8];
    std::thread th[8];
    bool forked[8] = {false, false, false, false, false, false, false, false};
    for (int c = 0; c < 8; c++) {
      auto job = [this, c, depth, &kids, &part, &members]() {
        std::vector<int32_t> sub; members(c, sub);
        fill(kids[c], sub, depth + 1, false, part[c]);
      };
      if (fork && c < 7 && spare_threads.fetch_sub(1) > 0) { th[c] = std::thread(job); forked[c] = true; }
      else { if (fork && c < 7) spare_threads.fetch_add(1); job(); }
    }
    for (int c = 0; c < 8; c++) if (forked[c]) { th[c].join(); spare_threads.fetch_add(1); }
    // splice: child c's descendants follow those of children 0..c-1
    for (int c = 0; c < 8; c++) {
      const int32_t node_base = (int32_t)out.nodes.size(), ref_base = (int32_t)out.refs.size();
      int32_t f, n; get_links(kids[c], f, n);
      set_links(kids[c], f + (n < 0 ? node_base : ref_base), n);
      out.nodes[first + c] = kids[c];
      for (RtNode d : part[c].nodes) {
        get_links(d, f, n);
        set_links(d, f + (n < 0 ? node_base : ref_base), n);
        out.nodes.push_back(d);
      }
      out.refs.insert(out.refs.end(), part[c].refs.begin(), part[c].refs.end());
      if (part[c].max_depth > out.max_depth) out.max_depth = part[c].max_depth;
      out.too_deep |= part[c].too_deep;
      out.too_large |= part[c].too_large;
    }
  }

  void build() {   // main.cu:25-31
    const float ini[3] = {-RT_INF / 4, -RT_INF / 4, RT_INF / 4}, end[3] = {RT_INF / 4, RT_INF / 4, -RT_INF / 4};
    const int32_t nt = (int32_t)(sc.tri_verts.size() / 9);
    axes.resize(nt);
    for (int32_t t = 0; t < nt; t++) axes[t] = make_tri_axes(&sc.tri_verts[9 * (size_t)t]);
    std::vector<int32_t> all;
    for (int32_t t = 0; t < nt; t++) if (tri_axes_touch_box(axes[t], ini, end)) all.push_back(t);
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = getenv("RT_AMD_BUILD_THREADS")) hw = (unsigned)std::max(1, atoi(e));
    spare_threads = (int)std::min(std::max(hw, 1u), 32u) - 1;
    RtNode root{}; set_bounds(root, ini, end);
    Sub rest;
    fill(root, all, 0, true, rest);
    if (rest.too_large)
      throw Fail{RT_ERR_LIMIT, "octree leaves would list more than 2^28 triangles (the mesh does not separate under subdivision)"};
    if (rest.too_deep)
      throw Fail{RT_ERR_LIMIT, "octree would be deeper than the 20-entry traversal stack (more than 50 triangles meet in one point)"};
    // the root takes index 0 and its descendants follow: shift every link by one node
    sc.nodes.clear(); sc.nodes.reserve(rest.nodes.size() + 2);
    int32_t f, n; get_links(root, f, n); set_links(root, f + 1, n);
    sc.nodes.push_back(root);
    for (RtNode d : rest.nodes) { get_links(d, f, n); if (n < 0) set_links(d, f + 1, n); sc.nodes.push_back(d); }
    sc.leaf_refs.swap(rest.refs);
  }
};

void tree_summary(const rt_scene& sc, int32_t id, int depth, rt_scene_info& o) {
  int32_t first, count;
  memcpy(&first, &sc.nodes[id].a[3], 4); memcpy(&count, &sc.nodes[id].b[3], 4);
  if (depth > o.tree_depth) o.tree_depth = depth;
  if (count < 0) { for (int c = 0; c < 8; c++) tree_summary(sc, first + c, depth + 1, o); return; }
  o.n_leaves++; o.n_leaf_refs += count;
  if (!count) o.n_empty_leaves++;
  if (count > o.max_leaf_size) o.max_leaf_size = count;
}

// -------------------------------------------------------------- scene load
struct MaterialTable {
  std::vector<RtMaterial> mats;
  std::map<std::string, int> by_name;
  std::vector<std::string> texture_of;               // per material: name of its texture ("" = none)
  std::map<std::string, std::string> texture_file;   // texture name -> file path
};

MaterialTable load_resources(const std::string& path) {   // parsing.cu:177-269
  MaterialTable T;
  const Json root = read_json_file(path);
  for (size_t i = 0; i < root.keys.size(); i++) {
    if (root.keys[i] == "textures") {   // {"__type__": "color_texture", "name": ..., "file_path": ...}
      for (const Json& entry : root.items[i].items) {
        std::string name, file;
        for (size_t k = 0; k < entry.keys.size(); k++) {
          if (entry.keys[k] == "name") name = entry.items[k].text;
          else if (entry.keys[k] == "file_path") file = entry.items[k].text;
        }
        if (!name.empty()) T.texture_file[name] = file;
      }
      continue;
    }
    if (root.keys[i] != "materials") continue;
    for (const Json& entry : root.items[i].items) {
      std::string type, name, texture, filtering; std::vector<float> col, att; int shininess = 0; float ior = 0.f; bool for_ambient = false;
      for (size_t k = 0; k < entry.keys.size(); k++) {
        const std::string& key = entry.keys[k]; const Json& val = entry.items[k];
        if (key == "__type__" || key == "brdf") type = val.text;
        else if (key == "name") name = val.text;
        else if (key == "color") val.floats(col);
        else if (key == "attenuation") val.floats(att);
        else if (key == "refraction_index") ior = val.as_float();
        else if (key == "color_texture") texture = val.text;
        else if (key == "texture_filtering") filtering = val.text;
        else if (key == "use_for_ambient") for_ambient = val.text == "true";
        else if (key == "brdfParams") {
          for (size_t q = 0; q < val.keys.size(); q++) if (val.keys[q] == "shininess") shininess = val.items[q].as_int();
        }
      }
      RtMaterial m; memset(&m, 0, sizeof m);
      if (type == "lambert") m.type = RT_LAMBERT;
      else if (type == "blinnPhong") { m.type = RT_PHONG; m.param = (float)shininess; }
      else if (type == "reflective_material") m.type = RT_REFLECTIVE;
      else if (type == "dielectric_material") { m.type = RT_DIELECTRIC; m.param = ior; }
      else continue;
      if (!texture.empty() && (m.type == RT_LAMBERT || m.type == RT_PHONG)) {
        // `brdf_color_texture_material`: Lambert or Blinn-Phong whose colour is looked up in a texture.  The reference
        // has no behaviour for these (its parser reads past the end of their missing colour, its TextureMaterial is a
        // stub); what they do here is this library's definition, stated in rt_kernels.h (texture_colour).
        m.type = m.type == RT_LAMBERT ? RT_TEX_LAMBERT : RT_TEX_PHONG;
        const uint32_t flags = (filtering == "bilinear" ? RT_TEX_BILINEAR : 0u) | (for_ambient ? RT_TEX_AMBIENT : 0u);
        memcpy(&m.att[1], &flags, 4);
        T.by_name[name] = (int)T.mats.size();
        T.mats.push_back(m); T.texture_of.push_back(texture);
        continue;
      }
      // Other entries without a colour cannot be evaluated: the reference reads past the end of an empty vector for
      // them.  They stay unknown to scenes.
      if (col.size() < 3) continue;
      if (m.type == RT_DIELECTRIC && att.size() < 3) continue;
      for (int c = 0; c < 3; c++) { m.col[c] = col[c]; if (m.type == RT_DIELECTRIC) m.att[c] = att[c]; }
      T.by_name[name] = (int)T.mats.size();
      T.mats.push_back(m); T.texture_of.push_back("");
    }
  }
  return T;
}

void finish_scene(rt_scene& sc, int W, int H, int depth, double parse_seconds);

void load_scene_into(rt_scene& sc, const std::string& scene_path, const std::string& res_path,
                     const std::string& base_dir, int W, int H, int depth) {
  const auto t_parse = Clock::now();
  MaterialTable T = load_resources(res_path);
  sc.materials = T.mats;
  const Json world = read_json_file(scene_path);

  std::vector<float> eye, up, target; float fov = 0.f; bool have_camera = false;
  float bg[3] = {0.f, 0.f, 0.f}, scene_refr = 1.0f; bool shadows = true;   // parsing.cu:16-18
  std::vector<std::vector<int32_t>> mesh_mats;
  std::vector<std::vector<int32_t>> sphere_mats;

  for (size_t i = 0; i < world.keys.size(); i++) {   // parsing.cu:275-470, keys in file order
    const std::string& key = world.keys[i]; const Json& val = world.items[i];
    if (key == "camera") {
      for (size_t k = 0; k < val.keys.size(); k++) {
        const std::string& ck = val.keys[k]; const Json& cv = val.items[k];
        if (ck == "fov") fov = (float)(M_PI * cv.as_float() / 180);
        else if (ck == "position") cv.floats(eye);
        else if (ck == "up") cv.floats(up);
        else if (ck == "target") cv.floats(target);
      }
      have_camera = true;
    } else if (key == "objects") {
      for (const Json& obj : val.items) {
        std::string type, file; std::vector<float> pos, sca, rot, trans; float radius = 0.f; std::vector<int32_t> mats;
        for (size_t k = 0; k < obj.keys.size(); k++) {
          const std::string& ok = obj.keys[k]; const Json& ov = obj.items[k];
          if (ok == "__type__") type = ov.text;
          else if (ok == "file_path") file = ov.text;
          else if (ok == "scaling") ov.floats(sca);
          else if (ok == "rotation") ov.floats(rot);
          else if (ok == "translation") ov.floats(trans);
          else if (ok == "radius") radius = ov.as_float();
          else if (ok == "position") ov.floats(pos);
          else if (ok == "materials") {
            for (const Json& mn : ov.items) {
              auto it = T.by_name.find(mn.text);
              if (it == T.by_name.end()) throw Fail{RT_ERR_PARSE, "unknown material '" + mn.text + "' in " + scene_path};
              mats.push_back(it->second);
            }
          }
        }
        if (type == "sphere") {
          if (pos.size() < 3) throw Fail{RT_ERR_PARSE, "sphere without a position in " + scene_path};
          RtSphere s; memset(&s, 0, sizeof s);
          s.c[0] = pos[0]; s.c[1] = pos[1]; s.c[2] = pos[2]; s.r = radius;
          sc.spheres.push_back(s); sphere_mats.push_back(mats);
        } else if (type == "mesh") {
          const std::string path = (!file.empty() && file[0] == '/') ? file : base_dir + file;
          HostMesh m = load_obj(path);
          if ((sca.size() && sca.size() < 3) || (rot.size() && rot.size() < 3) || (trans.size() && trans.size() < 3))
            throw Fail{RT_ERR_PARSE, "mesh transform needs three components in " + scene_path};
          if (sca.size()) mesh_scale(m, sca);        // parsing.cu:380-382: always scale, rotate, translate
          if (rot.size()) mesh_rotate(m, rot);
          if (trans.size()) mesh_translate(m, trans);
          const int32_t mesh_id = (int32_t)mesh_mats.size();
          mesh_mats.push_back(mats);
          for (size_t f = 0; f < m.n_faces(); f++) {   // mesh.cu:65-97
            RtTriGeom g; RtTriShade s; memset(&g, 0, sizeof g); memset(&s, 0, sizeof s);
            const float* p[3]; const float* n[3];
            for (int c = 0; c < 3; c++) { p[c] = &m.pos[3 * m.face[3 * f + c]]; n[c] = &m.nrm[3 * m.face[3 * f + c]]; }
            for (int k = 0; k < 3; k++) {
              g.v0[k] = p[0][k]; g.e1[k] = p[1][k] - p[0][k]; g.e2[k] = p[2][k] - p[0][k];
              s.v1[k] = p[1][k]; s.v2[k] = p[2][k]; s.n0[k] = n[0][k]; s.n1[k] = n[1][k]; s.n2[k] = n[2][k];
            }
            g.mesh = mesh_id;
            sc.tri_geom.push_back(g); sc.tri_shade.push_back(s);
            for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) { sc.tri_verts.push_back(p[c][k]); sc.tri_normals.push_back(n[c][k]); }
            for (int c = 0; c < 3; c++) { sc.tri_uv.push_back(m.uv[2 * m.face[3 * f + c]]); sc.tri_uv.push_back(m.uv[2 * m.face[3 * f + c] + 1]); }
          }
        }
      }
    } else if (key == "lights") {
      for (const Json& li : val.items) {
        std::string type; std::vector<float> pos, col, dir; float angle_deg = 0.f;
        for (size_t k = 0; k < li.keys.size(); k++) {
          const std::string& lk = li.keys[k]; const Json& lv = li.items[k];
          if (lk == "__type__") type = lv.text;
          else if (lk == "position") lv.floats(pos);
          else if (lk == "color") lv.floats(col);
          else if (lk == "direction") lv.floats(dir);
          else if (lk == "angle") angle_deg = lv.as_float();
        }
        RtLight L; memset(&L, 0, sizeof L);
        const bool spot = type == "spot_light", point = type == "point_light", dirl = type == "directional_light",
                   amb = type == "ambient_light";
        if (!(spot || point || dirl || amb)) continue;
        if (col.size() < 3 || ((spot || point) && pos.size() < 3) || ((spot || dirl) && dir.size() < 3))
          throw Fail{RT_ERR_PARSE, "light is missing a field in " + scene_path};
        for (int c = 0; c < 3; c++) L.col[c] = col[c];
        if (spot || point) for (int c = 0; c < 3; c++) L.pos[c] = pos[c];
        if (spot || dirl) {
          float d[3] = {dir[0], dir[1], dir[2]};
          normalize_dacc(d);               // parsing.cu:427,437
          if (dirl) normalize_dacc(d);     // DirectionalLight's constructor normalises once more (lights.cu:25-27)
          for (int c = 0; c < 3; c++) L.dir[c] = d[c];
        }
        if (spot) { L.type = RT_SPOT; L.angle = (float)(M_PI * angle_deg / 180); L.cos_half = cosf(L.angle / 2); }
        else if (point) L.type = RT_PUNCTUAL;
        else if (dirl) L.type = RT_DIRECTIONAL;
        else L.type = RT_AMBIENT;
        sc.lights.push_back(L);
      }
    } else if (key == "params") {
      for (size_t k = 0; k < val.keys.size(); k++) {
        const std::string& pk = val.keys[k]; const Json& pv = val.items[k];
        if (pk == "background_color") { int c = 0; for (const Json& x : pv.items) if (c < 3) bg[c++] = x.as_float(); }
        else if (pk == "enable_shadows") shadows = (pv.text == "true");
        else if (pk == "refraction_index") scene_refr = pv.as_float();
      }
    }
  }
  if (!have_camera || eye.size() < 3 || up.size() < 3 || target.size() < 3)
    throw Fail{RT_ERR_PARSE, "scene has no usable camera: " + scene_path};

  // textures of the materials some object uses: decoded once each, RGB8 packed one texel to a word
  {
    std::map<std::string, int> loaded;
    auto use = [&](int32_t mi) {
      const std::string& tname = T.texture_of[mi];
      if (tname.empty()) return;
      auto it = loaded.find(tname);
      if (it == loaded.end()) {
        auto tf = T.texture_file.find(tname);
        if (tf == T.texture_file.end()) throw Fail{RT_ERR_PARSE, "material uses the unknown texture '" + tname + "' in " + res_path};
        const std::string path = (!tf->second.empty() && tf->second[0] == '/') ? tf->second : base_dir + tf->second;
        int tw = 0, th = 0; std::vector<uint8_t> rgb; std::string err;
        if (!rt_decode_image_file(path, tw, th, rgb, err)) throw Fail{err.rfind("No ", 0) == 0 ? RT_ERR_IO : RT_ERR_PARSE, err};
        // (a texture names its texels by a 32-bit offset into the scene's one texel array)
        if (sc.texels.size() + (size_t)tw * th >= (1ull << 32)) throw Fail{RT_ERR_LIMIT, "the textures of the scene hold more than 2^32 texels"};
        RtTexture t; t.first = (uint32_t)sc.texels.size(); t.width = (uint32_t)tw; t.height = (uint32_t)th; t.pad = 0;
        for (size_t q = 0; q < (size_t)tw * th; q++) sc.texels.push_back((uint32_t)rgb[3 * q] | ((uint32_t)rgb[3 * q + 1] << 8) | ((uint32_t)rgb[3 * q + 2] << 16));
        it = loaded.emplace(tname, (int)sc.textures.size()).first;
        sc.textures.push_back(t);
      }
      const uint32_t index = (uint32_t)it->second;
      memcpy(&sc.materials[mi].att[0], &index, 4);
    };
    for (auto& mm : sphere_mats) for (int32_t mi : mm) use(mi);
    for (auto& mm : mesh_mats) for (int32_t mi : mm) use(mi);
  }

  // material lists: spheres first, then meshes
  for (size_t s = 0; s < sc.spheres.size(); s++) {
    sc.spheres[s].mat_first = (int32_t)sc.mat_index.size(); sc.spheres[s].mat_count = (int32_t)sphere_mats[s].size();
    sc.mat_index.insert(sc.mat_index.end(), sphere_mats[s].begin(), sphere_mats[s].end());
  }
  for (auto& mm : mesh_mats) {
    sc.mesh_mat_first.push_back((int32_t)sc.mat_index.size()); sc.mesh_mat_count.push_back((int32_t)mm.size());
    sc.mat_index.insert(sc.mat_index.end(), mm.begin(), mm.end());
  }

  // camera basis and frustum (camera.cu:50-68)
  RtCamera& c = sc.cam; memset(&c, 0, sizeof c);
  float w[3] = {eye[0] - target[0], eye[1] - target[1], eye[2] - target[2]}; normalize_dacc(w);
  const float upv[3] = {up[0], up[1], up[2]};
  float u[3]; cross3(u, upv, w); normalize_dacc(u);
  float v[3]; cross3(v, w, u);
  const float top = RT_NEAR * tanf(fov / 2);
  const float right = (float)(unsigned short)W * top / (float)(unsigned short)H;
  for (int k = 0; k < 3; k++) { c.u[k] = u[k]; c.v[k] = v[k]; c.w[k] = w[k]; c.pos[k] = eye[k]; c.bg[k] = bg[k]; }
  c.top = top; c.bottom = -top; c.right = right; c.left = -right;
  c.scene_refr = scene_refr; c.width = W; c.height = H; c.max_depth = depth; c.shadows = shadows ? 1 : 0;

  finish_scene(sc, W, H, depth, seconds_since(t_parse));
}

// What every way of making an rt_scene ends with, once camera, lights, materials, spheres, material lists and the triangle
// arrays are in place: the octree over the triangles (main.cu:25-31), the flat records the kernels read, the summary.
void finish_scene(rt_scene& sc, int W, int H, int depth, double parse_seconds) {
  RtCamera& c = sc.cam;
  rt_scene_info& I = sc.info; memset(&I, 0, sizeof I);
  I.parse_seconds = parse_seconds;
  const auto t_build = Clock::now();
  if (!sc.mesh_mat_first.empty()) {   // (the reference builds a tree whenever the scene has a mesh, main.cu:25)
    TreeBuilder tb(sc); tb.build();
    // Leaf-ordered copy of the intersection records: a leaf's triangles are one contiguous run, so the
    // traversal streams them without the index indirection.
    // RT_LEAF_PAD extra records at the end, so that the traversal may always request "the next one" (the walk's
    // end-of-leaf marker stands for record 63 of the leaf).  Behind them, in the same array, one RtCullRec per
    // record (and as many pads: lane k of a wave reads record k of the leaf it enters, whatever the leaf's length).
    static_assert(RT_LEAF_CAP < 63 && RT_LEAF_PAD >= 64, "the walk keeps a leaf's triangles in a 64-bit mask whose bit 63 marks the end");
    const size_t n_rec = sc.leaf_refs.size() + RT_LEAF_PAD;
    sc.tri_leaf.assign(n_rec + (n_rec * sizeof(RtCullRec) + sizeof(RtTriLeaf) - 1) / sizeof(RtTriLeaf), RtTriLeaf{});
    sc.cull_offset = (uint32_t)(n_rec * sizeof(RtTriLeaf));
    RtCullRec* const cull = reinterpret_cast<RtCullRec*>(sc.tri_leaf.data() + n_rec);
    for (size_t r = sc.leaf_refs.size(); r < n_rec; r++) cull[r] = RtCullRec{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, __builtin_inff()};
    std::vector<RtCullRec> cull_of(sc.tri_geom.size());   // (a triangle sits in several leaves: its record is worked out once)
    for (size_t t = 0; t < sc.tri_geom.size(); t++) cull_of[t] = rt_cull_record(sc.tri_geom[t].v0, sc.tri_geom[t].e1, sc.tri_geom[t].e2);
    for (size_t r = 0; r < sc.leaf_refs.size(); r++) {
      const RtTriGeom& g = sc.tri_geom[sc.leaf_refs[r]];
      RtTriLeaf& t = sc.tri_leaf[r];
      for (int k = 0; k < 3; k++) { t.v0[k] = g.v0[k]; t.e1[k] = g.e1[k]; }
      t.e2xy[0] = g.e2[0]; t.e2xy[1] = g.e2[1]; t.e2yz[0] = g.e2[1]; t.e2yz[1] = g.e2[2]; t.e2zx[0] = g.e2[2]; t.e2zx[1] = g.e2[0];
      t.id = sc.leaf_refs[r];
      // third column of cofactors (rt_cofactor_solve: m02, m12, m22), operation for operation
      t.nx = g.e1[1] * g.e2[2] - g.e1[2] * g.e2[1];
      t.ny = -g.e1[0] * g.e2[2] + g.e1[2] * g.e2[0];
      t.nz = g.e1[0] * g.e2[1] - g.e1[1] * g.e2[0];
      cull[r] = cull_of[sc.leaf_refs[r]];
    }
    // The walk visits children in the fixed order 0..7 whatever the ray, so "what comes after this subtree" is a property of
    // the node: its next sibling that is not an empty leaf, else what comes after its parent.  With that link per node a
    // ray can be walked without a stack (rt_kernel_wave.hip, trace_lanes): accepted branch -> its first child that is not
    // an empty leaf; anything else -> the link.  0xffffffff ends the walk.
    sc.node_skip.assign(sc.nodes.size() + 1, 0xffffffffu);
    {
      struct Threader {
        rt_scene& sc;
        void run(int32_t node, uint32_t after) {
          sc.node_skip[node] = after;
          int32_t first, count;
          memcpy(&first, &sc.nodes[node].a[3], 4); memcpy(&count, &sc.nodes[node].b[3], 4);
          if (count >= 0) return;
          uint32_t next = after;
          for (int c = 7; c >= 0; c--) {
            int32_t cc; memcpy(&cc, &sc.nodes[first + c].b[3], 4);
            if (cc == 0) continue;   // an empty leaf is never visited
            run(first + c, next);
            next = (uint32_t)(first + c);
          }
        }
      } threader{sc};
      threader.run(0, 0xffffffffu);
    }
    sc.nodes.push_back(RtNode{});   // pad; not counted in info.n_nodes
    sc.wnodes.assign(sc.nodes.size() * RT_NODE_VARIANTS, RtWaveNode{});
    std::vector<uint8_t> sibling(sc.nodes.size(), 0);   // position of a record among the eight children of its parent
    for (size_t i = 0; i + 1 < sc.nodes.size(); i++) {
      int32_t first, count;
      memcpy(&first, &sc.nodes[i].a[3], 4); memcpy(&count, &sc.nodes[i].b[3], 4);
      if (count < 0) for (int c = 0; c < 8; c++) sibling[first + c] = (uint8_t)c;
    }
    for (size_t i = 0; i + 1 < sc.nodes.size(); i++) {
      const RtNode& n = sc.nodes[i];
      const float lo[3] = {n.a[0], n.a[1], n.b[2]}, hi[3] = {n.b[0], n.b[1], n.a[2]};
      int32_t first, count;
      memcpy(&first, &n.a[3], 4); memcpy(&count, &n.b[3], 4);
      uint32_t meta = (uint32_t)count;
      if (count < 0) {
        uint32_t occupied = 0;
        for (int c = 0; c < 8; c++) {
          int32_t cc; memcpy(&cc, &sc.nodes[first + c].b[3], 4);
          if (cc != 0) occupied |= 1u << c;
        }
        meta = 0x80000000u | occupied;
      }
      // Is the bundle test worth its price in this leaf?  Not for a couple of triangles, and not where most triangles are
      // large against the leaf (their bounding balls reach every ray that enters the cube) or cannot be ruled out at all.
      if (count >= RT_LEAF_ASK_MIN) {
        const RtCullRec* const cull = reinterpret_cast<const RtCullRec*>(reinterpret_cast<const char*>(sc.tri_leaf.data()) + sc.cull_offset);
        const float side = hi[0] - lo[0];
        int32_t compact = 0;
        for (int32_t r = first; r < first + count; r++) if (std::isfinite(cull[r].Q) && cull[r].R <= side) compact++;
        if (2 * compact >= count) meta |= RT_LEAF_ASK;
      }
      meta |= (uint32_t)sibling[i] << 24;
      for (int v = 0; v < RT_NODE_VARIANTS; v++) {
        RtWaveNode& w = sc.wnodes[(size_t)v * sc.nodes.size() + i];
        for (int k = 0; k < 3; k++) { const bool sw = (v >> k) & 1; w.lohi[2 * k] = sw ? hi[k] : lo[k]; w.lohi[2 * k + 1] = sw ? lo[k] : hi[k]; }
        w.first = first; w.meta = meta;
      }
    }
  }
  I.build_seconds = seconds_since(t_build);
  I.width = W; I.height = H; I.max_depth = depth;
  I.n_triangles = (int32_t)sc.tri_geom.size(); I.n_spheres = (int32_t)sc.spheres.size();
  I.n_lights = (int32_t)sc.lights.size(); I.n_materials = (int32_t)sc.materials.size();
  I.n_nodes = sc.nodes.empty() ? 0 : (int32_t)sc.nodes.size() - 1; I.shadows_enabled = c.shadows;
  if (!sc.nodes.empty()) tree_summary(sc, 0, 0, I);
  for (int32_t mi : sc.mat_index) if (sc.materials[mi].type == RT_REFLECTIVE || sc.materials[mi].type == RT_DIELECTRIC) I.has_bounce_materials = 1;
  {
    auto pushes_of = [&](int32_t first, int32_t count) {
      int32_t p = 0;
      for (int32_t k = 0; k < count; k++) {
        const int32_t t = sc.materials[sc.mat_index[first + k]].type;
        p += t == RT_REFLECTIVE ? 1 : (t == RT_DIELECTRIC ? 2 : 0);
      }
      return p;
    };
    sc.max_pushes_per_hit = 0;
    for (const RtSphere& sp : sc.spheres) sc.max_pushes_per_hit = std::max(sc.max_pushes_per_hit, pushes_of(sp.mat_first, sp.mat_count));
    sc.dielectric_mesh = 0;
    for (size_t m = 0; m < sc.mesh_mat_first.size(); m++) {
      sc.max_pushes_per_hit = std::max(sc.max_pushes_per_hit, pushes_of(sc.mesh_mat_first[m], sc.mesh_mat_count[m]));
      for (int32_t k = 0; k < sc.mesh_mat_count[m]; k++)
        if (sc.materials[sc.mat_index[sc.mesh_mat_first[m] + k]].type == RT_DIELECTRIC) sc.dielectric_mesh = 1;
    }
  }
}

void dump_tree_rec(const rt_scene& sc, int32_t id, int32_t* kind, float* bounds, int32_t* refs, size_t& ni, size_t& ri) {
  int32_t first, count;
  memcpy(&first, &sc.nodes[id].a[3], 4); memcpy(&count, &sc.nodes[id].b[3], 4);
  for (int k = 0; k < 3; k++) { bounds[6 * ni + k] = sc.nodes[id].a[k]; bounds[6 * ni + 3 + k] = sc.nodes[id].b[k]; }
  kind[ni++] = count;
  if (count < 0) { for (int c = 0; c < 8; c++) dump_tree_rec(sc, first + c, kind, bounds, refs, ni, ri); }
  else for (int32_t k = 0; k < count; k++) refs[ri++] = sc.leaf_refs[first + k];
}

// ---------------------------------------------------------------------- PNG
uint32_t crc_table[256]; bool crc_ready = false;
uint32_t crc32_update(uint32_t c, const uint8_t* p, size_t n) {
  if (!crc_ready) {
    for (uint32_t i = 0; i < 256; i++) { uint32_t k = i; for (int b = 0; b < 8; b++) k = (k & 1) ? 0xEDB88320u ^ (k >> 1) : k >> 1; crc_table[i] = k; }
    crc_ready = true;
  }
  for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void png_chunk(std::vector<uint8_t>& out, const char* tag, const std::vector<uint8_t>& body) {
  put_be32(out, (uint32_t)body.size());
  const size_t start = out.size();
  out.insert(out.end(), tag, tag + 4);
  out.insert(out.end(), body.begin(), body.end());
  put_be32(out, crc32_update(0xFFFFFFFFu, &out[start], out.size() - start) ^ 0xFFFFFFFFu);
}

}  // namespace

// ==================================================================== C ABI
extern "C" int rt_scene_load(const char* scene_json, const char* resources_json, const char* base_dir,
                             int width, int height, int max_depth, rt_scene** out) {
  if (!scene_json || !resources_json || !out) { rt_set_error("rt_scene_load: NULL argument"); return RT_ERR_ARG; }
  *out = nullptr;
  if (width < 1 || height < 1 || width > 65535 || height > 65535) { rt_set_error("image size must be 1..65535 per side"); return RT_ERR_ARG; }
  // The ray stack holds 10 rays (ray.cuh:15-18); the reference is only safe up to depth 9.
  if (max_depth < 0 || max_depth >= RT_RAY_STACK) { rt_set_error("max ray depth must be 0..9"); return RT_ERR_LIMIT; }
  std::string bd = base_dir ? base_dir : "";
  if (!bd.empty() && bd.back() != '/') bd.push_back('/');
  rt_scene* sc = nullptr;
  try {
    sc = new rt_scene();
    load_scene_into(*sc, scene_json, resources_json, bd, width, height, max_depth);
  } catch (const Fail& f) {
    delete sc; rt_set_error(f.msg); return f.code;
  } catch (const std::bad_alloc&) {
    delete sc; rt_set_error("out of host memory"); return RT_ERR_NOMEM;
  } catch (const std::exception& e) {
    delete sc; rt_set_error(e.what()); return RT_ERR_PARSE;
  }
  *out = sc;
  return RT_OK;
}

// A scene from the caller's arrays: the same rt_scene as rt_scene_load makes of the files that describe it.
extern "C" int rt_scene_create(const rt_scene_desc* D, rt_scene** out) {
  if (!D || !out) { rt_set_error("rt_scene_create: NULL argument"); return RT_ERR_ARG; }
  *out = nullptr;
  const rt_camera_desc& C = D->camera;
  if (C.width < 1 || C.height < 1 || C.width > 65535 || C.height > 65535) { rt_set_error("image size must be 1..65535 per side"); return RT_ERR_ARG; }
  if (C.ray_depth < 0 || C.ray_depth >= RT_RAY_STACK) { rt_set_error("max ray depth must be 0..9"); return RT_ERR_LIMIT; }
  if (D->n_lights < 0 || D->n_materials < 0 || D->n_mat_index < 0 || D->n_spheres < 0 || D->n_meshes < 0 || D->n_triangles < 0 ||
      (D->n_lights && !D->lights) || (D->n_materials && !D->materials) || (D->n_mat_index && !D->mat_index) || (D->n_spheres && !D->spheres) ||
      (D->n_meshes && (!D->mesh_mat_first || !D->mesh_mat_count)) || (D->n_triangles && (!D->tri_verts || !D->tri_normals || !D->tri_mesh))) {
    rt_set_error("rt_scene_create: a count without its array, or a negative count"); return RT_ERR_ARG;
  }
  if (D->n_triangles && !D->n_meshes) { rt_set_error("rt_scene_create: triangles without a mesh"); return RT_ERR_ARG; }
  auto list_ok = [&](int32_t first, int32_t count) { return first >= 0 && count >= 0 && (long long)first + count <= D->n_mat_index; };
  for (int32_t i = 0; i < D->n_mat_index; i++)
    if (D->mat_index[i] < 0 || D->mat_index[i] >= D->n_materials) { rt_set_error("rt_scene_create: material index out of range"); return RT_ERR_ARG; }
  for (int32_t i = 0; i < D->n_spheres; i++)
    if (!list_ok(D->spheres[i].mat_first, D->spheres[i].mat_count)) { rt_set_error("rt_scene_create: material list of a sphere out of range"); return RT_ERR_ARG; }
  for (int32_t i = 0; i < D->n_meshes; i++)
    if (!list_ok(D->mesh_mat_first[i], D->mesh_mat_count[i])) { rt_set_error("rt_scene_create: material list of a mesh out of range"); return RT_ERR_ARG; }
  for (int32_t i = 0; i < D->n_triangles; i++)
    if (D->tri_mesh[i] < 0 || D->tri_mesh[i] >= D->n_meshes) { rt_set_error("rt_scene_create: mesh index of a triangle out of range"); return RT_ERR_ARG; }
  for (int32_t i = 0; i < D->n_materials; i++)
    if (D->materials[i].type < RT_LAMBERT || D->materials[i].type > RT_DIELECTRIC) {
      rt_set_error("rt_scene_create: material type must be 0..3 (the reference has no behaviour for TEXTURE; texture materials come in through rt_scene_load)");
      return RT_ERR_ARG;
    }
  for (int32_t i = 0; i < D->n_lights; i++)
    if (D->lights[i].type < RT_PUNCTUAL || D->lights[i].type > RT_DIRECTIONAL) { rt_set_error("rt_scene_create: light type must be 0..3"); return RT_ERR_ARG; }
  rt_scene* sc = nullptr;
  try {
    const auto t0 = Clock::now();
    sc = new rt_scene();
    RtCamera& c = sc->cam; memset(&c, 0, sizeof c);
    for (int k = 0; k < 3; k++) { c.u[k] = C.u[k]; c.v[k] = C.v[k]; c.w[k] = C.w[k]; c.pos[k] = C.pos[k]; c.bg[k] = C.bg_col[k]; }
    c.left = C.left; c.right = C.right; c.bottom = C.bottom; c.top = C.top; c.scene_refr = C.scene_refr;
    c.width = C.width; c.height = C.height; c.max_depth = C.ray_depth; c.shadows = C.shadows ? 1 : 0;
    for (int32_t i = 0; i < D->n_lights; i++) {
      const rt_light_desc& l = D->lights[i];
      RtLight L; memset(&L, 0, sizeof L);
      for (int k = 0; k < 3; k++) { L.col[k] = l.col[k]; L.pos[k] = l.pos[k]; L.dir[k] = l.dir[k]; }
      L.type = l.type; L.angle = l.angle;
      if (l.type == RT_SPOT) L.cos_half = cosf(l.angle / 2);   // lights.cu:289,343
      sc->lights.push_back(L);
    }
    for (int32_t i = 0; i < D->n_materials; i++) {
      const rt_material_desc& m = D->materials[i];
      RtMaterial M; memset(&M, 0, sizeof M);
      for (int k = 0; k < 3; k++) { M.col[k] = m.col[k]; M.att[k] = m.att[k]; }
      M.param = m.param; M.type = m.type;
      sc->materials.push_back(M);
    }
    sc->mat_index.assign(D->mat_index, D->mat_index + D->n_mat_index);
    for (int32_t i = 0; i < D->n_spheres; i++) {
      RtSphere S; memset(&S, 0, sizeof S);
      for (int k = 0; k < 3; k++) S.c[k] = D->spheres[i].c[k];
      S.r = D->spheres[i].r; S.mat_first = D->spheres[i].mat_first; S.mat_count = D->spheres[i].mat_count;
      sc->spheres.push_back(S);
    }
    sc->mesh_mat_first.assign(D->mesh_mat_first, D->mesh_mat_first + D->n_meshes);
    sc->mesh_mat_count.assign(D->mesh_mat_count, D->mesh_mat_count + D->n_meshes);
    const size_t n = (size_t)D->n_triangles;
    sc->tri_geom.resize(n); sc->tri_shade.resize(n);
    sc->tri_verts.assign(D->tri_verts, D->tri_verts + 9 * n);
    sc->tri_normals.assign(D->tri_normals, D->tri_normals + 9 * n);
    if (D->tri_uv) sc->tri_uv.assign(D->tri_uv, D->tri_uv + 6 * n); else sc->tri_uv.assign(6 * n, 0.f);
    for (size_t f = 0; f < n; f++) {   // mesh.cu:65-97, as the loader does for the faces of a file
      RtTriGeom& g = sc->tri_geom[f]; RtTriShade& t = sc->tri_shade[f];
      memset(&g, 0, sizeof g); memset(&t, 0, sizeof t);
      const float* p = D->tri_verts + 9 * f; const float* nn = D->tri_normals + 9 * f;
      for (int k = 0; k < 3; k++) {
        g.v0[k] = p[k]; g.e1[k] = p[3 + k] - p[k]; g.e2[k] = p[6 + k] - p[k];
        t.v1[k] = p[3 + k]; t.v2[k] = p[6 + k]; t.n0[k] = nn[k]; t.n1[k] = nn[3 + k]; t.n2[k] = nn[6 + k];
      }
      g.mesh = D->tri_mesh[f];
    }
    finish_scene(*sc, C.width, C.height, C.ray_depth, seconds_since(t0));
  } catch (const Fail& f) {
    delete sc; rt_set_error(f.msg); return f.code;
  } catch (const std::bad_alloc&) {
    delete sc; rt_set_error("out of host memory"); return RT_ERR_NOMEM;
  } catch (const std::exception& e) {
    delete sc; rt_set_error(e.what()); return RT_ERR_PARSE;
  }
  *out = sc;
  return RT_OK;
}

extern "C" int rt_scene_describe(rt_scene* s, rt_scene_desc* out) {
  if (!s || !out) { rt_set_error("rt_scene_describe: NULL argument"); return RT_ERR_ARG; }
  if (!s->textures.empty()) { rt_set_error("rt_scene_describe: the scene has texture materials, which a description cannot carry"); return RT_ERR_LIMIT; }
  try {
    memset(out, 0, sizeof *out);
    const RtCamera& c = s->cam; rt_camera_desc& C = out->camera;
    for (int k = 0; k < 3; k++) { C.u[k] = c.u[k]; C.v[k] = c.v[k]; C.w[k] = c.w[k]; C.pos[k] = c.pos[k]; C.bg_col[k] = c.bg[k]; }
    C.left = c.left; C.right = c.right; C.bottom = c.bottom; C.top = c.top; C.scene_refr = c.scene_refr;
    C.width = c.width; C.height = c.height; C.ray_depth = c.max_depth; C.shadows = c.shadows;
    s->desc_lights.clear(); s->desc_materials.clear(); s->desc_spheres.clear(); s->desc_tri_mesh.clear();
    for (const RtLight& L : s->lights) {
      rt_light_desc l; memset(&l, 0, sizeof l);
      for (int k = 0; k < 3; k++) { l.col[k] = L.col[k]; l.pos[k] = L.pos[k]; l.dir[k] = L.dir[k]; }
      l.angle = L.angle; l.type = L.type;
      s->desc_lights.push_back(l);
    }
    for (const RtMaterial& M : s->materials) {
      rt_material_desc m; memset(&m, 0, sizeof m);
      // (a texture material that no object of the scene uses - s->textures is empty - is described as its plain counterpart: nothing reads it)
      for (int k = 0; k < 3; k++) { m.col[k] = M.col[k]; m.att[k] = M.type >= RT_TEX_LAMBERT ? 0.f : M.att[k]; }
      m.param = M.param; m.type = M.type >= RT_TEX_LAMBERT ? M.type - RT_TEX_LAMBERT : M.type;
      s->desc_materials.push_back(m);
    }
    for (const RtSphere& S : s->spheres) {
      rt_sphere_desc d; memset(&d, 0, sizeof d);
      for (int k = 0; k < 3; k++) d.c[k] = S.c[k];
      d.r = S.r; d.mat_first = S.mat_first; d.mat_count = S.mat_count;
      s->desc_spheres.push_back(d);
    }
    for (const RtTriGeom& g : s->tri_geom) s->desc_tri_mesh.push_back(g.mesh);
  } catch (const std::bad_alloc&) {
    rt_set_error("out of host memory"); return RT_ERR_NOMEM;
  }
  out->n_lights = (int32_t)s->desc_lights.size(); out->lights = s->desc_lights.data();
  out->n_materials = (int32_t)s->desc_materials.size(); out->materials = s->desc_materials.data();
  out->n_mat_index = (int32_t)s->mat_index.size(); out->mat_index = s->mat_index.data();
  out->n_spheres = (int32_t)s->desc_spheres.size(); out->spheres = s->desc_spheres.data();
  out->n_meshes = (int32_t)s->mesh_mat_first.size(); out->mesh_mat_first = s->mesh_mat_first.data(); out->mesh_mat_count = s->mesh_mat_count.data();
  out->n_triangles = (int32_t)s->tri_geom.size();
  out->tri_verts = s->tri_verts.data(); out->tri_normals = s->tri_normals.data(); out->tri_uv = s->tri_uv.empty() ? nullptr : s->tri_uv.data();
  out->tri_mesh = s->desc_tri_mesh.data();
  return RT_OK;
}

extern "C" void rt_scene_destroy(rt_scene* s) { delete s; }

extern "C" int rt_scene_get_info(const rt_scene* s, rt_scene_info* out) {
  if (!s || !out) { rt_set_error("rt_scene_get_info: NULL argument"); return RT_ERR_ARG; }
  *out = s->info;
  return RT_OK;
}

extern "C" int rt_scene_dump_tree(const rt_scene* s, int32_t* kind, float* bounds, int32_t* refs) {
  if (!s || !kind || !bounds || !refs) { rt_set_error("rt_scene_dump_tree: NULL argument"); return RT_ERR_ARG; }
  if (s->nodes.empty()) return RT_OK;
  size_t ni = 0, ri = 0;
  dump_tree_rec(*s, 0, kind, bounds, refs, ni, ri);
  return RT_OK;
}

extern "C" int rt_scene_dump_triangles(const rt_scene* s, float* verts, float* normals, int32_t* mesh) {
  if (!s || !verts || !normals || !mesh) { rt_set_error("rt_scene_dump_triangles: NULL argument"); return RT_ERR_ARG; }
  if (!s->tri_verts.empty()) {
    memcpy(verts, s->tri_verts.data(), s->tri_verts.size() * 4);
    memcpy(normals, s->tri_normals.data(), s->tri_normals.size() * 4);
  }
  for (size_t i = 0; i < s->tri_geom.size(); i++) mesh[i] = s->tri_geom[i].mesh;
  return RT_OK;
}

extern "C" int rt_scene_dump_camera(const rt_scene* s, float* o) {
  if (!s || !o) { rt_set_error("rt_scene_dump_camera: NULL argument"); return RT_ERR_ARG; }
  const RtCamera& c = s->cam;
  memcpy(o, c.u, 12); memcpy(o + 3, c.v, 12); memcpy(o + 6, c.w, 12); memcpy(o + 9, c.pos, 12);
  o[12] = c.left; o[13] = c.right; o[14] = c.bottom; o[15] = c.top; memcpy(o + 16, c.bg, 12);
  return RT_OK;
}

// 8-bit RGB, filter 0 on every row (display.cpp:7-16 writes the same pixels through lodepng).  The scanlines are cut
// into bands that are deflated concurrently as raw streams ending on a byte boundary (Z_SYNC_FLUSH; the last one
// Z_FINISH) and concatenated behind one zlib header; the Adler-32 of the whole is combined from the bands'.
extern "C" int rt_write_png(const char* path, int width, int height, const uint8_t* rgb) {
  if (!path || !rgb || width < 1 || height < 1) { rt_set_error("rt_write_png: bad argument"); return RT_ERR_ARG; }
  const size_t row = (size_t)width * 3;
  const int band_rows = std::max(1, (int)((1u << 20) / (row + 1)));          // about 1 MiB of scanlines per band
  const int n_bands = (height + band_rows - 1) / band_rows;
  struct Band { std::vector<uint8_t> z; uLong adler = 1; size_t raw_n = 0; int rc = Z_OK; };
  std::vector<Band> bands(n_bands);
  auto pack = [&](int bi) {
    Band& B = bands[bi];
    const int y0 = bi * band_rows, y1 = std::min(height, y0 + band_rows);
    std::vector<uint8_t> raw((row + 1) * (size_t)(y1 - y0));
    for (int y = y0; y < y1; y++) { raw[(row + 1) * (y - y0)] = 0; memcpy(&raw[(row + 1) * (y - y0) + 1], rgb + row * y, row); }
    B.raw_n = raw.size();
    B.adler = adler32(1, raw.data(), (uInt)raw.size());
    z_stream zs; memset(&zs, 0, sizeof zs);
    if ((B.rc = deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY)) != Z_OK) return;
    B.z.resize(deflateBound(&zs, (uLong)raw.size()) + 16);
    zs.next_in = raw.data(); zs.avail_in = (uInt)raw.size(); zs.next_out = B.z.data(); zs.avail_out = (uInt)B.z.size();
    const int last = bi == n_bands - 1;
    const int r = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
    B.rc = (last ? r == Z_STREAM_END : (r == Z_OK && zs.avail_in == 0)) ? Z_OK : Z_BUF_ERROR;
    B.z.resize(zs.total_out);
    deflateEnd(&zs);
  };
  {
    unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    nt = std::min<unsigned>(nt, (unsigned)n_bands);
    std::atomic<int> next{0};
    auto worker = [&]() { for (int bi; (bi = next.fetch_add(1)) < n_bands;) pack(bi); };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
  }
  std::vector<uint8_t> z;
  z.push_back(0x78); z.push_back(0x01);
  uLong adler = 1;
  for (int bi = 0; bi < n_bands; bi++) {
    if (bands[bi].rc != Z_OK) { rt_set_error("rt_write_png: deflate failed"); return RT_ERR_NOMEM; }
    z.insert(z.end(), bands[bi].z.begin(), bands[bi].z.end());
    adler = bi == 0 ? bands[bi].adler : adler32_combine(adler, bands[bi].adler, (z_off_t)bands[bi].raw_n);
  }
  put_be32(z, (uint32_t)adler);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr; put_be32(ihdr, (uint32_t)width); put_be32(ihdr, (uint32_t)height);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  png_chunk(out, "IHDR", ihdr); png_chunk(out, "IDAT", z); png_chunk(out, "IEND", {});
  FILE* f = fopen(path, "wb");
  if (!f) { rt_set_error(std::string("cannot write ") + path); return RT_ERR_IO; }
  const size_t w = fwrite(out.data(), 1, out.size(), f);
  fclose(f);
  if (w != out.size()) { rt_set_error(std::string("short write to ") + path); return RT_ERR_IO; }
  return RT_OK;
}
