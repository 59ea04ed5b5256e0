// rt_device.hip -- gfx950 kernels of the render path and the device half of the C ABI.
//
// One lane renders one pixel; a 64-lane wave owns an 8x8 pixel tile so that the rays of a wave
// walk the same octree nodes and their node / triangle records arrive in a few cache lines.
// The per-lane traversal stack lives in LDS ([entry][lane], conflict-free), node records are two
// 16-byte loads, triangle records three.  The arithmetic is the reference's, operation by
// operation (see rt_math.h); this file is built with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_internal.h"
#include "rt_math.h"

#define RT_BLOCK 256
#define RT_TILE_W 16   // pixels per workgroup tile: four 8x8 wave tiles
#define RT_TILE_H 16

struct RtDeviceView {
  RtCamera cam;
  const float4* nodes;        // 2 per node
  const int32_t* leaf_refs;
  const float4* tri_geom;     // 3 per triangle
  const float4* tri_shade;    // 4 per triangle
  const RtSphere* spheres;
  const RtLight* lights;
  const RtMaterial* materials;
  const int32_t* mat_index;
  const int32_t* mesh_mat_first;
  const int32_t* mesh_mat_count;
  int32_t n_spheres, n_lights, has_oct, pad;
};

struct rt_dscene {
  int device = 0;
  RtDeviceView view;
  std::vector<void*> allocs;
  unsigned long long* d_counters = nullptr;   // 9 x u64
  rt_scene_info info;
};

namespace {

enum { C_RAYS = 0, C_SHADOW, C_NODE, C_LEAF, C_TRI, C_SPH, C_UB_LEAVE, C_UB_OVER, C_UB_STACK, C_COUNT };

struct Hit {
  float t;
  int32_t prim;    // >=0 triangle, <=-2 sphere (-2 - index), -1 nothing yet
  float bet, gam;  // barycentrics of the accepted triangle
};

struct Tally { uint32_t rays, shadow, node, leaf, tri, sph; };

__device__ __forceinline__ RtVec3 ld3(const float* p) { return {p[0], p[1], p[2]}; }

// Closest hit of ray.cu:29-54.  Octree first (depth-first, children 0..7, a child is entered iff
// rt_node_accept with the running closest t says so), then every sphere.  When STOP_BELOW is set
// the walk ends as soon as t drops below `stop_t`: for a shadow ray only "t < light distance"
// is consumed (ray.cu:70) and t never grows, so the answer is already fixed at that point.
template <bool COUNT, bool STOP_BELOW>
__device__ __forceinline__ void trace(const RtDeviceView& S, const RtVec3& eye, const RtVec3& dir, Hit& hit, float stop_t,
                                      uint32_t* stk /* &lds[0][lane], stride RT_BLOCK */, Tally& tl) {
  tl.rays++;
  if (S.has_oct) {
    int sp = 0;
    {   // the root is entered without a box test (ray.cu:36-38); it is always a branch
      const float4 ra = S.nodes[0];
      stk[0] = (uint32_t)__float_as_int(ra.w) << 4;
      sp = 1;
    }
    while (sp) {
      const uint32_t e = stk[(sp - 1) * RT_BLOCK];
      const uint32_t cur = e & 15u;
      const uint32_t child = (e >> 4) + cur;
      if (cur == 7u) sp--; else stk[(sp - 1) * RT_BLOCK] = e + 1;
      const float4 na = S.nodes[2 * child], nb = S.nodes[2 * child + 1];
      if (COUNT) tl.node++;
      if (!rt_node_accept_literal({na.x, na.y, na.z}, {nb.x, nb.y, nb.z}, eye, dir, hit.t)) continue;
      const int32_t first = __float_as_int(na.w), count = __float_as_int(nb.w);
      if (count < 0) { stk[sp * RT_BLOCK] = (uint32_t)first << 4; sp++; continue; }
      if (COUNT) tl.leaf++;
      for (int32_t k = 0; k < count; k++) {   // octtree.cu:264-274
        const int32_t id = S.leaf_refs[first + k];
        const float4 g0 = S.tri_geom[3 * id], g1 = S.tri_geom[3 * id + 1], g2 = S.tri_geom[3 * id + 2];
        if (COUNT) tl.tri++;
        float t, bet, gam;
        if (rt_triangle_test({g0.x, g0.y, g0.z}, {g0.w, g1.x, g1.y}, {g1.z, g1.w, g2.x}, eye, dir, hit.t, t, bet, gam)) {
          hit.t = t; hit.prim = id; hit.bet = bet; hit.gam = gam;
        }
      }
      if (STOP_BELOW && hit.t < stop_t) return;
    }
  }
  for (int32_t s = 0; s < S.n_spheres; s++) {   // ray.cu:52-53
    if (COUNT) tl.sph++;
    if (rt_sphere_test(ld3(S.spheres[s].c), S.spheres[s].r, eye, dir, hit.t)) {
      hit.prim = -2 - s;
      if (STOP_BELOW && hit.t < stop_t) return;
    }
  }
}

// Hit point and shading normal of the accepted primitive (triangle.cu:141-147, spheres.cu:92-93).
__device__ __forceinline__ void surface(const RtDeviceView& S, const RtVec3& eye, const RtVec3& dir, const Hit& h,
                                        RtVec3& P, RtVec3& N) {
  if (h.prim >= 0) {
    const float4 g0 = S.tri_geom[3 * h.prim];
    const float4 s0 = S.tri_shade[4 * h.prim], s1 = S.tri_shade[4 * h.prim + 1], s2 = S.tri_shade[4 * h.prim + 2],
                 s3 = S.tri_shade[4 * h.prim + 3];
    const float alp = 1 - h.bet - h.gam, bet = h.bet, gam = h.gam;
    const RtVec3 v0 = {g0.x, g0.y, g0.z}, v1 = {s0.x, s0.y, s0.z}, v2 = {s0.w, s1.x, s1.y};
    const RtVec3 n0 = {s1.z, s1.w, s2.x}, n1 = {s2.y, s2.z, s2.w}, n2 = {s3.x, s3.y, s3.z};
    N = {(n0.x * alp + n1.x * bet) + n2.x * gam, (n0.y * alp + n1.y * bet) + n2.y * gam, (n0.z * alp + n1.z * bet) + n2.z * gam};
    N = rt_normalized(N);
    P = {(v0.x * alp + v1.x * bet) + v2.x * gam, (v0.y * alp + v1.y * bet) + v2.y * gam, (v0.z * alp + v1.z * bet) + v2.z * gam};
  } else {
    const RtSphere& sp = S.spheres[-2 - h.prim];
    P = {eye.x + dir.x * h.t, eye.y + dir.y * h.t, eye.z + dir.z * h.t};
    N = rt_normalized({P.x - sp.c[0], P.y - sp.c[1], P.z - sp.c[2]});
  }
}

struct BounceRay {   // the fields of DRay that survive a push (vector.cu:188-207)
  RtVec3 eye, dir, att;
  float level[RT_MAX_LEVELS];
  int32_t nlevel, depth;
};

// vector.cu:147-156
__device__ __forceinline__ RtVec3 refract_dir(const RtVec3& d, const RtVec3& n, float rori) {
  const float costhe = -rt_dot(d, n);
  const float sinthe = sqrtf(1 - costhe * costhe);      // powf(x, 2) rounds to x*x
  const float sinfi = rori * sinthe;
  const float cosfi = sqrtf(1 - sinfi * sinfi);
  RtVec3 r;
  r.x = ((n.x * costhe + d.x) / sinthe) * sinfi - n.x * cosfi;
  r.y = ((n.y * costhe + d.y) / sinthe) * sinfi - n.y * cosfi;
  r.z = ((n.z * costhe + d.z) / sinthe) * sinfi - n.z * cosfi;
  return r;
}

// Which image rows a launch renders and where they are stored.  Local row lr (0..n_local) is image
// row  row0 + (lr / band_rows) * band_stride + lr % band_rows ; it is stored at image row position
// when `compact` is 0 and at row lr of a band-packed buffer when it is 1.
struct RowMap { int32_t row0, band_rows, band_stride, n_local, compact, pad; };

template <bool BOUNCE, bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) void render_tiles(const RtDeviceView S, const RowMap rows,
                                                         uint8_t* __restrict__ rgb, int32_t* __restrict__ ids,
                                                         float* __restrict__ ts, unsigned long long* __restrict__ counters) {
  __shared__ uint32_t lds_stack[RT_OCT_STACK][RT_BLOCK];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int W = S.cam.width, H = S.cam.height;
  const int col = blockIdx.x * RT_TILE_W + (wave & 1) * 8 + (lane & 7);
  const int lrow = blockIdx.y * RT_TILE_H + (wave >> 1) * 8 + (lane >> 3);
  const int row = rows.row0 + (lrow / rows.band_rows) * rows.band_stride + lrow % rows.band_rows;
  const int orow = rows.compact ? lrow : row;   // row of the output buffers this pixel is stored at
  uint32_t* stk = &lds_stack[0][tid];
  Tally tl = {0, 0, 0, 0, 0, 0};

  if (col < W && lrow < rows.n_local && row < H) {
    const int i = col, j = H - 1 - row;   // camera.cu:94,118: j counts from the bottom
    const RtCamera& c = S.cam;
    // camera.cu:96-99; (i+0.5) is a double, so product and sum are double before the store to float
    const float ic = (float)((double)c.left + (double)((c.right - c.left) / (float)(unsigned short)W) * ((double)i + 0.5));
    const float jc = (float)((double)c.bottom + (double)((c.top - c.bottom) / (float)(unsigned short)H) * ((double)j + 0.5));
    const float kz = -RT_NEAR;
    RtVec3 d = {ic * c.u[0] + jc * c.v[0] + kz * c.w[0], ic * c.u[1] + jc * c.v[1] + kz * c.w[1],
                ic * c.u[2] + jc * c.v[2] + kz * c.w[2]};
    d = rt_normalized(d);

    float color[3] = {0.f, 0.f, 0.f};
    BounceRay rstack[BOUNCE ? RT_RAY_STACK : 1];
    int rsp = 0;
    BounceRay cur;
    cur.eye = {c.pos[0], c.pos[1], c.pos[2]}; cur.dir = d; cur.att = {1.f, 1.f, 1.f};
    cur.level[0] = c.scene_refr; cur.level[1] = 0.f; cur.level[2] = 0.f; cur.nlevel = 0; cur.depth = 0;
    bool primary = true, have = true;

    while (have) {
      Hit hit = {RT_INF, -1, 0.f, 0.f};
      trace<COUNT, false>(S, cur.eye, cur.dir, hit, 0.f, stk, tl);
      if (primary) {
        primary = false;
        const size_t p = (size_t)orow * W + col;
        if (ids) ids[p] = hit.prim;
        if (ts) ts[p] = hit.t;
      }
      float colr[3] = {0.f, 0.f, 0.f};
      if (hit.prim == -1) {
        colr[0] = c.bg[0]; colr[1] = c.bg[1]; colr[2] = c.bg[2];   // ray.cu:86
      } else {
        RtVec3 P, N;
        surface(S, cur.eye, cur.dir, hit, P, N);
        int32_t mfirst, mcount;
        if (hit.prim >= 0) {
          const int32_t mesh = __float_as_int(S.tri_geom[3 * hit.prim + 2].y);
          mfirst = S.mesh_mat_first[mesh]; mcount = S.mesh_mat_count[mesh];
        } else { mfirst = S.spheres[-2 - hit.prim].mat_first; mcount = S.spheres[-2 - hit.prim].mat_count; }
        for (int32_t mi = 0; mi < mcount; mi++) {   // ray.cu:77-85
          const RtMaterial M = S.materials[S.mat_index[mfirst + mi]];
          if (M.type == RT_LAMBERT || M.type == RT_PHONG) {   // material.cu:15-24, lights.cu:260-367
            for (int32_t li = 0; li < S.n_lights; li++) {
              const RtLight L = S.lights[li];
              if (L.type == RT_AMBIENT) {
                if (M.type == RT_LAMBERT) for (int k = 0; k < 3; k++) colr[k] = colr[k] + M.col[k] * L.col[k];
                continue;
              }
              RtVec3 l; float ldist;
              if (L.type == RT_DIRECTIONAL) { l = {L.dir[0] * -1, L.dir[1] * -1, L.dir[2] * -1}; ldist = RT_INF; }
              else l = rt_normalized({L.pos[0] - P.x, L.pos[1] - P.y, L.pos[2] - P.z});
              const RtVec3 sP = {P.x + l.x * RT_EPS, P.y + l.y * RT_EPS, P.z + l.z * RT_EPS};
              if (L.type != RT_DIRECTIONAL) ldist = rt_len({L.pos[0] - sP.x, L.pos[1] - sP.y, L.pos[2] - sP.z});
              if (c.shadows) {   // ray.cu:56-72
                Hit sh = {RT_INF, -1, 0.f, 0.f};
                tl.shadow++;
                trace<COUNT, !COUNT>(S, sP, l, sh, ldist, stk, tl);
                if (ldist > sh.t) continue;
              }
              if (L.type == RT_SPOT && !(-rt_dot(l, ld3(L.dir)) > L.cos_half)) continue;
              float w;
              if (M.type == RT_LAMBERT) { const float nl = rt_dot(N, l); w = nl > 0.f ? nl : 0.f; }
              else {
                const RtVec3 h = rt_normalized({cur.dir.x * -1 + l.x, cur.dir.y * -1 + l.y, cur.dir.z * -1 + l.z});
                const float nh = rt_dot(N, h);
                w = powf(nh > 0.f ? nh : 0.f, M.param);
              }
              for (int k = 0; k < 3; k++) colr[k] = colr[k] + (M.col[k] * L.col[k]) * w;
            }
          } else if (BOUNCE && cur.depth < c.max_depth) {
            const float s2 = 2 * rt_dot(cur.dir, N);
            const RtVec3 rl = rt_normalized({cur.dir.x - N.x * s2, cur.dir.y - N.y * s2, cur.dir.z - N.z * s2});
            if (M.type == RT_REFLECTIVE) {   // material.cu:25-35
              if (rsp < RT_RAY_STACK) {
                BounceRay& r = rstack[rsp++];
                r.eye = {P.x + rl.x * RT_EPS, P.y + rl.y * RT_EPS, P.z + rl.z * RT_EPS}; r.dir = rl;
                r.att = {M.col[0] * cur.att.x, M.col[1] * cur.att.y, M.col[2] * cur.att.z};
                for (int k = 0; k < 3; k++) r.level[k] = cur.level[k];
                r.nlevel = cur.nlevel; r.depth = cur.depth + 1;
              } else atomicAdd(&counters[C_UB_STACK], 1ull);
            } else {   // RT_DIELECTRIC, material.cu:36-99
              const float refr = cur.level[cur.nlevel];
              const RtVec3 rlsP = {P.x + N.x * RT_EPS, P.y + N.y * RT_EPS, P.z + N.z * RT_EPS};
              const float q = (refr - M.param) / (refr + M.param);
              const float R0 = q * q;
              const float dn = rt_dot(cur.dir, N);
              const bool entering = dn < 0;
              float rOrI, cs; RtVec3 nn; float K[3];
              if (entering) { rOrI = refr / M.param; cs = -dn; nn = N; K[0] = K[1] = K[2] = 1.0f; }
              else {
                rOrI = M.param / refr; cs = dn; nn = {N.x * -1, N.y * -1, N.z * -1};
                for (int k = 0; k < 3; k++) K[k] = expf(M.att[k] * -hit.t);
              }
              const float totRefl = 1 - (rOrI * rOrI) * (1 - cs * cs);
              const float a3[3] = {cur.att.x, cur.att.y, cur.att.z};
              float attl[3], attf[3];
              for (int k = 0; k < 3; k++) { attl[k] = (M.col[k] * K[k]) * a3[k]; attf[k] = attl[k]; }
              if (totRefl >= 0) {
                const RtVec3 rf = refract_dir(cur.dir, nn, rOrI);
                const float fresnel = (float)((double)R0 + (double)(1 - R0) * pow(1.0 - (double)cs, 5.0));
                for (int k = 0; k < 3; k++) attf[k] = attf[k] * (1 - fresnel);
                int nl;
                if (entering) {
                  nl = cur.nlevel + 1;
                  if (nl > 2) { nl = 2; atomicAdd(&counters[C_UB_OVER], 1ull); }
                  cur.level[nl] = M.param;
                } else {
                  nl = cur.nlevel - 1;
                  if (nl < 0) { nl = 0; atomicAdd(&counters[C_UB_LEAVE], 1ull); }
                }
                if (rsp < RT_RAY_STACK) {
                  BounceRay& r = rstack[rsp++];
                  r.eye = {P.x + rf.x * RT_EPS, P.y + rf.y * RT_EPS, P.z + rf.z * RT_EPS}; r.dir = rf;
                  r.att = {attf[0], attf[1], attf[2]};
                  for (int k = 0; k < 3; k++) r.level[k] = cur.level[k];
                  r.nlevel = nl; r.depth = cur.depth + 1;
                } else atomicAdd(&counters[C_UB_STACK], 1ull);
                for (int k = 0; k < 3; k++) attl[k] = attl[k] * fresnel;
              }
              if (rsp < RT_RAY_STACK) {
                BounceRay& r = rstack[rsp++];
                r.eye = rlsP; r.dir = rl; r.att = {attl[0], attl[1], attl[2]};
                for (int k = 0; k < 3; k++) r.level[k] = cur.level[k];
                r.nlevel = cur.nlevel; r.depth = cur.depth + 1;
              } else atomicAdd(&counters[C_UB_STACK], 1ull);
            }
          }
        }
      }
      color[0] = color[0] + cur.att.x * colr[0];   // camera.cu:110-111
      color[1] = color[1] + cur.att.y * colr[1];
      color[2] = color[2] + cur.att.z * colr[2];
      have = false;
      if (BOUNCE && rsp > 0) { cur = rstack[--rsp]; have = true; }
    }
    const size_t p = ((size_t)orow * W + col) * 3;
    for (int k = 0; k < 3; k++) {   // camera.cu:113-122; NaN < 1 is false, so NaN -> 1 -> 255
      const float v = (color[k] < 1.f ? color[k] : 1.f) * 255.f;
      rgb[p + k] = v < 0.f ? (uint8_t)0 : (uint8_t)(int)v;
    }
  }

  // work counters: wave reduction, one atomic per wave and counter
  uint32_t vals[6] = {tl.rays, tl.shadow, tl.node, tl.leaf, tl.tri, tl.sph};
  const int n_out = COUNT ? 6 : 2;
  for (int k = 0; k < n_out; k++) {
    uint32_t v = vals[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0 && v) atomicAdd(&counters[k], (unsigned long long)v);
  }
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      rt_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                       \
      return RT_ERR_DEVICE;                                                                  \
    }                                                                                        \
  } while (0)

template <typename T>
int upload(rt_dscene* d, const std::vector<T>& v, const void** out) {
  *out = nullptr;
  if (v.empty()) return RT_OK;
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
  d->allocs.push_back(p);
  HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = p;
  return RT_OK;
}

}  // namespace

extern "C" const char* rt_version(void) { return "rt_amd 0.1 gfx950 (literal-arithmetic kernels)"; }

extern "C" void rt_dscene_destroy(rt_dscene* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  for (void* p : d->allocs) (void)hipFree(p);
  delete d;
}

extern "C" int rt_scene_upload(const rt_scene* s, int device, rt_dscene** out) {
  if (!s || !out) { rt_set_error("rt_scene_upload: NULL argument"); return RT_ERR_ARG; }
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) { rt_set_error("no HIP device is visible; the render path has no CPU fallback"); return RT_ERR_DEVICE; }
  if (device < 0 || device >= n_dev) { rt_set_error("device index out of range"); return RT_ERR_ARG; }
  HIP_TRY(hipSetDevice(device));
  static_assert(sizeof(RtNode) == 32 && sizeof(RtTriGeom) == 48 && sizeof(RtTriShade) == 64, "record sizes");
  rt_dscene* d = new rt_dscene();
  d->device = device; d->info = s->info;
  RtDeviceView& V = d->view; memset(&V, 0, sizeof V);
  V.cam = s->cam;
  int rc;
#define UP(field, vec) if ((rc = upload(d, vec, (const void**)&V.field)) != RT_OK) { rt_dscene_destroy(d); return rc; }
  UP(nodes, s->nodes) UP(leaf_refs, s->leaf_refs) UP(tri_geom, s->tri_geom) UP(tri_shade, s->tri_shade)
  UP(spheres, s->spheres) UP(lights, s->lights) UP(materials, s->materials) UP(mat_index, s->mat_index)
  UP(mesh_mat_first, s->mesh_mat_first) UP(mesh_mat_count, s->mesh_mat_count)
#undef UP
  V.n_spheres = (int32_t)s->spheres.size(); V.n_lights = (int32_t)s->lights.size(); V.has_oct = s->nodes.empty() ? 0 : 1;
  void* c = nullptr;
  if (hipMalloc(&c, C_COUNT * sizeof(unsigned long long)) != hipSuccess) { rt_dscene_destroy(d); rt_set_error("hipMalloc(counters) failed"); return RT_ERR_DEVICE; }
  d->allocs.push_back(c); d->d_counters = (unsigned long long*)c;
  *out = d;
  return RT_OK;
}

static int launch_render(rt_dscene* d, const RowMap& rows, uint8_t* d_rgb, int32_t* d_ids, float* d_t, void* stream_,
                         unsigned flags, rt_stats* stats) {
  const int W = d->view.cam.width;
  hipStream_t stream = (hipStream_t)stream_;
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipMemsetAsync(d->d_counters, 0, C_COUNT * sizeof(unsigned long long), stream));
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  const bool timed = (flags & RT_RENDER_TIME) != 0;
  if (timed) { HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1)); HIP_TRY(hipEventRecord(ev0, stream)); }
  if (rows.n_local > 0) {
    const dim3 grid((W + RT_TILE_W - 1) / RT_TILE_W, (rows.n_local + RT_TILE_H - 1) / RT_TILE_H), block(RT_BLOCK);
    const bool bounce = d->info.has_bounce_materials != 0, count = (flags & RT_RENDER_COUNT_TESTS) != 0;
#define LAUNCH(B, C) hipLaunchKernelGGL((render_tiles<B, C>), grid, block, 0, stream, d->view, rows, d_rgb, d_ids, d_t, d->d_counters)
    if (bounce && count) LAUNCH(true, true);
    else if (bounce) LAUNCH(true, false);
    else if (count) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    HIP_TRY(hipGetLastError());
  }
  float ms = 0.f;
  if (timed) {
    HIP_TRY(hipEventRecord(ev1, stream)); HIP_TRY(hipEventSynchronize(ev1));
    HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
  }
  if (stats) {
    unsigned long long h[C_COUNT];
    HIP_TRY(hipMemcpyAsync(h, d->d_counters, sizeof h, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    memset(stats, 0, sizeof *stats);
    stats->rays = h[C_RAYS]; stats->shadow_rays = h[C_SHADOW]; stats->node_tests = h[C_NODE]; stats->leaf_visits = h[C_LEAF];
    stats->tri_tests = h[C_TRI]; stats->sphere_tests = h[C_SPH]; stats->ub_leave_level0 = h[C_UB_LEAVE];
    stats->ub_level_overflow = h[C_UB_OVER]; stats->ub_stack_overflow = h[C_UB_STACK]; stats->kernel_ms = ms;
  }
  return RT_OK;
}

extern "C" int rt_render_rows(rt_dscene* d, int row_begin, int row_end, uint8_t* d_rgb, int32_t* d_ids, float* d_t,
                              void* stream, unsigned flags, rt_stats* stats) {
  if (!d || !d_rgb) { rt_set_error("rt_render_rows: NULL argument"); return RT_ERR_ARG; }
  const int H = d->view.cam.height;
  if (row_begin < 0 || row_end > H || row_begin > row_end) { rt_set_error("rt_render_rows: rows outside the image"); return RT_ERR_ARG; }
  const int n = row_end - row_begin;
  const RowMap rows = {row_begin, n > 0 ? n : 1, 0, n, 0, 0};
  return launch_render(d, rows, d_rgb, d_ids, d_t, stream, flags, stats);
}

extern "C" int rt_band_count(const rt_dscene* d, int band_rows, int rank, int nranks) {
  if (!d || band_rows < 1 || nranks < 1 || rank < 0 || rank >= nranks) return RT_ERR_ARG;
  const int bands = (d->view.cam.height + band_rows - 1) / band_rows;
  return bands > rank ? (bands - rank + nranks - 1) / nranks : 0;
}

extern "C" int rt_render_bands(rt_dscene* d, int band_rows, int rank, int nranks, uint8_t* d_rgb_local, int32_t* d_ids_local,
                               float* d_t_local, void* stream, unsigned flags, rt_stats* stats) {
  if (!d || !d_rgb_local) { rt_set_error("rt_render_bands: NULL argument"); return RT_ERR_ARG; }
  if (band_rows < 1 || band_rows % RT_TILE_H || nranks < 1 || rank < 0 || rank >= nranks) {
    rt_set_error("rt_render_bands: band_rows must be a positive multiple of 16 and 0 <= rank < nranks");
    return RT_ERR_ARG;
  }
  const int mine = rt_band_count(d, band_rows, rank, nranks);
  const RowMap rows = {rank * band_rows, band_rows, band_rows * nranks, mine * band_rows, 1, 0};
  return launch_render(d, rows, d_rgb_local, d_ids_local, d_t_local, stream, flags, stats);
}

extern "C" int rt_render_rows_host(rt_dscene* d, int row_begin, int row_end, uint8_t* h_rgb, int32_t* h_ids, float* h_t,
                                   unsigned flags, rt_stats* stats) {
  if (!d || !h_rgb) { rt_set_error("rt_render_rows_host: NULL argument"); return RT_ERR_ARG; }
  const int W = d->view.cam.width, H = d->view.cam.height;
  if (row_begin < 0 || row_end > H || row_begin > row_end) { rt_set_error("rt_render_rows_host: rows outside the image"); return RT_ERR_ARG; }
  HIP_TRY(hipSetDevice(d->device));
  const size_t px = (size_t)W * H;
  uint8_t* d_rgb = nullptr; int32_t* d_ids = nullptr; float* d_t = nullptr;
  HIP_TRY(hipMalloc((void**)&d_rgb, px * 3));
  int rc = RT_OK;
  if (h_ids && hipMalloc((void**)&d_ids, px * 4) != hipSuccess) rc = RT_ERR_DEVICE;
  if (rc == RT_OK && h_t && hipMalloc((void**)&d_t, px * 4) != hipSuccess) rc = RT_ERR_DEVICE;
  if (rc == RT_OK) rc = rt_render_rows(d, row_begin, row_end, d_rgb, d_ids, d_t, nullptr, flags, stats);
  if (rc == RT_OK) {
    const size_t off = (size_t)row_begin * W, n = (size_t)(row_end - row_begin) * W;
    if (hipMemcpy(h_rgb + off * 3, d_rgb + off * 3, n * 3, hipMemcpyDeviceToHost) != hipSuccess) rc = RT_ERR_DEVICE;
    if (rc == RT_OK && h_ids && hipMemcpy(h_ids + off, d_ids + off, n * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = RT_ERR_DEVICE;
    if (rc == RT_OK && h_t && hipMemcpy(h_t + off, d_t + off, n * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = RT_ERR_DEVICE;
    if (rc == RT_ERR_DEVICE) rt_set_error("device to host copy failed");
  } else if (rc == RT_ERR_DEVICE && !*rt_last_error()) rt_set_error("hipMalloc failed");
  (void)hipFree(d_rgb); (void)hipFree(d_ids); (void)hipFree(d_t);
  return rc;
}
