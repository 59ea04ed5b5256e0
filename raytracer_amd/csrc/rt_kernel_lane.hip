// rt_kernel_lane.hip -- one lane = one pixel with a private traversal stack, literal reference arithmetic.
//
// This is the straightforward formulation of the path: every node test evaluates the six face solves
// of octtree.cu:276-321 and every triangle test the solve of triangle.cu:107-152 exactly as written
// (rt_math.h, *_literal / rt_triangle_test).  It is kept as the on-device cross-check of the
// wave-synchronous kernel (RT_RENDER_LITERAL); built with -ffp-contract=off.
#include "rt_kernels.h"

namespace {

struct Tally { uint32_t rays, shadow, node, leaf, tri, sph; };

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

template <bool BOUNCE, bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) void render_tiles(const RtDeviceView S, const RowMap rows,
                                                         uint8_t* __restrict__ rgb, int32_t* __restrict__ ids,
                                                         float* __restrict__ ts, unsigned long long* __restrict__ counters_all, int tally) {
  unsigned long long* const counters = counters_all + ((blockIdx.x + blockIdx.y) & (RT_COUNTER_COPIES - 1)) * RT_COUNTER_STRIDE;
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
          RtMaterial M = S.materials[S.mat_index[mfirst + mi]];
          bool seen_by_ambient = true;
          if (M.type >= RT_TEX_LAMBERT) {   // a texture material: Lambert / Blinn-Phong with the colour looked up (rt_kernels.h)
            float tu, tv;
            hit_uv(S.tri_uv, hit, N, tu, tv);
            const uint32_t flags = __float_as_uint(M.att[1]);
            const RtVec3 tc = texture_colour(S.textures, S.texels, __float_as_uint(M.att[0]), flags, tu, tv);
            M.col[0] = tc.x; M.col[1] = tc.y; M.col[2] = tc.z;
            M.type -= RT_TEX_LAMBERT; seen_by_ambient = (flags & RT_TEX_AMBIENT) != 0u;
          }
          if (M.type == RT_LAMBERT || M.type == RT_PHONG) {   // material.cu:15-24, lights.cu:260-367
            for (int32_t li = 0; li < S.n_lights; li++) {
              const RtLight L = S.lights[li];
              if (L.type == RT_AMBIENT) {
                if (M.type == RT_LAMBERT && seen_by_ambient) for (int k = 0; k < 3; k++) colr[k] = colr[k] + M.col[k] * L.col[k];
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
  if (!tally) return;
  uint32_t vals[6] = {tl.rays, tl.shadow, tl.node, tl.leaf, tl.tri, tl.sph};
  const int n_out = COUNT ? 6 : 2;
  for (int k = 0; k < n_out; k++) {
    uint32_t v = vals[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0 && v) atomicAdd(&counters[k], (unsigned long long)v);
  }
}


}  // namespace

int rt_preload_lane_kernels() {
  hipFuncAttributes attr;   // asking for a kernel's attributes makes the runtime load the code object it is in
  return (int)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&render_tiles<false, false>));
}

void rt_launch_lane_kernel(const RtDeviceView& view, const RowMap& rows, bool bounce, bool count, uint8_t* rgb, int32_t* ids,
                           float* ts, unsigned long long* counters, bool tally, hipStream_t stream) {
  const dim3 grid((view.cam.width + RT_TILE_W - 1) / RT_TILE_W, (rows.n_local + RT_TILE_H - 1) / RT_TILE_H), block(RT_BLOCK);
#define LAUNCH(B, C) hipLaunchKernelGGL((render_tiles<B, C>), grid, block, 0, stream, view, rows, rgb, ids, ts, counters, tally ? 1 : 0)
  if (bounce && count) LAUNCH(true, true);
  else if (bounce) LAUNCH(true, false);
  else if (count) LAUNCH(false, true);
  else LAUNCH(false, false);
#undef LAUNCH
}
