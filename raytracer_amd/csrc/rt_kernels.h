// rt_kernels.h -- declarations shared by the HIP translation units of the render path.
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>

#include "rt_internal.h"
#include "rt_math.h"

#define RT_BLOCK 256
#define RT_TILE_W 16   // pixels per workgroup tile: four 8x8 wave tiles
#define RT_TILE_H 16

// Device-resident scene, passed to kernels by value.
struct RtDeviceView {
  RtCamera cam;
  const float4* nodes;        // 2 per node
  const float4* wnodes;       // 2 per node, RtWaveNode layout, RT_NODE_VARIANTS copies of n_wnodes records
  int32_t n_wnodes;           // records per copy (nodes + pad)
  const int32_t* leaf_refs;
  const float4* tri_geom;     // 3 per triangle, indexed by triangle id
  const float4* tri_leaf;     // 4 per leaf reference (RtTriLeaf), in leaf order (same index space as leaf_refs)
  const float4* tri_shade;    // 4 per triangle
  const RtSphere* spheres;
  const RtLight* lights;
  const RtMaterial* materials;
  const int32_t* mat_index;
  const int32_t* mesh_mat_first;
  const int32_t* mesh_mat_count;
  int32_t n_spheres, n_lights, has_oct, pad;
};

// Which image rows a launch renders and where they are stored.  Local row lr (0..n_local) is image
// row  row0 + (lr / band_rows) * band_stride + lr % band_rows ; it is stored at image row position
// when `compact` is 0 and at row lr of a band-packed buffer when it is 1.
// small != 0: a wave owns a 4x4 tile (16 of its 64 lanes carry a pixel) and a workgroup 8x8 pixels; used for frames with
// too few 8x8 tiles to fill the chip, where four times as many, shorter walks finish sooner.
struct RowMap { int32_t row0, band_rows, band_stride, n_local, compact, small; };

enum { C_RAYS = 0, C_SHADOW, C_NODE, C_LEAF, C_TRI, C_SPH, C_UB_LEAVE, C_UB_OVER, C_UB_STACK, C_TRACED,
       // wave-level iteration counts of the instrumented production kernel (how often each code path runs)
       C_W_TRACE, C_W_NODE, C_W_NODE_EXACT, C_W_NODE_TIE, C_W_TRI, C_W_TRI_S2, C_W_TRI_S3, C_W_TRI_FIN, C_COUNT };

// The counters exist in RT_COUNTER_COPIES copies of RT_COUNTER_STRIDE slots (one 256-byte line each); a workgroup adds to
// the copy picked by its index, the host adds the copies up.  One shared copy would serialise a quarter of a million
// same-address atomics per 4K frame in the memory-side atomic unit (measured: ~1.3 ms of a 1.6 ms sphere.json frame).
#define RT_COUNTER_COPIES 64
#define RT_COUNTER_STRIDE 32
static_assert(C_COUNT <= RT_COUNTER_STRIDE, "counter copy too small");

struct Hit {
  float t;
  int32_t prim;    // >=0 triangle, <=-2 sphere (-2 - index), -1 nothing yet
  float bet, gam;  // barycentrics of the accepted triangle
};

struct BounceRay {   // the fields of DRay that survive a push (vector.cu:188-207)
  RtVec3 eye, dir, att;
  float level[RT_MAX_LEVELS];
  int32_t nlevel, depth;
};

__device__ __forceinline__ RtVec3 ld3(const float* p) { return {p[0], p[1], p[2]}; }

// Hit point and shading normal of the accepted primitive (triangle.cu:141-147, spheres.cu:92-93).
__device__ __forceinline__ void surface(const RtDeviceView& S, const RtVec3& eye, const RtVec3& dir, const Hit& h,
                                        RtVec3& P, RtVec3& N) {
  // Results are formed in locals and written once at the end: stores through the references from both arms get
  // merged by the optimiser into a store through a selected pointer, which pins the caller's state in scratch memory.
  float px, py, pz, nx, ny, nz;
  if (h.prim >= 0) {
    const float4 g0 = S.tri_geom[3 * h.prim];
    const float4 s0 = S.tri_shade[4 * h.prim], s1 = S.tri_shade[4 * h.prim + 1], s2 = S.tri_shade[4 * h.prim + 2],
                 s3 = S.tri_shade[4 * h.prim + 3];
    const float alp = 1 - h.bet - h.gam, bet = h.bet, gam = h.gam;
    const RtVec3 v0 = {g0.x, g0.y, g0.z}, v1 = {s0.x, s0.y, s0.z}, v2 = {s0.w, s1.x, s1.y};
    const RtVec3 n0 = {s1.z, s1.w, s2.x}, n1 = {s2.y, s2.z, s2.w}, n2 = {s3.x, s3.y, s3.z};
    const RtVec3 n = rt_normalized({(n0.x * alp + n1.x * bet) + n2.x * gam, (n0.y * alp + n1.y * bet) + n2.y * gam,
                                    (n0.z * alp + n1.z * bet) + n2.z * gam});
    nx = n.x; ny = n.y; nz = n.z;
    px = (v0.x * alp + v1.x * bet) + v2.x * gam; py = (v0.y * alp + v1.y * bet) + v2.y * gam; pz = (v0.z * alp + v1.z * bet) + v2.z * gam;
  } else {
    const RtSphere& sp = S.spheres[-2 - h.prim];
    px = eye.x + dir.x * h.t; py = eye.y + dir.y * h.t; pz = eye.z + dir.z * h.t;
    const RtVec3 n = rt_normalized({px - sp.c[0], py - sp.c[1], pz - sp.c[2]});
    nx = n.x; ny = n.y; nz = n.z;
  }
  P.x = px; P.y = py; P.z = pz; N.x = nx; N.y = ny; N.z = nz;
}

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

// Launchers (one per kernel family); all asynchronous on `stream`.
void rt_launch_lane_kernel(const RtDeviceView& view, const RowMap& rows, bool bounce, bool count, uint8_t* rgb, int32_t* ids,
                           float* ts, unsigned long long* counters, bool tally, hipStream_t stream);
void rt_launch_wave_kernel(const RtDeviceView& view, const RowMap& rows, bool bounce, bool count, uint8_t* rgb, int32_t* ids,
                           float* ts, unsigned long long* counters, bool tally, hipStream_t stream);

#endif
