// rt_kernels.h -- declarations shared by the HIP translation units of the render path.
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>

#include "rt_internal.h"
#include "rt_math.h"

#define RT_BLOCK 256     // literal cross-check kernel (the wave kernel has its own, RT_WBLOCK)
#define RT_TILE_W 16   // its pixels per workgroup: four 8x8 wave tiles; bands are whole multiples of RT_TILE_H rows
#define RT_TILE_H 16

// Device-resident scene, passed to kernels by value.
struct RtDeviceView {
  RtCamera cam;
  const float4* nodes;        // 2 per node
  const float4* wnodes;       // 2 per node, RtWaveNode layout, RT_NODE_VARIANTS copies of n_wnodes records
  int32_t n_wnodes;           // records per copy (nodes + pad)
  const uint32_t* node_skip;  // per node: the node a stackless walk goes to after the node's subtree (0xffffffff: the walk is over)
  const int32_t* leaf_refs;
  const float4* tri_geom;     // 3 per triangle, indexed by triangle id
  const float4* tri_leaf;     // 4 per leaf reference (RtTriLeaf), in leaf order (same index space as leaf_refs)
  uint32_t cull_offset;       // bytes from tri_leaf to the RtCullRec of leaf reference 0 (same array)
  const float4* tri_shade;    // 4 per triangle
  const RtSphere* spheres;
  const RtLight* lights;
  const RtMaterial* materials;
  const int32_t* mat_index;
  const int32_t* mesh_mat_first;
  const int32_t* mesh_mat_count;
  int32_t n_spheres, n_lights, has_oct, pad;
  const float* tri_uv;          // 6 per triangle, indexed by triangle id; nullptr (like the next two) in scenes without texture materials
  const RtTexture* textures;
  const uint32_t* texels;
};

// Which image rows a launch renders and where they are stored.  Local row lr (0..n_local) is image
// row  row0 + (lr / band_rows) * band_stride + lr % band_rows ; it is stored at image row position
// when `compact` is 0 and at row lr of a band-packed buffer when it is 1.
// small != 0: a wave owns a 4x4 tile (16 of its 64 lanes carry a pixel) and a workgroup 8x8 pixels; used for frames with
// too few 8x8 tiles to fill the chip, where four times as many, shorter walks finish sooner.
struct RowMap { int32_t row0, band_rows, band_stride, n_local, compact, small, strips, every_triangle; };   // host side: strips = 1: deal the tiles to the dies in strips whatever their number; every_triangle = 1: no bundle test in front of a leaf's triangles

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

// ---- bounce rays handled generation by generation ("wavefront") ---------------------------------------------------
// A pixel's rays form a tree (a reflective hit pushes one ray, a dielectric hit up to two; ray/ray.cuh:10-33,
// material.cu:25-99).  The reference works it off a 10-entry LIFO stack per pixel and adds att * colour of every ray to
// the pixel in the order the rays are processed.  Instead of keeping that stack inside one lane (whose wave then spends
// dozens of dependent walks on a few pixels while the chip runs empty), every pushed ray can be appended to a queue and
// the queue rendered by the next launch with 64 of them per wave.
//   * A pixel whose primary hit pushes a ray gets a BLOCK of term slots, one slot per node of its ray tree: with at most
//     P pushes per hit and ray depth limit D the tree has 1 + P + ... + P^D nodes, the root in slot 0 and the k-th push
//     (k = 0 .. P-1) of the ray in slot s in slot s * P + 1 + k.  Every ray writes its contribution att * colour into
//     its own slot (each slot is written at most once per frame, by one lane: no atomics, no sort).
//   * A last kernel adds up the slots of each block in the order the reference's stack processes the rays - depth first,
//     of the rays pushed by one hit the last one first - so the float sums, and the bytes, are the reference's.
//   * A slot is valid when its fourth word carries the frame's tag; stale slots of earlier frames do not (the pool is
//     cleared when it is created and when the tag wraps), so nothing is cleared per frame.
// Everything a generation needs to know about the one before it (how many rays, how many blocks) stays on the device:
// the launches of a frame are enqueued back to back without a host round trip, and the one read-back at the end only
// tells whether a queue or the block pool overflowed (the frame is then rendered on the lanes' stacks instead).
struct RtQRay {   // 64 B
  float e[3], d[3], att[3], level[RT_MAX_LEVELS];
  int32_t nlevel, depth;
  uint32_t block;     // the pixel's term block
  uint32_t slot;      // this ray's node in the pixel's ray tree
};
static_assert(sizeof(RtQRay) == 64 && RT_MAX_LEVELS == 3, "queue record");
struct RtTermRGB { float r, g, b; uint32_t tag; };
struct RtQState {     // device-resident bookkeeping of a frame rendered from queues
  uint32_t count[2];  // rays in queue 0 / 1 (may exceed the capacity: then `overflow` is set)
  uint32_t n_blocks;  // term blocks handed out
  uint32_t longest;   // rays of the frame's longest generation so far
  uint32_t overflow;  // a queue or the block pool was full
  uint32_t box_lo[3], box_hi[3];   // box around the origins of the queue about to be rendered (floats in an order-preserving code)
  uint32_t grid_short;   // a generation held more rays than its launch had waves for: the frame is not complete
  uint32_t pad[4];
};
static_assert(sizeof(RtQState) == 64, "queue state");
#define RT_TERM_SLOTS_MAX 128u   // nodes of a pixel's ray tree beyond which bounce rays stay on the lanes' stacks
#define RT_QUEUE_SORT_MIN 4096u  // shorter queues are rendered in the order they were filled (`order` is not used)
#define RT_QUEUE_SMALL 200000u   // shorter queues are rendered with 16 rays per wave instead of 64
struct RtQueues {
  const RtQRay* in; RtQRay* out;          // in == nullptr: the pass that casts the primary rays
  const uint32_t* order;                  // ray i of the launch is in[order[i]] (nullptr: in[i])
  RtQState* state;
  RtTermRGB* terms;                       // [slot][block], block_cap blocks per slot
  uint32_t* block_pixel;                  // output position (in pixels) of a block's pixel
  uint32_t in_index, out_index;           // which of state->count belongs to `in` / `out`
  uint32_t queue_cap, block_cap;
  uint32_t pushes;                        // P: most rays one hit can push (the fan-out of the slot numbering)
  uint32_t tag;                           // of this frame's terms
  uint32_t small_below;                   // queues shorter than this are rendered with 16 rays per wave instead of 64
  uint32_t launch_rays;                   // the launch that renders `in` has waves for so many rays (<= queue_cap)
  int32_t enabled;
};
// Order in which a block's slots are added up: depth first, later pushes first (the LIFO stack's order); n <= RT_TERM_SLOTS_MAX.
struct RtSlotOrder { uint8_t slot[RT_TERM_SLOTS_MAX]; uint32_t n; };

// camera.cu:113-122: min(c, 1) * 255 truncated; NaN < 1 is false, so NaN -> 1 -> 255; negatives stored as 0
__device__ __forceinline__ uint8_t rt_quantise(float c) {
  const float v = (c < 1.f ? c : 1.f) * 255.f;
  return v < 0.f ? (uint8_t)0 : (uint8_t)(int)v;
}

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

// ---- texture materials -----------------------------------------------------------------------------------------------
// The reference declares them (resources.json `brdf_color_texture_material`, material.cuh:10 TEXTURE) but has no
// behaviour: its loader is a stub and `colorate` has no case for them.  What follows is this library's definition, the
// same float operations in the CPU restatement the tests compare with:
//   * texture coordinates of a hit: on a triangle the corners' `vt` interpolated like the normals, (uv0 alp + uv1 bet) +
//     uv2 gam; on a sphere longitude and latitude of the unit normal, u = 0.5 + atan2f(n.z, n.x) / 2pi, v = 0.5 + asinf(n.y) / pi;
//   * u and v repeat (x - floorf(x)); v = 1 is the image's top row;
//   * nearest_neighbor: texel (int)(u W), (int)((1 - v) H); bilinear: the four texels around (u W - 0.5, (1 - v) H - 0.5),
//     wrapped, blended first along x then along y; a texel's channel is its byte / 255;
//   * the colour takes the place of the material's colour in the Lambert / Blinn-Phong sums (material.cu:15-24,
//     lights.cu:260-367); the ambient light sees it only with `use_for_ambient`.
__device__ __forceinline__ void hit_uv(const float* __restrict__ tri_uv, const Hit& h, const RtVec3& N, float& u, float& v) {
  if (h.prim >= 0) {
    const float* t = tri_uv + 6 * (size_t)h.prim;
    const float alp = 1 - h.bet - h.gam;
    u = (t[0] * alp + t[2] * h.bet) + t[4] * h.gam;
    v = (t[1] * alp + t[3] * h.bet) + t[5] * h.gam;
  } else {
    u = 0.5f + atan2f(N.z, N.x) * 0.15915494f;
    v = 0.5f + asinf(N.y < -1.f ? -1.f : (N.y > 1.f ? 1.f : N.y)) * 0.31830987f;
  }
}

__device__ __forceinline__ RtVec3 texel_rgb(uint32_t w) {
  return {(float)(w & 255u) / 255.f, (float)((w >> 8) & 255u) / 255.f, (float)((w >> 16) & 255u) / 255.f};
}

__device__ __forceinline__ RtVec3 texture_colour(const RtTexture* __restrict__ textures, const uint32_t* __restrict__ texels, uint32_t index,
                                                 uint32_t flags, float u, float v) {
  const RtTexture T = textures[index];
  const uint32_t* px = texels + T.first;
  const int W = (int)T.width, H = (int)T.height;
  if (!(u - u == 0.f) || !(v - v == 0.f)) { u = 0.f; v = 0.f; }   // (NaN or infinite coordinates: the first texel)
  u = u - floorf(u); v = v - floorf(v);
  const float fx = u * (float)W, fy = (1.f - v) * (float)H;
  if (!(flags & RT_TEX_BILINEAR)) {
    int x = (int)fx, y = (int)fy;
    x = x < W ? x : W - 1; y = y < H ? y : H - 1;
    return texel_rgb(px[(size_t)y * W + x]);
  }
  const float gx = fx - 0.5f, gy = fy - 0.5f;
  const float x0f = floorf(gx), y0f = floorf(gy);
  const float tx = gx - x0f, ty = gy - y0f;
  int x0 = (int)x0f, y0 = (int)y0f;
  int x1 = x0 + 1, y1 = y0 + 1;
  x0 = x0 < 0 ? x0 + W : x0; y0 = y0 < 0 ? y0 + H : y0;
  x1 = x1 >= W ? x1 - W : x1; y1 = y1 >= H ? y1 - H : y1;
  x0 = x0 < W ? x0 : W - 1; y0 = y0 < H ? y0 : H - 1;
  const RtVec3 c00 = texel_rgb(px[(size_t)y0 * W + x0]), c10 = texel_rgb(px[(size_t)y0 * W + x1]);
  const RtVec3 c01 = texel_rgb(px[(size_t)y1 * W + x0]), c11 = texel_rgb(px[(size_t)y1 * W + x1]);
  const float sx = 1.f - tx, sy = 1.f - ty;
  const RtVec3 top = {c00.x * sx + c10.x * tx, c00.y * sx + c10.y * tx, c00.z * sx + c10.z * tx};
  const RtVec3 bot = {c01.x * sx + c11.x * tx, c01.y * sx + c11.y * tx, c01.z * sx + c11.z * tx};
  return {top.x * sy + bot.x * ty, top.y * sy + bot.y * ty, top.z * sy + bot.z * ty};
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
// wf.enabled: bounce rays go to wf.out and terms to the term list instead of the in-kernel stack; wf.in != nullptr: the
// launch renders the rays of that queue instead of primary rays (rows, rgb, ids, ts unused).
// Each loads the code object of its source file's kernels on the current device (returns a hipError_t as int): the HIP
// runtime loads a code object when one of its kernels is first asked for, and a caller that must not allocate inside a
// render call (rt_render_rows_static) wants that behind it.
int rt_preload_wave_kernels();
int rt_preload_lane_kernels();
int rt_preload_queue_kernels();
void rt_launch_wave_kernel(const RtDeviceView& view, const RowMap& rows, bool bounce, bool count, uint8_t* rgb, int32_t* ids,
                           float* ts, unsigned long long* counters, bool tally, const RtQueues& wf, hipStream_t stream);
// Coherence order of a queue's rays (any order gives the same image; waves of rays that start close together and head
// the same way are what the shared walk is good at): a counting sort on (direction class, cell of the origin on a grid
// over the box around the origins themselves), all on the device - the number of rays is read from `state`.
//   dir_grid > 0: the class is the face of the cube around the origin that the ray leaves through and a dir_grid x dir_grid
//   cell on it (dir_grid even: the rays of a class agree on their signs), the grid has 2^cell_bits cells a side;
//   dir_grid == 0: the class is the direction's octant, the grid 64^3, the octant leading from octant_first_min rays on.
//   keys: one word per ray; bins: RT_QBINS words; sums: RT_QBIN_BLOCKS words; order: the sorted ray indices.
#ifndef RT_QCELL_BITS
#define RT_QCELL_BITS 6u
#endif
#define RT_QBINS (8u << (3u * RT_QCELL_BITS))
#define RT_QBIN_BLOCK 1024u
#define RT_QBIN_BLOCKS (RT_QBINS / RT_QBIN_BLOCK)
void rt_launch_ray_order(const RtQRay* rays, RtQState* state, uint32_t in_index, uint32_t queue_cap, uint32_t octant_first_min, uint32_t dir_grid, uint32_t cell_bits,
                         uint32_t* keys, uint32_t* bins, uint32_t* sums, uint32_t* order, hipStream_t stream);
// Between two generations: the queue that was just rendered becomes the next output queue (its count is cleared), the
// box around the origins is emptied.
void rt_launch_queue_advance(RtQState* state, uint32_t clear_index, uint32_t launch_rays, hipStream_t stream);
// Adds up the valid slots of every block in the stack's order, quantises and stores the pixel.
void rt_launch_resolve_blocks(const RtQueues& q, const RtSlotOrder& order, uint8_t* rgb, hipStream_t stream);

#endif
