// rt_device.hip -- device half of the C ABI: scene upload, kernel dispatch, counters, timing.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "rt_kernels.h"

// Buffers of the generation-by-generation rendering of bounce rays, allocated on first use and grown on demand.
struct WavefrontBuffers {
  RtQRay* queue[2] = {nullptr, nullptr};
  unsigned long long* keys[2] = {nullptr, nullptr};
  RtTermRGB* vals[2] = {nullptr, nullptr};
  uint32_t* ray_keys[2] = {nullptr, nullptr};   // coherence keys of a queue's rays, unsorted / sorted
  uint32_t* ray_index[2] = {nullptr, nullptr};  // identity / the sorted order
  void* ray_sort_temp = nullptr; size_t ray_sort_temp_bytes = 0;
  uint32_t* counts = nullptr;        // [0], [1]: fill of queue[0], queue[1]; [2]: terms
  void* sort_temp = nullptr; size_t sort_temp_bytes = 0;
  uint32_t queue_cap = 0, term_cap = 0;
};

struct rt_dscene {
  int device = 0;
  RtDeviceView view;
  std::vector<void*> allocs;
  unsigned long long* d_counters = nullptr;   // RT_COUNTER_COPIES x RT_COUNTER_STRIDE x u64
  rt_scene_info info;
  int32_t max_pushes_per_hit = 0, dielectric_mesh = 0;
  WavefrontBuffers wf;
};

namespace {

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

extern "C" const char* rt_version(void) { return "rt_amd 1.5 gfx950 (wave-synchronous traversal, exact division-free predicates)"; }

extern "C" void rt_dscene_destroy(rt_dscene* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  for (void* p : d->allocs) (void)hipFree(p);
  for (int k = 0; k < 2; k++) {
    (void)hipFree(d->wf.queue[k]); (void)hipFree(d->wf.keys[k]); (void)hipFree(d->wf.vals[k]);
    (void)hipFree(d->wf.ray_keys[k]); (void)hipFree(d->wf.ray_index[k]);
  }
  (void)hipFree(d->wf.ray_sort_temp);
  (void)hipFree(d->wf.counts); (void)hipFree(d->wf.sort_temp);
  delete d;
}

extern "C" int rt_scene_upload(const rt_scene* s, int device, rt_dscene** out) {
  if (!s || !out) { rt_set_error("rt_scene_upload: NULL argument"); return RT_ERR_ARG; }
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) { rt_set_error("no HIP device is visible; the render path has no CPU fallback"); return RT_ERR_DEVICE; }
  if (device < 0 || device >= n_dev) { rt_set_error("device index out of range"); return RT_ERR_ARG; }
  HIP_TRY(hipSetDevice(device));
  static_assert(sizeof(RtNode) == 32 && sizeof(RtWaveNode) == 32 && sizeof(RtTriGeom) == 48 && sizeof(RtTriShade) == 64 && sizeof(RtTriLeaf) == 64, "record sizes");
  rt_dscene* d = new rt_dscene();
  d->device = device; d->info = s->info; d->max_pushes_per_hit = s->max_pushes_per_hit; d->dielectric_mesh = s->dielectric_mesh;
  RtDeviceView& V = d->view; memset(&V, 0, sizeof V);
  V.cam = s->cam;
  int rc;
#define UP(field, vec) if ((rc = upload(d, vec, (const void**)&V.field)) != RT_OK) { rt_dscene_destroy(d); return rc; }
  UP(nodes, s->nodes) UP(wnodes, s->wnodes) UP(leaf_refs, s->leaf_refs) UP(tri_geom, s->tri_geom) UP(tri_leaf, s->tri_leaf) UP(tri_shade, s->tri_shade)
  UP(spheres, s->spheres) UP(lights, s->lights) UP(materials, s->materials) UP(mat_index, s->mat_index)
  UP(mesh_mat_first, s->mesh_mat_first) UP(mesh_mat_count, s->mesh_mat_count)
#undef UP
  V.n_spheres = (int32_t)s->spheres.size(); V.n_lights = (int32_t)s->lights.size(); V.has_oct = s->nodes.empty() ? 0 : 1; V.n_wnodes = (int32_t)s->nodes.size();
  void* c = nullptr;
  if (hipMalloc(&c, RT_COUNTER_COPIES * RT_COUNTER_STRIDE * sizeof(unsigned long long)) != hipSuccess) { rt_dscene_destroy(d); rt_set_error("hipMalloc(counters) failed"); return RT_ERR_DEVICE; }
  d->allocs.push_back(c); d->d_counters = (unsigned long long*)c;
  *out = d;
  return RT_OK;
}

// Bounce rays generation by generation (see RtQueues).  Eligible when the reference's 10-entry stack can never overflow,
// so that leaving the stack out changes nothing: a ray at depth k <= D-1 finds at most k (p-1) older entries and pushes
// p more (D = ray depth limit, p = most pushes of one hit).  *done = false: the caller renders the stacked way instead
// (not eligible, or a queue would have overflowed).
static int render_wavefront(rt_dscene* d, const RowMap& rows, bool count, uint8_t* d_rgb, int32_t* d_ids, float* d_t, bool tally,
                            hipStream_t stream, bool* done) {
  *done = false;
  const int D = d->view.cam.max_depth, p = d->max_pushes_per_hit;
  const unsigned long long n_out = (unsigned long long)d->view.cam.width * (rows.compact ? rows.n_local : d->view.cam.height);
  const int path_bits = 4 * std::max(D, 1);
  if (p < 1 || p > RT_PATH_MAX_PUSHES || D > 9 || (D - 1) * (p - 1) + p > RT_RAY_STACK || n_out >= (1ull << (64 - path_bits))) return RT_OK;
  WavefrontBuffers& B = d->wf;
  const unsigned long long n_px = (unsigned long long)d->view.cam.width * rows.n_local;
  const unsigned long long want_q = std::max<unsigned long long>(1u << 16, 2 * n_px), want_t = std::max<unsigned long long>(1u << 16, 3 * n_px);
  if (want_q > 0xffffffffull / 2 || want_t > 0xffffffffull / 2) return RT_OK;
  if (!B.counts) { HIP_TRY(hipMalloc((void**)&B.counts, 4 * sizeof(uint32_t))); }
  if (B.queue_cap < want_q) {
    for (int k = 0; k < 2; k++) { (void)hipFree(B.queue[k]); B.queue[k] = nullptr; }
    B.queue_cap = 0;
    for (int k = 0; k < 2; k++) HIP_TRY(hipMalloc((void**)&B.queue[k], want_q * sizeof(RtQRay)));
    for (int k = 0; k < 2; k++) {
      (void)hipFree(B.ray_keys[k]); (void)hipFree(B.ray_index[k]); B.ray_keys[k] = nullptr; B.ray_index[k] = nullptr;
      HIP_TRY(hipMalloc((void**)&B.ray_keys[k], want_q * sizeof(uint32_t)));
      HIP_TRY(hipMalloc((void**)&B.ray_index[k], want_q * sizeof(uint32_t)));
    }
    (void)hipFree(B.ray_sort_temp); B.ray_sort_temp = nullptr;
    size_t rb = 0;
    HIP_TRY(rt_sort_u32(nullptr, &rb, B.ray_keys[0], B.ray_keys[1], B.ray_index[0], B.ray_index[1], (uint32_t)want_q, 30u, stream));
    HIP_TRY(hipMalloc(&B.ray_sort_temp, rb ? rb : 16));
    B.ray_sort_temp_bytes = rb;
    B.queue_cap = (uint32_t)want_q;
  }
  if (B.term_cap < want_t) {
    for (int k = 0; k < 2; k++) { (void)hipFree(B.keys[k]); (void)hipFree(B.vals[k]); B.keys[k] = nullptr; B.vals[k] = nullptr; }
    (void)hipFree(B.sort_temp); B.sort_temp = nullptr; B.sort_temp_bytes = 0; B.term_cap = 0;
    for (int k = 0; k < 2; k++) {
      HIP_TRY(hipMalloc((void**)&B.keys[k], want_t * sizeof(unsigned long long)));
      HIP_TRY(hipMalloc((void**)&B.vals[k], want_t * sizeof(RtTermRGB)));
    }
    size_t bytes = 0;
    HIP_TRY(rt_sort_terms(nullptr, &bytes, B.keys[0], B.keys[1], B.vals[0], B.vals[1], (uint32_t)want_t, 64u, stream));
    HIP_TRY(hipMalloc(&B.sort_temp, bytes ? bytes : 16));
    B.sort_temp_bytes = bytes; B.term_cap = (uint32_t)want_t;
  }
  HIP_TRY(hipMemsetAsync(B.counts, 0, 4 * sizeof(uint32_t), stream));
  RtQueues q;
  q.in = nullptr; q.order = nullptr; q.n_in = 0; q.out = B.queue[0]; q.out_count = B.counts + 0; q.out_cap = B.queue_cap;
  q.term_keys = B.keys[0]; q.term_vals = B.vals[0]; q.term_count = B.counts + 2; q.term_cap = B.term_cap; q.enabled = 1; q.path_bits = path_bits; q.rays_per_wave = 64;
  rt_launch_wave_kernel(d->view, rows, true, count, d_rgb, d_ids, d_t, d->d_counters, tally, q, stream);
  HIP_TRY(hipGetLastError());
  uint32_t h[4];
  int cur = 0;
  const float root_lo = -RT_INF / 4, root_side = RT_INF / 2;   // the octree's root cube (main.cu:25-31)
  for (int gen = 1; gen <= D + 1; gen++) {
    HIP_TRY(hipMemcpyAsync(h, B.counts, sizeof h, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (h[cur] > B.queue_cap || h[2] > B.term_cap) return RT_OK;   // would not fit: render the stacked way
    if (h[cur] == 0) break;
    HIP_TRY(hipMemsetAsync(B.counts + (cur ^ 1), 0, sizeof(uint32_t), stream));
    q.in = B.queue[cur]; q.n_in = h[cur]; q.out = B.queue[cur ^ 1]; q.out_count = B.counts + (cur ^ 1);
    q.order = nullptr;
    q.rays_per_wave = h[cur] < 131072u ? 16 : 64;   // a short queue: shorter walks in four times as many waves (measured break-even)
    static const bool sort_rays = !(getenv("RT_AMD_SORT_RAYS") && atoi(getenv("RT_AMD_SORT_RAYS")) == 0);   // measurements
    if (sort_rays && d->view.has_oct && h[cur] >= 4096) {   // the queue is put in an order that keeps similar rays together
      // measured on the glass bunny: with a million rays a generation (4K) grouping by octant first wins (9.0 -> 7.0 ms, origin
      // first 8.2); with a few hundred thousand (720p, 1080p) grouping by origin first does (4.1 -> 3.7 ms, 4.4 -> 3.9 ms)
      rt_launch_ray_keys(B.queue[cur], h[cur], root_lo, root_side, h[cur] >= 600000u, B.ray_keys[0], B.ray_index[0], stream);
      size_t rb = B.ray_sort_temp_bytes;
      HIP_TRY(rt_sort_u32(B.ray_sort_temp, &rb, B.ray_keys[0], B.ray_keys[1], B.ray_index[0], B.ray_index[1], h[cur], 30u, stream));
      q.order = B.ray_index[1];
    }
    rt_launch_wave_kernel(d->view, rows, true, count, nullptr, nullptr, nullptr, d->d_counters, tally, q, stream);
    HIP_TRY(hipGetLastError());
    cur ^= 1;
  }
  HIP_TRY(hipMemcpyAsync(h, B.counts, sizeof h, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (h[2] > B.term_cap || h[cur] != 0) return RT_OK;
  if (h[2] > 0) {
    unsigned end_bit = (unsigned)path_bits;   // only the bits that can be set are sorted on
    while (end_bit < 64 && (n_out >> (end_bit - path_bits)) != 0) end_bit++;
    size_t bytes = B.sort_temp_bytes;
    HIP_TRY(rt_sort_terms(B.sort_temp, &bytes, B.keys[0], B.keys[1], B.vals[0], B.vals[1], h[2], end_bit, stream));
    rt_launch_resolve_terms(B.keys[1], B.vals[1], h[2], path_bits, d_rgb, stream);
    HIP_TRY(hipGetLastError());
  }
  *done = true;
  return RT_OK;
}

static int launch_render(rt_dscene* d, RowMap rows, uint8_t* d_rgb, int32_t* d_ids, float* d_t, void* stream_,
                         unsigned flags, rt_stats* stats) {
  hipStream_t stream = (hipStream_t)stream_;
  HIP_TRY(hipSetDevice(d->device));
  rows.strips = (flags & RT_RENDER_STRIPS) ? 1 : 0;
  // the counters are only read back for a caller that passes an rt_stats: no need to clear them for anybody else
  if (stats) HIP_TRY(hipMemsetAsync(d->d_counters, 0, RT_COUNTER_COPIES * RT_COUNTER_STRIDE * sizeof(unsigned long long), stream));
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  const bool timed = (flags & RT_RENDER_TIME) != 0;
  if (timed) { HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1)); HIP_TRY(hipEventRecord(ev0, stream)); }
  if (rows.n_local > 0) {
    if (!(flags & RT_RENDER_LITERAL)) {
      // Small frames are cut into 4x4 tiles where that was measured to pay: heavy meshes when the 8x8 tiles do not even
      // fill the chip's wave slots (256 CUs x 4 SIMDs x 5), and scenes with mirrors or glass over a mesh, whose frame time
      // is the longest chain of dependent walks of one tile, up to about 720p.  (meshDragon 512x512: 1.50 -> 1.18 ms,
      // glass bunny 512x512: 7.8 -> 4.8 ms; light scenes and larger frames are faster on 8x8 tiles.)
      const long tiles = ((long)(d->view.cam.width + 7) / 8) * ((rows.n_local + 7) / 8);
      static const int forced = getenv("RT_AMD_TILE") ? atoi(getenv("RT_AMD_TILE")) : 0;   // 4 or 8: measurements
      const bool mesh = d->info.n_triangles > 0;
      // (a glass mesh rendered from queues: 8x8 tiles again - only its first generation is cut into tiles, and with the
      //  tiles dealt out from the middle of the picture that one is 2 to 8 % faster on the large tiles at 512x512 .. 1280x720)
      const bool from_queues = (flags & RT_RENDER_QUEUED) || (!(flags & RT_RENDER_STACKED) && d->dielectric_mesh);
      rows.small = forced ? (forced == 4)
                          : ((d->info.n_triangles >= 20000 && tiles < 5120) ||
                             (d->info.has_bounce_materials && mesh && tiles < 16384 && !from_queues));
    }
    const bool bounce = d->info.has_bounce_materials != 0, count = (flags & RT_RENDER_COUNT_TESTS) != 0;
    if (flags & RT_RENDER_LITERAL) rt_launch_lane_kernel(d->view, rows, bounce, count, d_rgb, d_ids, d_t, d->d_counters, stats != nullptr, stream);
    else {
      bool done = false;
      // Queues pay where rays scatter inside a glass mesh (glass bunny 1920x1080: 8.2 -> 4.3 ms, 512x512: 4.9 -> 3.7 ms; a tie
      // at 4K); mirrors and sphere scenes are faster on the lanes' stacks (their bounce rays stay neighbours, and the term
      // sort costs more than it saves).  RT_RENDER_QUEUED / RT_RENDER_STACKED force one way.
      const bool queued = (flags & RT_RENDER_QUEUED) || (!(flags & RT_RENDER_STACKED) && d->dielectric_mesh);
      if (bounce && queued) {
        if (stats) {   // a frame that has to be rendered again must not be counted twice
          const int rc = render_wavefront(d, rows, count, d_rgb, d_ids, d_t, true, stream, &done);
          if (rc != RT_OK) return rc;
          if (!done) HIP_TRY(hipMemsetAsync(d->d_counters, 0, RT_COUNTER_COPIES * RT_COUNTER_STRIDE * sizeof(unsigned long long), stream));
        } else {
          const int rc = render_wavefront(d, rows, count, d_rgb, d_ids, d_t, false, stream, &done);
          if (rc != RT_OK) return rc;
        }
      }
      if (!done) {
        RtQueues none; memset(&none, 0, sizeof none);
        rt_launch_wave_kernel(d->view, rows, bounce, count, d_rgb, d_ids, d_t, d->d_counters, stats != nullptr, none, stream);
      }
    }
    HIP_TRY(hipGetLastError());
  }
  float ms = 0.f;
  if (timed) {
    HIP_TRY(hipEventRecord(ev1, stream)); HIP_TRY(hipEventSynchronize(ev1));
    HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
  }
  if (stats) {
    static thread_local unsigned long long hc[RT_COUNTER_COPIES * RT_COUNTER_STRIDE];
    HIP_TRY(hipMemcpyAsync(hc, d->d_counters, sizeof hc, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    unsigned long long h[C_COUNT] = {0};
    for (int c = 0; c < RT_COUNTER_COPIES; c++) for (int k = 0; k < C_COUNT; k++) h[k] += hc[c * RT_COUNTER_STRIDE + k];
    memset(stats, 0, sizeof *stats);
    stats->rays = h[C_RAYS]; stats->shadow_rays = h[C_SHADOW]; stats->node_tests = h[C_NODE]; stats->leaf_visits = h[C_LEAF];
    stats->tri_tests = h[C_TRI]; stats->sphere_tests = h[C_SPH]; stats->ub_leave_level0 = h[C_UB_LEAVE];
    stats->ub_level_overflow = h[C_UB_OVER]; stats->ub_stack_overflow = h[C_UB_STACK];
    stats->rays_traced = (flags & RT_RENDER_LITERAL) ? h[C_RAYS] : h[C_TRACED]; stats->kernel_ms = ms;
    for (int k = 0; k < 8; k++) stats->wave_paths[k] = h[C_W_TRACE + k];
  }
  return RT_OK;
}

extern "C" int rt_render_rows(rt_dscene* d, int row_begin, int row_end, uint8_t* d_rgb, int32_t* d_ids, float* d_t,
                              void* stream, unsigned flags, rt_stats* stats) {
  if (!d || !d_rgb) { rt_set_error("rt_render_rows: NULL argument"); return RT_ERR_ARG; }
  const int H = d->view.cam.height;
  if (row_begin < 0 || row_end > H || row_begin > row_end) { rt_set_error("rt_render_rows: rows outside the image"); return RT_ERR_ARG; }
  const int n = row_end - row_begin;
  const RowMap rows = {row_begin, n > 0 ? n : 1, 0, n, 0, 0, 0};
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
  const RowMap rows = {rank * band_rows, band_rows, band_rows * nranks, mine * band_rows, 1, 0, 0};
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
