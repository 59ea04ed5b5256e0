// rt_device.hip -- device half of the C ABI: scene upload, render contexts, kernel dispatch, counters, timing.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "rt_kernels.h"

namespace {

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      rt_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                       \
      return RT_ERR_DEVICE;                                                                  \
    }                                                                                        \
  } while (0)

// Makes `device` current for the lifetime of the guard and puts the caller's device back afterwards: a process that
// drives several GPUs (or has torch on another device) is left where it was by every entry point.
struct DeviceGuard {
  int prev = -1; bool switched = false; hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) { err = hipSetDevice(device); switched = err == hipSuccess; }
  }
  ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// A pair of timing events that is destroyed on every way out of the scope.
struct EventPair {
  hipEvent_t a = nullptr, b = nullptr;
  ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};

// The uploaded scene arrays.  Shared by every render context opened on them (rt_dscene_fork) and freed with the last.
struct SceneArrays {
  int device = 0;
  RtDeviceView view;
  std::vector<void*> allocs;
  rt_scene_info info;
  int32_t max_pushes_per_hit = 0, dielectric_mesh = 0;
  ~SceneArrays() {
    DeviceGuard g(device);
    for (void* p : allocs) (void)hipFree(p);
  }
};

template <typename T>
int upload(SceneArrays* d, const std::vector<T>& v, const void** out) {
  *out = nullptr;
  if (v.empty()) return RT_OK;
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
  d->allocs.push_back(p);
  HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = p;
  return RT_OK;
}

// Scratch memory of a render: device memory beyond the scene and the caller's outputs.  One arena, carved per call:
//   [0, RT_COUNTER_BYTES)   the work counters (every render);
//   behind them             the buffers of the generation-by-generation rendering of bounce rays (glass meshes only).
// A context's arena is hipMalloc'ed and grows on demand; the arena of the entry point that must not allocate is a
// fixed array in the code object (see rt_render_rows_static).
constexpr size_t RT_COUNTER_BYTES = RT_COUNTER_COPIES * RT_COUNTER_STRIDE * sizeof(unsigned long long);

struct Arena {
  unsigned char* base = nullptr;
  size_t bytes = 0;
  bool fixed = false;   // true: never grown, never freed
};

// Offsets of the queue-mode buffers inside an arena (all 256-byte aligned).
struct WavefrontLayout {
  size_t state, queue[2], keys, order, bins, sums, block_pixel, terms, terms_bytes, total;
  uint32_t queue_cap, block_cap, slots;
};

size_t carve(size_t& cursor, size_t bytes) {
  const size_t at = cursor;
  cursor += (bytes + 255) & ~(size_t)255;
  return at;
}

void wavefront_layout(uint32_t queue_cap, uint32_t block_cap, uint32_t slots, WavefrontLayout* L) {
  size_t c = RT_COUNTER_BYTES;
  L->queue_cap = queue_cap; L->block_cap = block_cap; L->slots = slots;
  L->state = carve(c, sizeof(RtQState));
  for (int k = 0; k < 2; k++) L->queue[k] = carve(c, (size_t)queue_cap * sizeof(RtQRay));
  L->keys = carve(c, (size_t)queue_cap * sizeof(uint32_t));
  L->order = carve(c, (size_t)queue_cap * sizeof(uint32_t));
  L->bins = carve(c, (size_t)RT_QBINS * sizeof(uint32_t));
  L->sums = carve(c, (size_t)RT_QBIN_BLOCKS * sizeof(uint32_t));
  L->block_pixel = carve(c, (size_t)block_cap * sizeof(uint32_t));
  L->terms_bytes = (size_t)slots * block_cap * sizeof(RtTermRGB);
  L->terms = carve(c, L->terms_bytes);
  L->total = c;
}

// Which part of an arena currently is a cleared term pool, and the tag of the last frame rendered into it.  A slot
// counts when its tag is the frame's; the tags of one pool never repeat, so nothing is cleared between frames - only when
// the pool moves or changes size (other data may have lain there) or the tag would wrap.
struct TermPool { size_t at = 0, bytes = 0; uint32_t tag = 0; };

// One frame in flight: its scratch arena, and the buffers of the host-memory entry point (device frame + pinned staging),
// kept from call to call.
struct RenderCtx {
  Arena arena;
  WavefrontLayout wf_layout; bool wf_laid_out = false;
  uint32_t wf_longest = 0u;   // rays of the longest generation of the last frame rendered from queues (0: none yet)
  uint32_t wf_attempts = 0u;  // how many times the last frame from queues was rendered (2: its launches, sized by the frame before, were too short)
  TermPool own_pool; TermPool* pool = &own_pool;   // (the fixed arena's lives with the arena, not with the context)
  uint8_t* d_rgb = nullptr; int32_t* d_ids = nullptr; float* d_t = nullptr; size_t frame_px = 0;
  void* pinned = nullptr; size_t pinned_bytes = 0;
  hipStream_t host_stream = nullptr;
  void* landing = nullptr;   // page-locked host words for read-backs (borrowed, never freed here); nullptr: pageable memory
  RenderCtx() = default;
  RenderCtx(const RenderCtx&) = delete;
  RenderCtx& operator=(const RenderCtx&) = delete;
  void release() {
    if (!arena.fixed) (void)hipFree(arena.base);
    arena = Arena(); own_pool = TermPool(); wf_laid_out = false; wf_longest = 0u;
    (void)hipFree(d_rgb); (void)hipFree(d_ids); (void)hipFree(d_t); d_rgb = nullptr; d_ids = nullptr; d_t = nullptr; frame_px = 0;
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr; pinned_bytes = 0;
    if (host_stream) (void)hipStreamDestroy(host_stream);
    host_stream = nullptr;
  }
};

int arena_reserve(Arena& a, size_t bytes, const char* what) {
  if (a.bytes >= bytes) return RT_OK;
  if (a.fixed) {
    rt_set_error(std::string(what) + ": needs " + std::to_string(bytes) + " bytes of scratch, the fixed arena holds " + std::to_string(a.bytes));
    return RT_ERR_LIMIT;
  }
  // the new arena first: when it cannot be had the old one (and the counters every launch is handed) stays as it is
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    rt_set_error(std::string(what) + ": hipMalloc of " + std::to_string(bytes) + " bytes of scratch failed");
    return RT_ERR_NOMEM;
  }
  (void)hipFree(a.base);
  a.base = (unsigned char*)p; a.bytes = bytes;
  return RT_OK;
}

// ---- scratch of the entry point that allocates nothing (rt_render_rows_static) -------------------------------------
// A fixed array in the code object: every GPU the library runs on holds its own copy, at its own address, until the
// process ends.  Host code never uses the array's name as an address: the device address is asked for once per
// device (static_scratch_setup, with that device current) and kept.
#define RT_STATIC_SCRATCH_BYTES (64u << 20)
#define RT_STATIC_MAX_DEVICES 64
#define RT_STATIC_LANDING_BYTES (RT_COUNTER_BYTES + 256)
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

struct StaticScratch {
  unsigned char* ptr = nullptr;   // device address of `scratch` on this device
  void* landing = nullptr;        // page-locked host words the counters and the queue state are copied into
  TermPool pool;                  // (the array starts out zeroed, like a pool that was just cleared nowhere: see TermPool)
  std::atomic<bool> ready{false};   // (read by rt_render_rows_static without the mutex)
};
StaticScratch g_static[RT_STATIC_MAX_DEVICES];
std::mutex g_static_mutex;

// Once per device, with that device current (called from rt_scene_upload, the library's per-device set-up).
int static_scratch_setup(int device) {
  if (device < 0 || device >= RT_STATIC_MAX_DEVICES) return RT_OK;   // such a device simply has no static entry point
  std::lock_guard<std::mutex> lock(g_static_mutex);
  StaticScratch& S = g_static[device];
  if (S.ready) return RT_OK;
  // every code object a render call can launch from is loaded now: the runtime loads one when a kernel of it is first
  // asked for, which allocates, and the static entry point may be a process's very first render
  HIP_TRY((hipError_t)rt_preload_wave_kernels());
  HIP_TRY((hipError_t)rt_preload_lane_kernels());
  HIP_TRY((hipError_t)rt_preload_queue_kernels());
  void* p = nullptr;
  HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
  if (!S.landing) HIP_TRY(hipHostMalloc(&S.landing, RT_STATIC_LANDING_BYTES, hipHostMallocDefault));
  S.ptr = (unsigned char*)p; S.ready = true;
  return RT_OK;
}

}  // namespace

struct rt_dscene {
  std::shared_ptr<SceneArrays> scene;
  RenderCtx ctx;
};

// The library's name and version; when RT_AMD_* variables are set in the environment (they move thresholds and switch
// parts of the kernels off, for measurements) they are named behind it, so that no line of a benchmark or of `run --stats`
// is ever taken with a non-default setting without saying so.
extern char** environ;
extern "C" const char* rt_version(void) {
  static const std::string text = [] {
    std::string v = "rt_amd 3.0 gfx950 (wave-synchronous traversal in gfx950 assembly, exact division-free predicates, leaf triangles ruled out per wave)";
    std::string knobs;
    for (char** e = environ; e && *e; e++)
      if (!strncmp(*e, "RT_AMD_", 7) && strncmp(*e, "RT_AMD_LIB=", 11) && strncmp(*e, "RT_AMD_BUILD_THREADS=", 21)) { knobs += knobs.empty() ? "" : " "; knobs += *e; }
    if (!knobs.empty()) v += " [non-default: " + knobs + "]";
    return v;
  }();
  return text.c_str();
}

extern "C" void rt_dscene_destroy(rt_dscene* d) {
  if (!d) return;
  {
    DeviceGuard g(d->scene->device);
    d->ctx.release();
  }
  delete d;
}

extern "C" int rt_device_prepare(int device) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) { rt_set_error("no HIP device is visible; the render path has no CPU fallback"); return RT_ERR_DEVICE; }
  if (device < 0 || device >= n_dev) { rt_set_error("device index out of range"); return RT_ERR_ARG; }
  DeviceGuard g(device);
  HIP_TRY(g.err);
  HIP_TRY(hipFree(nullptr));   // (creates the device's context)
  return static_scratch_setup(device);   // (loads the code objects of every kernel file too)
}

extern "C" int rt_scene_upload(const rt_scene* s, int device, rt_dscene** out) {
  if (!s || !out) { rt_set_error("rt_scene_upload: NULL argument"); return RT_ERR_ARG; }
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) { rt_set_error("no HIP device is visible; the render path has no CPU fallback"); return RT_ERR_DEVICE; }
  if (device < 0 || device >= n_dev) { rt_set_error("device index out of range"); return RT_ERR_ARG; }
  DeviceGuard g(device);
  HIP_TRY(g.err);
  static_assert(sizeof(RtNode) == 32 && sizeof(RtWaveNode) == 32 && sizeof(RtTriGeom) == 48 && sizeof(RtTriShade) == 64 && sizeof(RtTriLeaf) == 64, "record sizes");
  // (the walk addresses a leaf's triangle records by 32-bit byte offsets into their array)
  if (s->tri_leaf.size() * sizeof(RtTriLeaf) >= (1ull << 32)) { rt_set_error("rt_scene_upload: more than 4 GiB of leaf triangle records"); return RT_ERR_LIMIT; }
  std::unique_ptr<rt_dscene> d(new rt_dscene());
  d->scene = std::make_shared<SceneArrays>();
  SceneArrays* A = d->scene.get();
  A->device = device; A->info = s->info; A->max_pushes_per_hit = s->max_pushes_per_hit; A->dielectric_mesh = s->dielectric_mesh;
  RtDeviceView& V = A->view; memset(&V, 0, sizeof V);
  V.cam = s->cam;
  int rc;
#define UP(field, vec) if ((rc = upload(A, vec, (const void**)&V.field)) != RT_OK) return rc;
  UP(nodes, s->nodes) UP(wnodes, s->wnodes) UP(node_skip, s->node_skip) UP(leaf_refs, s->leaf_refs) UP(tri_geom, s->tri_geom) UP(tri_leaf, s->tri_leaf) UP(tri_shade, s->tri_shade)
  UP(spheres, s->spheres) UP(lights, s->lights) UP(materials, s->materials) UP(mat_index, s->mat_index)
  UP(mesh_mat_first, s->mesh_mat_first) UP(mesh_mat_count, s->mesh_mat_count)
  if (!s->textures.empty()) { UP(tri_uv, s->tri_uv) UP(textures, s->textures) UP(texels, s->texels) }   // only scenes with texture materials
#undef UP
  V.cull_offset = s->cull_offset;
  V.n_spheres = (int32_t)s->spheres.size(); V.n_lights = (int32_t)s->lights.size(); V.has_oct = s->nodes.empty() ? 0 : 1; V.n_wnodes = (int32_t)s->nodes.size();
  if ((rc = arena_reserve(d->ctx.arena, RT_COUNTER_BYTES, "rt_scene_upload")) != RT_OK) { d->ctx.release(); return rc; }
  if ((rc = static_scratch_setup(device)) != RT_OK) { d->ctx.release(); return rc; }
  *out = d.release();
  return RT_OK;
}

extern "C" int rt_dscene_get_info(const rt_dscene* d, rt_scene_info* out, int* device) {
  if (!d || !out) { rt_set_error("rt_dscene_get_info: NULL argument"); return RT_ERR_ARG; }
  *out = d->scene->info;
  if (device) *device = d->scene->device;
  return RT_OK;
}

extern "C" int rt_dscene_fork(const rt_dscene* d, rt_dscene** out) {
  if (!d || !out) { rt_set_error("rt_dscene_fork: NULL argument"); return RT_ERR_ARG; }
  *out = nullptr;
  DeviceGuard g(d->scene->device);
  HIP_TRY(g.err);
  std::unique_ptr<rt_dscene> f(new rt_dscene());
  f->scene = d->scene;
  const int rc = arena_reserve(f->ctx.arena, RT_COUNTER_BYTES, "rt_dscene_fork");
  if (rc != RT_OK) { f->ctx.release(); return rc; }
  *out = f.release();
  return RT_OK;
}

namespace {

// Bounce rays generation by generation (see RtQueues).  Eligible when the reference's 10-entry stack can never overflow,
// so that leaving the stack out changes nothing - a ray at depth k <= D-1 finds at most k (p-1) older entries and pushes
// p more (D = ray depth limit, p = most pushes of one hit) - and when a pixel's ray tree has few enough nodes for a
// block of term slots.
// Is the scene rendered from queues, and is the arena ready for it?  *eligible = false: the stacked way (ineligible scene
// or frame).  Called before anything of the frame is enqueued, because it may replace the arena.
constexpr unsigned long long RT_TERM_POOL_MAX_BYTES = 8ull << 30;   // of a context's arena; a frame that needs more blocks falls back

uint32_t tree_slots(int D, int p) {   // 1 + p + ... + p^D, saturating
  unsigned long long n = 0, level = 1;
  for (int k = 0; k <= D; k++) { n += level; level *= (unsigned long long)p; if (n > 0xffffffffull) return 0xffffffffu; }
  return (uint32_t)n;
}

int prepare_wavefront(const SceneArrays& S, RenderCtx& ctx, const RowMap& rows, hipStream_t stream, bool* eligible) {
  *eligible = false;
  const int D = S.view.cam.max_depth, p = S.max_pushes_per_hit;
  const unsigned long long n_out = (unsigned long long)S.view.cam.width * (rows.compact ? rows.n_local : S.view.cam.height);
  // (a block names its pixel by a 32-bit position in the output buffer)
  if (p < 1 || D < 1 || (D - 1) * (p - 1) + p > RT_RAY_STACK || n_out > 0xffffffffull) return RT_OK;
  const uint32_t slots = tree_slots(D, p);
  if (slots > RT_TERM_SLOTS_MAX) return RT_OK;
  unsigned long long n_px = (unsigned long long)S.view.cam.width * rows.n_local;
  // (tests: capacities as if the call had this many pixels, so that the queues overflow and the frame falls back to the stacks)
  if (const char* cap = getenv("RT_AMD_QUEUE_CAP_PIXELS")) n_px = std::min<unsigned long long>(n_px, strtoull(cap, nullptr, 10));
  // a fixed arena is cut to what the call needs; one that may grow keeps a floor so that small frames do not make it grow step by step
  const unsigned long long floor_n = (ctx.arena.fixed || getenv("RT_AMD_QUEUE_CAP_PIXELS")) ? 64ull : (1ull << 16);
  const unsigned long long want_q = std::max<unsigned long long>(floor_n, 2 * n_px);
  unsigned long long want_b = std::max<unsigned long long>(floor_n, n_px);
  if (!ctx.arena.fixed) want_b = std::min<unsigned long long>(want_b, RT_TERM_POOL_MAX_BYTES / ((unsigned long long)slots * sizeof(RtTermRGB)));
  if (want_q > 0xffffffffull / 2 || want_b > 0xffffffffull / 2) return RT_OK;
  if (!ctx.wf_laid_out || ctx.wf_layout.queue_cap < want_q || ctx.wf_layout.block_cap < want_b || ctx.wf_layout.slots != slots || ctx.arena.fixed || getenv("RT_AMD_QUEUE_CAP_PIXELS")) {
    WavefrontLayout L;
    wavefront_layout((uint32_t)want_q, (uint32_t)want_b, slots, &L);
    if (ctx.arena.bytes < L.total) {
      if (ctx.arena.fixed) {
        rt_set_error("the bounce-ray queues of " + std::to_string(n_px) + " pixels need " + std::to_string(L.total) + " bytes of scratch and the static array holds " +
                     std::to_string(ctx.arena.bytes) + ": render fewer rows per call, or pass RT_RENDER_STACKED (bounce rays on the lanes' stacks need none)");
        return RT_ERR_LIMIT;
      }
      HIP_TRY(hipStreamSynchronize(stream));   // nothing may still be using the arena that is about to be replaced
      int rc;
      if ((rc = arena_reserve(ctx.arena, L.total, "bounce-ray queues")) != RT_OK) return rc;
      *ctx.pool = TermPool();
    }
    ctx.wf_layout = L; ctx.wf_laid_out = true;
  }
  *eligible = true;
  return RT_OK;
}

// The order in which the reference's stack works off a pixel's rays: a ray, then the rays it pushed, the last one first,
// each with everything below it (ray.cuh:22-33).
void slot_order(uint32_t slot, int depth, int D, int p, RtSlotOrder* o) {
  o->slot[o->n++] = (uint8_t)slot;
  if (depth < D) for (int k = p - 1; k >= 0; k--) slot_order(slot * (uint32_t)p + 1u + (uint32_t)k, depth + 1, D, p, o);
}

// *done = false: a queue or the block pool was too small; the caller renders the stacked way instead.
// Everything is enqueued back to back - how many rays a generation holds stays on the device; the one read-back at the
// end tells whether something overflowed.
// The launches that render a queue are sized for the longest generation of the context's last frame and a margin (a full
// queue when there was none): the 260 000 waves of a full queue at 3840x2160, of which a glass-bunny generation fills 19 000, take
// 47 us to start and leave.  The device notes when a generation outgrew its launch; such a frame is rendered a second time with
// launches for full queues.
int render_wavefront(const SceneArrays& S, RenderCtx& ctx, const RowMap& rows, bool count, uint8_t* d_rgb, int32_t* d_ids, float* d_t, bool tally,
                     hipStream_t stream, bool* done) {
  *done = false;
  const int D = S.view.cam.max_depth, p = S.max_pushes_per_hit;
  const WavefrontLayout& L = ctx.wf_layout;
  unsigned char* const base = ctx.arena.base;
  unsigned long long* const counters = (unsigned long long*)base;
  RtQState* const state = (RtQState*)(base + L.state);
  RtQRay* const queue[2] = {(RtQRay*)(base + L.queue[0]), (RtQRay*)(base + L.queue[1])};
  TermPool& pool = *ctx.pool;
  if (pool.at != L.terms || pool.bytes != L.terms_bytes || pool.tag == 0xffffffffu) {
    HIP_TRY(hipMemsetAsync(base + L.terms, 0, L.terms_bytes, stream));
    pool.at = L.terms; pool.bytes = L.terms_bytes; pool.tag = 0u;
  }
  static const uint32_t small_below = getenv("RT_AMD_QUEUE_SMALL") ? (uint32_t)strtoul(getenv("RT_AMD_QUEUE_SMALL"), nullptr, 10) : RT_QUEUE_SMALL;   // measurements
  static const bool sort_rays = !(getenv("RT_AMD_SORT_RAYS") && atoi(getenv("RT_AMD_SORT_RAYS")) == 0);   // measurements
  static const uint32_t octant_first_min = getenv("RT_AMD_OCTANT_FIRST_MIN") ? (uint32_t)strtoul(getenv("RT_AMD_OCTANT_FIRST_MIN"), nullptr, 10) : 600000u;
  // The order of a queue's rays: direction class first (the face of the cube around the origin a ray leaves through, 4 x 4 cells
  // on that face), then the origin's cell on a 16^3 grid.  Measured on the glass bunny against the octant of the direction
  // and a 64^3 grid: 1920x1080 3.16 -> 2.87 ms, 512x512 2.05 -> 1.98, 4K 5.11 -> 5.08; finer directions with coarser origins
  // are slower (8 x 8 with 8^3: 5.6 ms at 4K, 32 x 32 with 4^3: 6.8): where the rays start counts for more than where they head.
  // RT_AMD_SORT_DIRGRID=0 brings the octant order back (measurements).
  static const uint32_t dir_grid = getenv("RT_AMD_SORT_DIRGRID") ? (uint32_t)strtoul(getenv("RT_AMD_SORT_DIRGRID"), nullptr, 10) : 4u;
  static const uint32_t cell_bits = getenv("RT_AMD_SORT_CELLBITS") ? (uint32_t)strtoul(getenv("RT_AMD_SORT_CELLBITS"), nullptr, 10) : 4u;
  if (dir_grid && ((unsigned long long)6 * dir_grid * dir_grid << (3u * cell_bits)) > RT_QBINS) { rt_set_error("RT_AMD_SORT_DIRGRID / RT_AMD_SORT_CELLBITS: more bins than the sort has"); return RT_ERR_ARG; }
  static const bool full_launches = getenv("RT_AMD_QUEUE_FULL_LAUNCHES") != nullptr;   // measurements
  RtSlotOrder order; order.n = 0u;
  slot_order(0u, 0, D, p, &order);
  RtQState h_pageable;
  // (a context with page-locked landing words keeps the queue state behind the counters' place there)
  RtQState* const h = ctx.landing ? (RtQState*)((unsigned char*)ctx.landing + RT_COUNTER_BYTES) : &h_pageable;
  for (int attempt = 0; attempt < 2; attempt++) {
    unsigned long long cover = L.queue_cap;
    if (attempt == 0 && ctx.wf_longest && !full_launches) cover = std::min<unsigned long long>(cover, (unsigned long long)ctx.wf_longest + ctx.wf_longest / 4u + 65536ull);
    pool.tag++;
    HIP_TRY(hipMemsetAsync(state, 0, sizeof(RtQState), stream));
    RtQueues q;
    memset(&q, 0, sizeof q);
    q.in = nullptr; q.out = queue[0]; q.order = nullptr; q.state = state;
    q.terms = (RtTermRGB*)(base + L.terms); q.block_pixel = (uint32_t*)(base + L.block_pixel);
    q.in_index = 1u; q.out_index = 0u; q.queue_cap = L.queue_cap; q.block_cap = L.block_cap; q.pushes = (uint32_t)p; q.tag = pool.tag; q.enabled = 1;
    q.small_below = small_below; q.launch_rays = (uint32_t)cover;
    rt_launch_wave_kernel(S.view, rows, true, count, d_rgb, d_ids, d_t, counters, tally, q, stream);
    HIP_TRY(hipGetLastError());
    uint32_t cur = 0u;
    for (int gen = 1; gen <= D; gen++) {   // rays of depth `gen` (camera.cu:104-112 works them off one by one)
      rt_launch_queue_advance(state, cur ^ 1u, q.launch_rays, stream);
      q.in = queue[cur]; q.in_index = cur; q.out = queue[cur ^ 1u]; q.out_index = cur ^ 1u; q.order = nullptr;
      if (sort_rays && S.view.has_oct) {   // the queue is put in an order that keeps similar rays together
        uint32_t* const ray_order = (uint32_t*)(base + L.order);
        rt_launch_ray_order(queue[cur], state, cur, L.queue_cap, octant_first_min, dir_grid, cell_bits, (uint32_t*)(base + L.keys), (uint32_t*)(base + L.bins),
                            (uint32_t*)(base + L.sums), ray_order, stream);
        q.order = ray_order;
      }
#ifdef RT_WAVE_TIMES
      rt_launch_wave_kernel(S.view, rows, true, count, nullptr, nullptr, d_t, counters, tally, q, stream);   // (diagnostic build: the waves log their lives there)
#else
      rt_launch_wave_kernel(S.view, rows, true, count, nullptr, nullptr, nullptr, counters, tally, q, stream);
#endif
      HIP_TRY(hipGetLastError());
      cur ^= 1u;
    }
    rt_launch_resolve_blocks(q, order, d_rgb, stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h, state, sizeof(RtQState), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (h->overflow || h->n_blocks > L.block_cap || h->count[0] > L.queue_cap || h->count[1] > L.queue_cap) return RT_OK;
    ctx.wf_longest = h->longest; ctx.wf_attempts = (uint32_t)attempt + 1u;
    if (!h->grid_short) { *done = true; return RT_OK; }
    // (the counters of the instrumented variant would hold both attempts; the tests that read them render on a context's first frame)
    if (count || tally) HIP_TRY(hipMemsetAsync(counters, 0, RT_COUNTER_BYTES, stream));
  }
  rt_set_error("a frame rendered with launches for full queues reports a generation longer than a queue");
  return RT_ERR_DEVICE;
}

// The render proper; the device of the scene is current.  `timed` needs two events (created here, destroyed on every way
// out); a context with a fixed arena never times (it must not create anything).
int launch_render(const SceneArrays& S, RenderCtx& ctx, RowMap rows, uint8_t* d_rgb, int32_t* d_ids, float* d_t, hipStream_t stream,
                  unsigned flags, rt_stats* stats) {
  rows.strips = (flags & RT_RENDER_STRIPS) ? 1 : 0;
  rows.every_triangle = (flags & RT_RENDER_EVERY_TRIANGLE) ? 1 : 0;
  const bool bounce = S.info.has_bounce_materials != 0, count = (flags & RT_RENDER_COUNT_TESTS) != 0;
  // Queues pay where rays scatter inside a glass mesh (glass bunny 1920x1080: 8.2 -> 4.3 ms, 512x512: 4.9 -> 3.7 ms; a tie
  // at 4K); mirrors and sphere scenes are faster on the lanes' stacks (their bounce rays stay neighbours, and the term
  // sort costs more than it saves).  RT_RENDER_QUEUED / RT_RENDER_STACKED force one way.
  // (scenes with texture materials stay on the stacks: the queue variant of the kernel is not built with texture lookups)
  bool queued = bounce && rows.n_local > 0 && !(flags & RT_RENDER_LITERAL) && !S.view.textures &&
                ((flags & RT_RENDER_QUEUED) || (!(flags & RT_RENDER_STACKED) && S.dielectric_mesh));
  if (queued) {
    const int rc = prepare_wavefront(S, ctx, rows, stream, &queued);
    if (rc != RT_OK) return rc;
  }
  if (!ctx.arena.base || ctx.arena.bytes < RT_COUNTER_BYTES) { rt_set_error("render context without a counter arena"); return RT_ERR_DEVICE; }
  unsigned long long* const d_counters = (unsigned long long*)ctx.arena.base;
  // the counters are only read back for a caller that passes an rt_stats: no need to clear them for anybody else
  if (stats) HIP_TRY(hipMemsetAsync(d_counters, 0, RT_COUNTER_BYTES, stream));
  bool used_queues = false, fell_back = false;
  EventPair ev;
  const bool timed = (flags & RT_RENDER_TIME) != 0;
  if (timed) { HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b)); HIP_TRY(hipEventRecord(ev.a, stream)); }
  if (rows.n_local > 0) {
    if (!(flags & RT_RENDER_LITERAL)) {
      // Small frames are cut into 4x4 tiles where that was measured to pay: heavy meshes up to about 1366x768 (a small frame
      // ends when its longest waves end, and the sixteen rays of a 4x4 tile form a narrower bundle, which rules out more of a
      // leaf's triangles, and walk a smaller union of subtrees: meshDragon 640x480 1.06 -> 0.71 ms, 1024x768 0.89 -> 0.75,
      // 1280x720 0.93 -> 0.73; from 1600x900 on the 8x8 tiles win, 0.86 against 0.94 ms), and scenes with mirrors or glass over
      // a mesh, whose frame time is the longest chain of dependent walks of one tile, up to about 720p (glass bunny 512x512:
      // 7.8 -> 4.8 ms on the stacks; light scenes and larger frames are faster on 8x8 tiles).
      const long tiles = ((long)(S.view.cam.width + 7) / 8) * ((rows.n_local + 7) / 8);
      static const int forced = getenv("RT_AMD_TILE") ? atoi(getenv("RT_AMD_TILE")) : 0;   // 4 or 8: measurements
      const bool mesh = S.info.n_triangles > 0;
      // (a glass mesh rendered from queues: 8x8 tiles again - only its first generation is cut into tiles, and with the
      //  tiles dealt out from the middle of the picture that one is 2 to 8 % faster on the large tiles at 512x512 .. 1280x720)
      const bool from_queues = !S.view.textures && ((flags & RT_RENDER_QUEUED) || (!(flags & RT_RENDER_STACKED) && S.dielectric_mesh));
      rows.small = forced ? (forced == 4)
                          : ((S.info.n_triangles >= 20000 && tiles < 16000) ||
                             (S.info.has_bounce_materials && mesh && tiles < 16384 && !from_queues));
    }
    if (flags & RT_RENDER_LITERAL) rt_launch_lane_kernel(S.view, rows, bounce, count, d_rgb, d_ids, d_t, d_counters, stats != nullptr, stream);
    else {
      bool done = false;
      if (queued) {
        const int rc = render_wavefront(S, ctx, rows, count, d_rgb, d_ids, d_t, stats != nullptr, stream, &done);
        if (rc != RT_OK) return rc;
        // a frame that has to be rendered again must not be counted twice
        if (!done && stats) HIP_TRY(hipMemsetAsync(d_counters, 0, RT_COUNTER_BYTES, stream));
        used_queues = done; fell_back = !done;
      }
      if (!done) {
        RtQueues none; memset(&none, 0, sizeof none);
        rt_launch_wave_kernel(S.view, rows, bounce, count, d_rgb, d_ids, d_t, d_counters, stats != nullptr, none, stream);
      }
    }
    HIP_TRY(hipGetLastError());
  }
  float ms = 0.f;
  if (timed) {
    HIP_TRY(hipEventRecord(ev.b, stream)); HIP_TRY(hipEventSynchronize(ev.b));
    HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
  }
  if (stats) {
    static thread_local unsigned long long hc_pageable[RT_COUNTER_COPIES * RT_COUNTER_STRIDE];
    unsigned long long* const hc = ctx.landing ? (unsigned long long*)ctx.landing : hc_pageable;
    HIP_TRY(hipMemcpyAsync(hc, ctx.arena.base, RT_COUNTER_BYTES, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    unsigned long long h[C_COUNT] = {0};
    for (int c = 0; c < RT_COUNTER_COPIES; c++) for (int k = 0; k < C_COUNT; k++) h[k] += hc[c * RT_COUNTER_STRIDE + k];
    memset(stats, 0, sizeof *stats);
    stats->rays = h[C_RAYS]; stats->shadow_rays = h[C_SHADOW]; stats->node_tests = h[C_NODE]; stats->leaf_visits = h[C_LEAF];
    stats->tri_tests = h[C_TRI]; stats->sphere_tests = h[C_SPH]; stats->ub_leave_level0 = h[C_UB_LEAVE];
    stats->ub_level_overflow = h[C_UB_OVER]; stats->ub_stack_overflow = h[C_UB_STACK];
    stats->rays_traced = (flags & RT_RENDER_LITERAL) ? h[C_RAYS] : h[C_TRACED]; stats->kernel_ms = ms;
    for (int k = 0; k < 8; k++) stats->wave_paths[k] = h[C_W_TRACE + k];
    stats->from_queues = used_queues ? ctx.wf_attempts : 0u; stats->queue_fallback = fell_back ? 1u : 0u;
  }
  return RT_OK;
}

int check_rows(const rt_dscene* d, const void* out, int row_begin, int row_end, const char* who) {
  if (!d || !out) { rt_set_error(std::string(who) + ": NULL argument"); return RT_ERR_ARG; }
  const int H = d->scene->view.cam.height;
  if (row_begin < 0 || row_end > H || row_begin > row_end) { rt_set_error(std::string(who) + ": rows outside the image"); return RT_ERR_ARG; }
  return RT_OK;
}

RowMap plain_rows(int row_begin, int row_end) {
  const int n = row_end - row_begin;
  return RowMap{row_begin, n > 0 ? n : 1, 0, n, 0, 0, 0};
}

}  // namespace

extern "C" int rt_render_rows(rt_dscene* d, int row_begin, int row_end, uint8_t* d_rgb, int32_t* d_ids, float* d_t,
                              void* stream, unsigned flags, rt_stats* stats) {
  int rc = check_rows(d, d_rgb, row_begin, row_end, "rt_render_rows");
  if (rc != RT_OK) return rc;
  DeviceGuard g(d->scene->device);
  HIP_TRY(g.err);
  return launch_render(*d->scene, d->ctx, plain_rows(row_begin, row_end), d_rgb, d_ids, d_t, (hipStream_t)stream, flags, stats);
}

extern "C" int rt_render_rows_static(rt_dscene* d, int row_begin, int row_end, uint8_t* d_rgb, int32_t* d_ids, float* d_t,
                                     void* stream, unsigned flags, rt_stats* stats) {
  int rc = check_rows(d, d_rgb, row_begin, row_end, "rt_render_rows_static");
  if (rc != RT_OK) return rc;
  if (flags & RT_RENDER_TIME) { rt_set_error("rt_render_rows_static: RT_RENDER_TIME creates events; time the stream with events of your own"); return RT_ERR_ARG; }
  const int device = d->scene->device;
  if (device < 0 || device >= RT_STATIC_MAX_DEVICES || !g_static[device].ready) {   // (set up by rt_scene_upload on that device)
    rt_set_error("rt_render_rows_static: no static scratch on this device");
    return RT_ERR_DEVICE;
  }
  DeviceGuard g(device);
  HIP_TRY(g.err);
  // a context on the stack around this GPU's copy of the array: nothing in it is owned, nothing is freed
  RenderCtx ctx;
  ctx.arena.base = g_static[device].ptr; ctx.arena.bytes = RT_STATIC_SCRATCH_BYTES; ctx.arena.fixed = true;
  ctx.landing = g_static[device].landing; ctx.pool = &g_static[device].pool;
  return launch_render(*d->scene, ctx, plain_rows(row_begin, row_end), d_rgb, d_ids, d_t, (hipStream_t)stream, flags, stats);
}

extern "C" int rt_band_count(const rt_dscene* d, int band_rows, int rank, int nranks) {
  if (!d || band_rows < 1 || nranks < 1 || rank < 0 || rank >= nranks) return RT_ERR_ARG;
  const int bands = (d->scene->view.cam.height + band_rows - 1) / band_rows;
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
  DeviceGuard g(d->scene->device);
  HIP_TRY(g.err);
  return launch_render(*d->scene, d->ctx, rows, d_rgb_local, d_ids_local, d_t_local, (hipStream_t)stream, flags, stats);
}

// Into host memory: the frame is rendered into device buffers that the context keeps from call to call and copied out
// on the context's own stream.  Pinned destinations (rt_host_alloc, hipHostMalloc, hipHostRegister) are written by the
// copy engine directly; pageable ones go through the context's pinned staging buffer.
extern "C" int rt_render_rows_host(rt_dscene* d, int row_begin, int row_end, uint8_t* h_rgb, int32_t* h_ids, float* h_t,
                                   unsigned flags, rt_stats* stats) {
  int rc = check_rows(d, h_rgb, row_begin, row_end, "rt_render_rows_host");
  if (rc != RT_OK) return rc;
  const int W = d->scene->view.cam.width, H = d->scene->view.cam.height;
  DeviceGuard g(d->scene->device);
  HIP_TRY(g.err);
  RenderCtx& C = d->ctx;
  const size_t px = (size_t)W * H;
  if (!C.host_stream) HIP_TRY(hipStreamCreateWithFlags(&C.host_stream, hipStreamNonBlocking));
  if (C.frame_px < px) {
    (void)hipFree(C.d_rgb); (void)hipFree(C.d_ids); (void)hipFree(C.d_t); C.d_rgb = nullptr; C.d_ids = nullptr; C.d_t = nullptr; C.frame_px = 0;
    HIP_TRY(hipMalloc((void**)&C.d_rgb, px * 3));
    C.frame_px = px;
  }
  if (h_ids && !C.d_ids) HIP_TRY(hipMalloc((void**)&C.d_ids, px * 4));
  if (h_t && !C.d_t) HIP_TRY(hipMalloc((void**)&C.d_t, px * 4));
  rt_stats local;
  rt_stats* st = stats ? stats : ((flags & RT_RENDER_TIME) ? &local : nullptr);
  rc = launch_render(*d->scene, C, plain_rows(row_begin, row_end), C.d_rgb, h_ids ? C.d_ids : nullptr, h_t ? C.d_t : nullptr, C.host_stream, flags, st);
  if (rc != RT_OK) return rc;
  const size_t off = (size_t)row_begin * W, n = (size_t)(row_end - row_begin) * W;
  EventPair ev;
  const bool timed = (flags & RT_RENDER_TIME) != 0 && stats;
  if (timed) { HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b)); HIP_TRY(hipEventRecord(ev.a, C.host_stream)); }
  auto copy_out = [&](void* dst, const void* src, size_t bytes) -> int {
    if (bytes == 0) return RT_OK;
    hipPointerAttribute_t at;
    const bool pinned_dst = hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();   // (an unregistered pointer is reported as an error; it only means "pageable")
    if (pinned_dst) { HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, C.host_stream)); return RT_OK; }
    // pageable destination: in slices through the pinned staging buffer, the CPU copying slice k out while the
    // copy engine brings in slice k+1
    const size_t slice = 4u << 20;
    if (C.pinned_bytes < 2 * slice) {
      if (C.pinned) (void)hipHostFree(C.pinned);
      C.pinned = nullptr; C.pinned_bytes = 0;
      HIP_TRY(hipHostMalloc(&C.pinned, 2 * slice, hipHostMallocDefault));
      C.pinned_bytes = 2 * slice;
    }
    hipEvent_t landed[2] = {nullptr, nullptr};
    struct Cleanup { hipEvent_t* e; ~Cleanup() { for (int k = 0; k < 2; k++) if (e[k]) (void)hipEventDestroy(e[k]); } } cleanup{landed};
    for (int k = 0; k < 2; k++) HIP_TRY(hipEventCreateWithFlags(&landed[k], hipEventDisableTiming));
    const size_t n_slices = (bytes + slice - 1) / slice;
    auto issue = [&](size_t k) -> int {
      const size_t b = std::min(slice, bytes - k * slice);
      HIP_TRY(hipMemcpyAsync((unsigned char*)C.pinned + (k & 1) * slice, (const unsigned char*)src + k * slice, b, hipMemcpyDeviceToHost, C.host_stream));
      HIP_TRY(hipEventRecord(landed[k & 1], C.host_stream));
      return RT_OK;
    };
    int r = issue(0);
    if (r != RT_OK) return r;
    for (size_t k = 0; k < n_slices; k++) {
      HIP_TRY(hipEventSynchronize(landed[k & 1]));
      if (k + 1 < n_slices && (r = issue(k + 1)) != RT_OK) return r;
      memcpy((unsigned char*)dst + k * slice, (unsigned char*)C.pinned + (k & 1) * slice, std::min(slice, bytes - k * slice));
    }
    return RT_OK;
  };
  if ((rc = copy_out(h_rgb + off * 3, C.d_rgb + off * 3, n * 3)) != RT_OK) return rc;
  if (h_ids && (rc = copy_out(h_ids + off, C.d_ids + off, n * 4)) != RT_OK) return rc;
  if (h_t && (rc = copy_out(h_t + off, C.d_t + off, n * 4)) != RT_OK) return rc;
  if (timed) HIP_TRY(hipEventRecord(ev.b, C.host_stream));
  HIP_TRY(hipStreamSynchronize(C.host_stream));
  if (timed) { float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b)); stats->copy_ms = ms; }
  return RT_OK;
}

extern "C" void* rt_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { rt_set_error("rt_host_alloc: hipHostMalloc failed"); return nullptr; }
  return p;
}

extern "C" void rt_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}
