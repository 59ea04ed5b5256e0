// rt_gather.hip -- the one exchange step of a frame rendered on several GPUs: the packed row bands of every rank are
// gathered on rank 0 over RCCL (point-to-point sends over xGMI, one per peer, in one group) and put back into
// image order by a copy kernel.  The reference renders on one GPU and ends with a single cudaMemcpy of the frame
// (main.cu:56-57); several GPUs are its listed future work (doc/GPU Raytracing.pdf p.26).
//
// RCCL is bound at run time (dlopen of librccl.so.1, the copy the process already holds if it has one - PyTorch ships
// its own - else ROCm's): a process that renders on one GPU never loads it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rt_kernels.h"

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

Rccl* rccl() {
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, []() {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      R.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (R.so) break;
    }
    if (!R.so) { R.error = std::string("cannot load librccl.so.1: ") + dlerror(); return; }
    bool ok = true;
#define BIND(field, sym) ok = ok && ((R.field = (decltype(R.field))dlsym(R.so, sym)) != nullptr)
    BIND(GetUniqueId, "ncclGetUniqueId"); BIND(CommInitRank, "ncclCommInitRank"); BIND(CommInitAll, "ncclCommInitAll");
    BIND(CommDestroy, "ncclCommDestroy"); BIND(Send, "ncclSend"); BIND(Recv, "ncclRecv"); BIND(GroupStart, "ncclGroupStart");
    BIND(GroupEnd, "ncclGroupEnd"); BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    if (!ok) R.error = "librccl.so.1 lacks a symbol of the send/receive interface";
  });
  return &R;
}

int rccl_ready(Rccl** out) {
  Rccl* R = rccl();
  if (!R->error.empty()) { rt_set_error(R->error); return RT_ERR_DEVICE; }
  *out = R;
  return RT_OK;
}

#define NCCL_TRY(R, expr)                                                                    \
  do {                                                                                       \
    ncclResult_t r_ = (expr);                                                                \
    if (r_ != ncclSuccess) {                                                                 \
      rt_set_error(std::string(#expr) + ": " + (R)->GetErrorString(r_));                     \
      return RT_ERR_DEVICE;                                                                  \
    }                                                                                        \
  } while (0)

struct DeviceScope {
  int prev = -1; bool switched = false;
  explicit DeviceScope(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceScope() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
};

// One image row per workgroup; a row is copied in 16-byte pieces when rows start on 16-byte boundaries on both sides,
// byte by byte otherwise.  Pure HBM traffic: each byte of the frame is read once and written once.
__global__ __launch_bounds__(256) void unpack_bands(const uint8_t* __restrict__ gathered, size_t tile_bytes, int nranks, int band_rows,
                                                    int row_bytes, int height, uint8_t* __restrict__ image, int wide) {
  const int y = blockIdx.x;
  if (y >= height) return;
  const int band = y / band_rows, r = band % nranks, k = band / nranks;
  const uint8_t* src = gathered + (size_t)r * tile_bytes + ((size_t)k * band_rows + (size_t)(y % band_rows)) * row_bytes;
  uint8_t* dst = image + (size_t)y * row_bytes;
  if (wide) {
    const uint4* s4 = (const uint4*)src; uint4* d4 = (uint4*)dst;
    for (int i = threadIdx.x; i < row_bytes / 16; i += 256) d4[i] = s4[i];
  } else {
    for (int i = threadIdx.x; i < row_bytes; i += 256) dst[i] = src[i];
  }
}

}  // namespace

struct rt_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1, device = 0;
};

extern "C" int rt_comm_unique_id(void* id128) {
  if (!id128) { rt_set_error("rt_comm_unique_id: NULL argument"); return RT_ERR_ARG; }
  Rccl* R; int rc = rccl_ready(&R);
  if (rc != RT_OK) return rc;
  static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "id size");
  ncclUniqueId id;
  NCCL_TRY(R, R->GetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return RT_OK;
}

extern "C" int rt_comm_init_rank(const void* id128, int nranks, int rank, int device, rt_comm** out) {
  if (!id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) { rt_set_error("rt_comm_init_rank: bad argument"); return RT_ERR_ARG; }
  *out = nullptr;
  Rccl* R; int rc = rccl_ready(&R);
  if (rc != RT_OK) return rc;
  DeviceScope g(device);
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  rt_comm* c = new rt_comm();
  c->rank = rank; c->nranks = nranks; c->device = device;
  const ncclResult_t r = R->CommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) { rt_set_error(std::string("ncclCommInitRank: ") + R->GetErrorString(r)); delete c; return RT_ERR_DEVICE; }
  *out = c;
  return RT_OK;
}

extern "C" int rt_comm_init_all(int ndev, const int* devices, rt_comm** out) {
  if (ndev < 1 || !devices || !out) { rt_set_error("rt_comm_init_all: bad argument"); return RT_ERR_ARG; }
  for (int i = 0; i < ndev; i++) out[i] = nullptr;
  Rccl* R; int rc = rccl_ready(&R);
  if (rc != RT_OK) return rc;
  std::vector<ncclComm_t> comms(ndev, nullptr);
  NCCL_TRY(R, R->CommInitAll(comms.data(), ndev, devices));
  for (int i = 0; i < ndev; i++) {
    out[i] = new rt_comm();
    out[i]->comm = comms[i]; out[i]->rank = i; out[i]->nranks = ndev; out[i]->device = devices[i];
  }
  return RT_OK;
}

extern "C" void rt_comm_destroy(rt_comm* c) {
  if (!c) return;
  Rccl* R = rccl();
  if (R->error.empty() && c->comm) { DeviceScope g(c->device); (void)R->CommDestroy(c->comm); }
  delete c;
}

extern "C" int rt_comm_group_begin(void) {
  Rccl* R; int rc = rccl_ready(&R);
  if (rc != RT_OK) return rc;
  NCCL_TRY(R, R->GroupStart());
  return RT_OK;
}

extern "C" int rt_comm_group_end(void) {
  Rccl* R; int rc = rccl_ready(&R);
  if (rc != RT_OK) return rc;
  NCCL_TRY(R, R->GroupEnd());
  return RT_OK;
}

extern "C" int rt_gather_bands(rt_comm* c, const uint8_t* d_local, size_t bytes_per_rank, uint8_t* d_gathered, void* stream) {
  if (!c || !d_local || (c->rank == 0 && !d_gathered)) { rt_set_error("rt_gather_bands: NULL argument"); return RT_ERR_ARG; }
  if (bytes_per_rank == 0) return RT_OK;
  Rccl* R; int rc = rccl_ready(&R);
  if (rc != RT_OK) return rc;
  DeviceScope g(c->device);
  hipStream_t s = (hipStream_t)stream;
  // Every rank sends its tile to rank 0 (rank 0 to itself: a device copy inside RCCL); rank 0 posts one receive per
  // peer.  All of it is one group, so the transfers of the seven peers run side by side, each over its own xGMI link.
  NCCL_TRY(R, R->GroupStart());
  ncclResult_t r = R->Send(d_local, bytes_per_rank, ncclUint8, 0, c->comm, s);
  if (r == ncclSuccess && c->rank == 0)
    for (int k = 0; k < c->nranks && r == ncclSuccess; k++) r = R->Recv(d_gathered + (size_t)k * bytes_per_rank, bytes_per_rank, ncclUint8, k, c->comm, s);
  const ncclResult_t e = R->GroupEnd();
  if (r != ncclSuccess || e != ncclSuccess) {
    rt_set_error(std::string("rt_gather_bands: ") + R->GetErrorString(r != ncclSuccess ? r : e));
    return RT_ERR_DEVICE;
  }
  return RT_OK;
}

extern "C" int rt_unpack_bands(const uint8_t* d_gathered, size_t bytes_per_rank, int nranks, int band_rows, int width, int height,
                               int bytes_per_pixel, uint8_t* d_image, void* stream) {
  if (!d_gathered || !d_image || nranks < 1 || band_rows < 1 || width < 1 || height < 1 || bytes_per_pixel < 1) {
    rt_set_error("rt_unpack_bands: bad argument");
    return RT_ERR_ARG;
  }
  const size_t row_bytes = (size_t)width * bytes_per_pixel;
  const int bands = (height + band_rows - 1) / band_rows, per_rank = (bands + nranks - 1) / nranks;
  if (bytes_per_rank < (size_t)per_rank * band_rows * row_bytes) { rt_set_error("rt_unpack_bands: tiles smaller than rank 0's bands"); return RT_ERR_ARG; }
  const int wide = (row_bytes % 16 == 0) && (bytes_per_rank % 16 == 0) && ((uintptr_t)d_gathered % 16 == 0) && ((uintptr_t)d_image % 16 == 0);
  hipLaunchKernelGGL(unpack_bands, dim3(height), dim3(256), 0, (hipStream_t)stream, d_gathered, bytes_per_rank, nranks, band_rows, (int)row_bytes,
                     height, d_image, wide);
  if (hipGetLastError() != hipSuccess) { rt_set_error("rt_unpack_bands: launch failed"); return RT_ERR_DEVICE; }
  return RT_OK;
}

// One frame on the n GPUs of one process: every device renders its bands on a stream of its own (one host thread per
// device, so that launches and counter read-backs do not wait for each other), the tiles are gathered on the first
// device, put back into image order there and copied to the host.  With n = 1 the gather is a device copy of the
// one tile and the result is rt_render_rows_host's.
extern "C" int rt_render_frame_gathered(rt_dscene* const* scenes, rt_comm* const* comms, int n, int band_rows, uint8_t* h_rgb,
                                        unsigned flags, rt_stats* stats) {
  if (!scenes || !comms || n < 1 || !h_rgb || band_rows < 1 || band_rows % RT_TILE_H) { rt_set_error("rt_render_frame_gathered: bad argument"); return RT_ERR_ARG; }
  rt_scene_info info;
  std::vector<int> dev(n, 0);
  for (int r = 0; r < n; r++) {
    if (!scenes[r] || !comms[r] || comms[r]->nranks != n || comms[r]->rank != r) { rt_set_error("rt_render_frame_gathered: scenes[r] / comms[r] must be rank r of n"); return RT_ERR_ARG; }
    rt_scene_info ir;
    int rc = rt_dscene_get_info(scenes[r], &ir, &dev[r]);
    if (rc != RT_OK) return rc;
    if (r == 0) info = ir;
    else if (ir.width != info.width || ir.height != info.height) { rt_set_error("rt_render_frame_gathered: the scenes differ in size"); return RT_ERR_ARG; }
    if (dev[r] != comms[r]->device) { rt_set_error("rt_render_frame_gathered: scenes[r] and comms[r] sit on different devices"); return RT_ERR_ARG; }
  }
  const int W = info.width, H = info.height;
  const int nb_max = rt_band_count(scenes[0], band_rows, 0, n);
  const size_t tile_bytes = (size_t)nb_max * band_rows * W * 3, frame_bytes = (size_t)W * H * 3;
  struct PerDevice { hipStream_t stream = nullptr; uint8_t* local = nullptr; int rc = RT_OK; std::string err; rt_stats st; };
  std::vector<PerDevice> P(n);
  uint8_t *gathered = nullptr, *image = nullptr;
  auto cleanup = [&]() {
    for (int r = 0; r < n; r++) { DeviceScope g(dev[r]); if (P[r].stream) (void)hipStreamDestroy(P[r].stream); (void)hipFree(P[r].local); }
    DeviceScope g(dev[0]); (void)hipFree(gathered); (void)hipFree(image);
  };
  auto fail = [&](int rc, const std::string& msg) { if (!msg.empty()) rt_set_error(msg); cleanup(); return rc; };
  for (int r = 0; r < n; r++) {
    DeviceScope g(dev[r]);
    if (hipStreamCreateWithFlags(&P[r].stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&P[r].local, tile_bytes) != hipSuccess)
      return fail(RT_ERR_DEVICE, "rt_render_frame_gathered: stream / tile allocation failed");
  }
  {
    DeviceScope g(dev[0]);
    if (hipMalloc((void**)&gathered, tile_bytes * n) != hipSuccess || hipMalloc((void**)&image, frame_bytes) != hipSuccess)
      return fail(RT_ERR_DEVICE, "rt_render_frame_gathered: frame allocation failed");
  }
  const unsigned render_flags = flags & ~(unsigned)RT_RENDER_TIME;
  const auto t0 = std::chrono::steady_clock::now();
  {
    std::vector<std::thread> th;
    auto work = [&](int r) {
      P[r].rc = rt_render_bands(scenes[r], band_rows, r, n, P[r].local, nullptr, nullptr, P[r].stream, render_flags, stats ? &P[r].st : nullptr);
      if (P[r].rc != RT_OK) P[r].err = rt_last_error();
    };
    for (int r = 1; r < n; r++) th.emplace_back(work, r);
    work(0);
    for (auto& t : th) t.join();
  }
  for (int r = 0; r < n; r++) if (P[r].rc != RT_OK) return fail(P[r].rc, P[r].err);
  int rc = rt_comm_group_begin();
  if (rc != RT_OK) return fail(rc, "");
  for (int r = 0; r < n && rc == RT_OK; r++) rc = rt_gather_bands(comms[r], P[r].local, tile_bytes, gathered, P[r].stream);
  const int rc_end = rt_comm_group_end();
  if (rc != RT_OK || rc_end != RT_OK) return fail(rc != RT_OK ? rc : rc_end, "");
  {
    DeviceScope g(dev[0]);
    rc = rt_unpack_bands(gathered, tile_bytes, n, band_rows, W, H, 3, image, P[0].stream);
    if (rc != RT_OK) return fail(rc, "");
    if (hipStreamSynchronize(P[0].stream) != hipSuccess) return fail(RT_ERR_DEVICE, "rt_render_frame_gathered: the gather failed");
    const float frame_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const auto t1 = std::chrono::steady_clock::now();
    if (hipMemcpy(h_rgb, image, frame_bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(RT_ERR_DEVICE, "rt_render_frame_gathered: copy to the host failed");
    if (stats) {
      rt_stats sum; memset(&sum, 0, sizeof sum);
      for (int r = 0; r < n; r++) {
        const uint64_t* a = (const uint64_t*)&P[r].st; uint64_t* b = (uint64_t*)&sum;
        for (size_t k = 0; k < offsetof(rt_stats, kernel_ms) / sizeof(uint64_t); k++) b[k] += a[k];
      }
      sum.kernel_ms = frame_ms;   // first launch to the last tile in place on the first device
      sum.copy_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t1).count();
      *stats = sum;
    }
  }
  for (int r = 1; r < n; r++) { DeviceScope g(dev[r]); (void)hipStreamSynchronize(P[r].stream); }
  cleanup();
  return RT_OK;
}

// ---- frames in flight ------------------------------------------------------------------------------------------------
// A pipeline that is set up once and then renders frame after frame without creating anything: per slot (a frame in
// flight) and device a render context on the uploaded scene (rt_dscene_fork), a stream and a band tile; on the first
// device the gathered tiles, the frame in image order and a page-locked host copy of it, and two events.  A short frame
// on one GPU of several ends on a few long waves while most of the chip idles; a second and third frame in flight fill
// that tail, which is why the pipeline exists.
struct rt_frame_ctx {
  int n = 0, band_rows = 0, W = 0, H = 0;
  size_t tile_bytes = 0, frame_bytes = 0;
  std::vector<int> dev;
  std::vector<rt_comm*> comms;      // empty: one device, no exchange
  struct Slot {
    std::vector<rt_dscene*> ctx;
    std::vector<hipStream_t> stream;
    std::vector<uint8_t*> local;    // (n > 1, or a communicator was given: the device's packed bands)
    uint8_t* gathered = nullptr; uint8_t* image = nullptr; uint8_t* host = nullptr;
    hipEvent_t begin = nullptr, end = nullptr;
    bool busy = false, on_device = false;
  };
  std::vector<Slot> slot;
};

extern "C" void rt_frame_ctx_destroy(rt_frame_ctx* f) {
  if (!f) return;
  for (auto& S : f->slot) {
    for (size_t r = 0; r < S.stream.size(); r++) {
      DeviceScope g(f->dev[r]);
      if (S.stream[r]) { (void)hipStreamSynchronize(S.stream[r]); (void)hipStreamDestroy(S.stream[r]); }
      if (r < S.local.size()) (void)hipFree(S.local[r]);
      if (r < S.ctx.size() && S.ctx[r]) rt_dscene_destroy(S.ctx[r]);
    }
    DeviceScope g(f->dev.empty() ? 0 : f->dev[0]);
    (void)hipFree(S.gathered); (void)hipFree(S.image);
    if (S.host) (void)hipHostFree(S.host);
    if (S.begin) (void)hipEventDestroy(S.begin);
    if (S.end) (void)hipEventDestroy(S.end);
  }
  delete f;
}

extern "C" int rt_frame_ctx_create(rt_dscene* const* scenes, rt_comm* const* comms, int n, int band_rows, int slots, rt_frame_ctx** out) {
  if (!scenes || !out || n < 1 || slots < 1 || slots > 8 || band_rows < 1 || band_rows % RT_TILE_H || (n > 1 && !comms)) {
    rt_set_error("rt_frame_ctx_create: bad argument (1..8 slots, band_rows a multiple of 16, a communicator per device when there are several)");
    return RT_ERR_ARG;
  }
  *out = nullptr;
  std::unique_ptr<rt_frame_ctx, void (*)(rt_frame_ctx*)> f(new rt_frame_ctx(), rt_frame_ctx_destroy);
  f->n = n; f->band_rows = band_rows; f->dev.assign(n, 0);
  rt_scene_info info;
  for (int r = 0; r < n; r++) {
    if (!scenes[r]) { rt_set_error("rt_frame_ctx_create: NULL scene"); return RT_ERR_ARG; }
    rt_scene_info ir;
    const int rc = rt_dscene_get_info(scenes[r], &ir, &f->dev[r]);
    if (rc != RT_OK) return rc;
    if (r == 0) info = ir;
    else if (ir.width != info.width || ir.height != info.height) { rt_set_error("rt_frame_ctx_create: the scenes differ in size"); return RT_ERR_ARG; }
    if (comms) {
      if (!comms[r] || comms[r]->nranks != n || comms[r]->rank != r || comms[r]->device != f->dev[r]) {
        rt_set_error("rt_frame_ctx_create: comms[r] must be rank r of n on the device of scenes[r]"); return RT_ERR_ARG;
      }
      f->comms.push_back(comms[r]);
    }
  }
  f->W = info.width; f->H = info.height;
  const int nb_max = rt_band_count(scenes[0], band_rows, 0, n);
  f->tile_bytes = (size_t)nb_max * band_rows * f->W * 3; f->frame_bytes = (size_t)f->W * f->H * 3;
  const bool exchange = !f->comms.empty();
  f->slot.resize(slots);
  for (auto& S : f->slot) {
    S.ctx.assign(n, nullptr); S.stream.assign(n, nullptr); S.local.assign(exchange ? n : 0, nullptr);
    for (int r = 0; r < n; r++) {
      DeviceScope g(f->dev[r]);
      const int rc = rt_dscene_fork(scenes[r], &S.ctx[r]);
      if (rc != RT_OK) return rc;
      if (hipStreamCreateWithFlags(&S.stream[r], hipStreamNonBlocking) != hipSuccess) { rt_set_error("rt_frame_ctx_create: stream creation failed"); return RT_ERR_DEVICE; }
      if (exchange && hipMalloc((void**)&S.local[r], f->tile_bytes) != hipSuccess) { rt_set_error("rt_frame_ctx_create: tile allocation failed"); return RT_ERR_NOMEM; }
    }
    DeviceScope g(f->dev[0]);
    if ((exchange && hipMalloc((void**)&S.gathered, f->tile_bytes * n) != hipSuccess) || hipMalloc((void**)&S.image, f->frame_bytes) != hipSuccess ||
        hipHostMalloc((void**)&S.host, f->frame_bytes, hipHostMallocDefault) != hipSuccess) { rt_set_error("rt_frame_ctx_create: frame allocation failed"); return RT_ERR_NOMEM; }
    if (hipEventCreate(&S.begin) != hipSuccess || hipEventCreate(&S.end) != hipSuccess) { rt_set_error("rt_frame_ctx_create: event creation failed"); return RT_ERR_DEVICE; }
  }
  *out = f.release();
  return RT_OK;
}

extern "C" int rt_frame_enqueue(rt_frame_ctx* f, int slot, unsigned flags) {
  if (!f || slot < 0 || slot >= (int)f->slot.size()) { rt_set_error("rt_frame_enqueue: bad argument"); return RT_ERR_ARG; }
  rt_frame_ctx::Slot& S = f->slot[slot];
  if (S.busy) { rt_set_error("rt_frame_enqueue: the slot's last frame has not been waited for"); return RT_ERR_ARG; }
  const unsigned render_flags = flags & ~(unsigned)(RT_RENDER_TIME | RT_FRAME_ON_DEVICE);
  { DeviceScope g(f->dev[0]); if (hipEventRecord(S.begin, S.stream[0]) != hipSuccess) { rt_set_error("rt_frame_enqueue: event record failed"); return RT_ERR_DEVICE; } }
  int rc = RT_OK;
  if (f->comms.empty()) {   // one device, no exchange: straight into the frame
    rc = rt_render_rows(S.ctx[0], 0, f->H, S.image, nullptr, nullptr, S.stream[0], render_flags, nullptr);
    if (rc != RT_OK) return rc;
  } else {
    for (int r = 0; r < f->n; r++) {
      rc = rt_render_bands(S.ctx[r], f->band_rows, r, f->n, S.local[r], nullptr, nullptr, S.stream[r], render_flags, nullptr);
      if (rc != RT_OK) return rc;
    }
    rc = rt_comm_group_begin();
    if (rc != RT_OK) return rc;
    for (int r = 0; r < f->n && rc == RT_OK; r++) rc = rt_gather_bands(f->comms[r], S.local[r], f->tile_bytes, S.gathered, S.stream[r]);
    const int rc_end = rt_comm_group_end();
    if (rc != RT_OK || rc_end != RT_OK) return rc != RT_OK ? rc : rc_end;
    rc = rt_unpack_bands(S.gathered, f->tile_bytes, f->n, f->band_rows, f->W, f->H, 3, S.image, S.stream[0]);
    if (rc != RT_OK) return rc;
  }
  DeviceScope g(f->dev[0]);
  S.on_device = (flags & RT_FRAME_ON_DEVICE) != 0;
  if ((!S.on_device && hipMemcpyAsync(S.host, S.image, f->frame_bytes, hipMemcpyDeviceToHost, S.stream[0]) != hipSuccess) ||
      hipEventRecord(S.end, S.stream[0]) != hipSuccess) { rt_set_error("rt_frame_enqueue: copy to the host failed"); return RT_ERR_DEVICE; }
  S.busy = true;
  return RT_OK;
}

extern "C" int rt_frame_wait(rt_frame_ctx* f, int slot, const uint8_t** h_rgb, float* frame_ms) {
  if (!f || slot < 0 || slot >= (int)f->slot.size()) { rt_set_error("rt_frame_wait: bad argument"); return RT_ERR_ARG; }
  rt_frame_ctx::Slot& S = f->slot[slot];
  if (!S.busy) { rt_set_error("rt_frame_wait: nothing was enqueued in the slot"); return RT_ERR_ARG; }
  DeviceScope g(f->dev[0]);
  if (hipEventSynchronize(S.end) != hipSuccess) { rt_set_error("rt_frame_wait: the frame failed"); return RT_ERR_DEVICE; }
  S.busy = false;
  if (frame_ms) { float ms = 0.f; if (hipEventElapsedTime(&ms, S.begin, S.end) != hipSuccess) ms = 0.f; *frame_ms = ms; }
  if (h_rgb) *h_rgb = S.on_device ? nullptr : S.host;
  return RT_OK;
}

extern "C" int rt_frame_device_image(rt_frame_ctx* f, int slot, uint8_t** d_rgb) {
  if (!f || !d_rgb || slot < 0 || slot >= (int)f->slot.size()) { rt_set_error("rt_frame_device_image: bad argument"); return RT_ERR_ARG; }
  *d_rgb = f->slot[slot].image;
  return RT_OK;
}
