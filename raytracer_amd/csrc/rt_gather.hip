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
