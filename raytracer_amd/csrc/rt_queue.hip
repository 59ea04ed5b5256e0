// rt_queue.hip -- the small kernels around the generation-by-generation rendering of bounce rays (RtQueues): the
// coherence order of a queue's rays (a counting sort whose size lives on the device), the bookkeeping between two
// generations, and the kernel that adds up a pixel's term slots in the order of the reference's ray stack.
// All of them are sized for the chip, not for the data: they read how many rays or blocks there are from RtQState and
// stride over them, so a frame's launches are enqueued without the host ever learning a count.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "rt_kernels.h"

namespace {

constexpr uint32_t QBLOCK = 256;
constexpr uint32_t QGRID = 1024;   // 4 workgroups per compute unit

__device__ __forceinline__ uint32_t spread3(uint32_t v) {   // up to 9 bits -> every third bit
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
static_assert(RT_QCELL_BITS <= 9u, "spread3 handles nine bits");

__device__ __forceinline__ uint32_t ray_count(const RtQState* state, uint32_t in_index, uint32_t queue_cap) {
  const uint32_t n = state->count[in_index];
  return n < queue_cap ? n : queue_cap;
}

// Lanes of the wave that hold the same key add to the key's counter once: rays that sit side by side in a queue were
// pushed by neighbouring pixels and mostly share their cell, and a counter that every ray of a queue hit on its own
// would serialise at the memory side.  Returns, for a scatter, the position the lane's ray gets (base of the run that
// the first lane of the group reserved + the lane's rank in the group) when `want_pos` is set.
// The groups are found first, with ballots alone; then the first lanes of ALL groups send their additions in one
// instruction.  An addition whose result is needed is a round trip to the memory side (the counters are shared by
// the dies), well over a microsecond; one after the other, group by group, that was what the scatter's time consisted of.
__device__ __forceinline__ uint32_t grouped_add(uint32_t* counters, uint32_t key, bool active, bool want_pos) {
  const int lane = (int)(threadIdx.x & 63u);
  uint64_t todo = __builtin_amdgcn_ballot_w64(active);
  uint32_t rank = 0u, size = 0u;
  int first = lane;
  while (todo) {
    const int leader = __builtin_ctzll(todo);
    const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
    const uint64_t same = __builtin_amdgcn_ballot_w64(active && key == k);
    if (active && key == k) { first = leader; rank = (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull)); size = (uint32_t)__builtin_popcountll(same); }
    todo &= ~same;
  }
  uint32_t base = 0u;
  if (active && first == lane) base = atomicAdd(&counters[key], size);
  if (!want_pos) return 0u;
  base = (uint32_t)__shfl((int)base, first, 64);
  return base + rank;
}

// Floats as unsigned integers of the same order (for atomicMin / atomicMax).
__device__ __forceinline__ uint32_t ordered(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float unordered(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// Pass 0: the box around the rays' origins, or around a sample of them (the grid of the keys is laid over it: however large
// the scene, the cells are a 16th of what the rays of this generation span).
constexpr uint32_t BGRID = 256;
__global__ __launch_bounds__(QBLOCK) void q_bounds(const RtQRay* __restrict__ rays, RtQState* __restrict__ state, uint32_t in_index, uint32_t queue_cap) {
  const uint32_t n = ray_count(state, in_index, queue_cap);
  if (n < RT_QUEUE_SORT_MIN) return;
  uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
  // Every ray costs the pass a 64-byte line for its 12 bytes of origin (1.2 M rays: 77 MB, 19 us), and the box only places a grid
  // whose border cells take whatever lies outside: of a long queue every eighth ray is looked at.
  const uint32_t step = n >= 65536u ? 8u : 1u;
  for (uint32_t i = (blockIdx.x * QBLOCK + threadIdx.x) * step; i < n; i += BGRID * QBLOCK * step)
    for (int k = 0; k < 3; k++) {
      const float e = rays[i].e[k];
      if (e - e == 0.f) { const uint32_t u = ordered(e); lo[k] = u < lo[k] ? u : lo[k]; hi[k] = u > hi[k] ? u : hi[k]; }   // (finite ones)
    }
  __shared__ uint32_t part[6][QBLOCK / 64];
  for (int k = 0; k < 3; k++) {
    uint32_t a = lo[k], b = hi[k];
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t a2 = (uint32_t)__shfl_down((int)a, off, 64), b2 = (uint32_t)__shfl_down((int)b, off, 64);
      a = a2 < a ? a2 : a; b = b2 > b ? b2 : b;
    }
    if ((threadIdx.x & 63u) == 0u) { part[k][threadIdx.x >> 6] = a; part[3 + k][threadIdx.x >> 6] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 6u) {
    const bool is_lo = threadIdx.x < 3u;
    uint32_t v = part[threadIdx.x][0];
    for (uint32_t w = 1; w < QBLOCK / 64; w++) { const uint32_t x = part[threadIdx.x][w]; v = is_lo ? (x < v ? x : v) : (x > v ? x : v); }
    if (is_lo) atomicMin(&state->box_lo[threadIdx.x], v); else atomicMax(&state->box_hi[threadIdx.x - 3u], v);
  }
}

// Pass 1: the key of every ray, and how many rays each key has.
__global__ __launch_bounds__(QBLOCK) void q_keys(const RtQRay* __restrict__ rays, const RtQState* __restrict__ state, uint32_t in_index, uint32_t queue_cap, uint32_t octant_first_min,
                                                 uint32_t dir_grid, uint32_t cell_bits, uint32_t* __restrict__ keys, uint32_t* __restrict__ bins) {
  const uint32_t n = ray_count(state, in_index, queue_cap);
  if (n < RT_QUEUE_SORT_MIN) return;
  float box_lo[3], inv_cell[3];
  for (int k = 0; k < 3; k++) {
    const uint32_t a = state->box_lo[k], b = state->box_hi[k];
    box_lo[k] = 0.f; inv_cell[k] = 0.f;
    if (a < b) {
      box_lo[k] = unordered(a);
      const float side = unordered(b) - box_lo[k];
      if (side > 0.f && side - side == 0.f) inv_cell[k] = (float)(1u << (dir_grid ? cell_bits : RT_QCELL_BITS)) / side;
    }
  }
  // with a million rays a generation grouping by octant first gives the better waves, with fewer grouping by origin first
  // (measured on the glass bunny at 4K and at 720p / 1080p)
  const bool octant_first = n >= octant_first_min;
  const uint32_t rounds = (n + QGRID * QBLOCK - 1u) / (QGRID * QBLOCK);
  for (uint32_t r = 0; r < rounds; r++) {   // whole waves take part in every round (the grouping ballots)
    const uint32_t i = (r * QGRID + blockIdx.x) * QBLOCK + threadIdx.x;
    const bool active = i < n;
    uint32_t key = 0u;
    if (active) {
      const RtQRay& q = rays[i];
      uint32_t cell[3];
      for (int k = 0; k < 3; k++) {
        const float g = (q.e[k] - box_lo[k]) * inv_cell[k];
        const float top = (float)((1u << (dir_grid ? cell_bits : RT_QCELL_BITS)) - 1u);
        cell[k] = g > 0.f ? (g < top ? (uint32_t)g : (uint32_t)top) : 0u;   // (NaN -> 0)
      }
      const uint32_t oct = (q.d[0] < 0.f ? 1u : 0u) | (q.d[1] < 0.f ? 2u : 0u) | (q.d[2] < 0.f ? 4u : 0u);
      const uint32_t morton = spread3(cell[0]) | (spread3(cell[1]) << 1) | (spread3(cell[2]) << 2);
      key = octant_first ? ((oct << (3u * RT_QCELL_BITS)) | morton) : ((morton << 3) | oct);
      if (dir_grid) {
        // direction first, finely: the face of the cube around the origin that the ray leaves through and a dir_grid x dir_grid
        // grid on that face whose lines include the face's axes, so that the rays of a bin agree on their signs; then the cell
        const float ad[3] = {fabsf(q.d[0]), fabsf(q.d[1]), fabsf(q.d[2])};
        const int ax = (ad[0] >= ad[1] && ad[0] >= ad[2]) ? 0 : (ad[1] >= ad[2] ? 1 : 2);
        const float inv_major = ad[ax] > 0.f ? 1.f / ad[ax] : 0.f;
        const float u = q.d[(ax + 1) % 3] * inv_major, v = q.d[(ax + 2) % 3] * inv_major;   // in [-1, 1]
        const float half = 0.5f * (float)dir_grid;
        int ub = (int)((u + 1.f) * half), vb = (int)((v + 1.f) * half);
        ub = ub < 0 ? 0 : (ub >= (int)dir_grid ? (int)dir_grid - 1 : ub); vb = vb < 0 ? 0 : (vb >= (int)dir_grid ? (int)dir_grid - 1 : vb);
        const uint32_t face = (uint32_t)ax * 2u + (q.d[ax] < 0.f ? 1u : 0u);
        key = (((face * dir_grid + (uint32_t)ub) * dir_grid + (uint32_t)vb) << (3u * cell_bits)) | morton;
      }
      keys[i] = key;
    }
    grouped_add(bins, key, active, false);
  }
}

// Exclusive prefix sum of the bins, in three steps: sums of blocks of RT_QBIN_BLOCK bins, prefix of those sums (one
// workgroup), prefix inside every block.  Afterwards bins[k] is the position of key k's first ray.
__global__ __launch_bounds__(QBLOCK) void q_scan_blocks(const uint32_t* __restrict__ bins, uint32_t* __restrict__ sums) {
  __shared__ uint32_t part[QBLOCK];
  const uint32_t b = blockIdx.x;
  uint32_t s = 0u;
  for (uint32_t k = threadIdx.x; k < RT_QBIN_BLOCK; k += QBLOCK) s += bins[b * RT_QBIN_BLOCK + k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t off = QBLOCK / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[b] = part[0];
}

__global__ __launch_bounds__(1024) void q_scan_sums(uint32_t* __restrict__ sums, uint32_t blocks) {
  // RT_QBIN_BLOCKS values, 1024 threads: each thread owns a run of consecutive values
  __shared__ uint32_t part[1024];
  constexpr uint32_t PER = (RT_QBIN_BLOCKS + 1023u) / 1024u;
  uint32_t local[PER];
  uint32_t s = 0u;
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t i = threadIdx.x * PER + k;
    local[k] = i < blocks ? sums[i] : 0u;
    s += local[k];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024u; off <<= 1) {   // inclusive scan of the threads' totals
    const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - s;
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t i = threadIdx.x * PER + k;
    if (i < blocks) sums[i] = run;
    run += local[k];
  }
}

__global__ __launch_bounds__(QBLOCK) void q_scan_bins(uint32_t* __restrict__ bins, const uint32_t* __restrict__ sums) {
  __shared__ uint32_t part[QBLOCK];
  constexpr uint32_t PER = RT_QBIN_BLOCK / QBLOCK;
  const uint32_t base = blockIdx.x * RT_QBIN_BLOCK + threadIdx.x * PER;
  uint32_t local[PER];
  uint32_t s = 0u;
  for (uint32_t k = 0; k < PER; k++) { local[k] = bins[base + k]; s += local[k]; }
  part[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t off = 1; off < QBLOCK; off <<= 1) {
    const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = sums[blockIdx.x] + part[threadIdx.x] - s;
  for (uint32_t k = 0; k < PER; k++) { bins[base + k] = run; run += local[k]; }
}

// Pass 2: every ray takes the next free position of its key.
__global__ __launch_bounds__(QBLOCK) void q_scatter(const RtQState* __restrict__ state, uint32_t in_index, uint32_t queue_cap,
                                                    const uint32_t* __restrict__ keys, uint32_t* __restrict__ bins, uint32_t* __restrict__ order) {
  const uint32_t n = ray_count(state, in_index, queue_cap);
  if (n < RT_QUEUE_SORT_MIN) return;
  const uint32_t rounds = (n + QGRID * QBLOCK - 1u) / (QGRID * QBLOCK);
  for (uint32_t r = 0; r < rounds; r++) {
    const uint32_t i = (r * QGRID + blockIdx.x) * QBLOCK + threadIdx.x;
    const bool active = i < n;
    const uint32_t key = active ? keys[i] : 0u;
    const uint32_t pos = grouped_add(bins, key, active, true);
    if (active) order[pos] = i;
  }
}

// Between two launches of a frame: the queue that was filled is about to be rendered by a launch with waves for `launch_rays` rays.
__global__ void q_advance(RtQState* state, uint32_t clear_index, uint32_t launch_rays) {
  const uint32_t n = state->count[clear_index ^ 1u];
  if (n > state->longest) state->longest = n;
  if (n > launch_rays) state->grid_short = 1u;
  state->count[clear_index] = 0u;
  for (int k = 0; k < 3; k++) { state->box_lo[k] = 0xffffffffu; state->box_hi[k] = 0u; }
}

// One thread per block of term slots (camera.cu:110-111: colour = colour + att * ray colour, starting from 0, in the
// order the stack processes the rays).
__global__ __launch_bounds__(QBLOCK) void q_resolve(const RtQState* __restrict__ state, const RtTermRGB* __restrict__ terms,
                                                    const uint32_t* __restrict__ block_pixel, uint32_t block_cap, uint32_t tag,
                                                    const RtSlotOrder order, uint8_t* __restrict__ rgb) {
  uint32_t n = state->n_blocks;
  n = n < block_cap ? n : block_cap;
  for (uint32_t b = blockIdx.x * QBLOCK + threadIdx.x; b < n; b += QGRID * QBLOCK) {
    float r = 0.f, g = 0.f, bl = 0.f;
    for (uint32_t k = 0; k < order.n; k++) {
      const RtTermRGB t = terms[(size_t)order.slot[k] * block_cap + b];
      if (t.tag == tag) { r = r + t.r; g = g + t.g; bl = bl + t.b; }
    }
    uint8_t* dst = rgb + (size_t)block_pixel[b] * 3;
    dst[0] = rt_quantise(r); dst[1] = rt_quantise(g); dst[2] = rt_quantise(bl);
  }
}

}  // namespace

void rt_launch_ray_order(const RtQRay* rays, RtQState* state, uint32_t in_index, uint32_t queue_cap, uint32_t octant_first_min, uint32_t dir_grid, uint32_t cell_bits,
                         uint32_t* keys, uint32_t* bins, uint32_t* sums, uint32_t* order, hipStream_t stream) {
  hipLaunchKernelGGL(q_bounds, dim3(BGRID), dim3(QBLOCK), 0, stream, rays, state, in_index, queue_cap);
  // (only the blocks of bins that this order's keys reach are cleared and summed: 393 216 of 2 M with the default grids)
  const uint64_t used = dir_grid ? ((uint64_t)6 * dir_grid * dir_grid << (3u * cell_bits)) : (uint64_t)RT_QBINS;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>(RT_QBIN_BLOCKS, (used + RT_QBIN_BLOCK - 1u) / RT_QBIN_BLOCK);
  (void)hipMemsetAsync(bins, 0, (size_t)blocks * RT_QBIN_BLOCK * sizeof(uint32_t), stream);
  hipLaunchKernelGGL(q_keys, dim3(QGRID), dim3(QBLOCK), 0, stream, rays, (const RtQState*)state, in_index, queue_cap, octant_first_min, dir_grid, cell_bits, keys, bins);
  hipLaunchKernelGGL(q_scan_blocks, dim3(blocks), dim3(QBLOCK), 0, stream, bins, sums);
  hipLaunchKernelGGL(q_scan_sums, dim3(1), dim3(1024), 0, stream, sums, blocks);
  hipLaunchKernelGGL(q_scan_bins, dim3(blocks), dim3(QBLOCK), 0, stream, bins, sums);
  hipLaunchKernelGGL(q_scatter, dim3(QGRID), dim3(QBLOCK), 0, stream, (const RtQState*)state, in_index, queue_cap, keys, bins, order);
}

int rt_preload_queue_kernels() {
  hipFuncAttributes attr;   // asking for a kernel's attributes makes the runtime load the code object it is in
  hipError_t e = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&q_advance));
  if (e == hipSuccess) e = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&q_resolve));
  return (int)e;
}

void rt_launch_queue_advance(RtQState* state, uint32_t clear_index, uint32_t launch_rays, hipStream_t stream) {
  hipLaunchKernelGGL(q_advance, dim3(1), dim3(1), 0, stream, state, clear_index, launch_rays);
}

void rt_launch_resolve_blocks(const RtQueues& q, const RtSlotOrder& order, uint8_t* rgb, hipStream_t stream) {
  hipLaunchKernelGGL(q_resolve, dim3(QGRID), dim3(QBLOCK), 0, stream, q.state, q.terms, q.block_pixel, q.block_cap, q.tag, order, rgb);
}
