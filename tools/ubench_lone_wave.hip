// What one wave alone on its SIMD pays for the steps of the octree walk (gfx950): the latency of a scalar load that misses
// the scalar cache, of several such loads in flight, of a vector load, and the issue interval of dependent and independent
// vector and scalar instructions.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench_lone_wave.hip -o build/ubench_lone_wave && build/ubench_lone_wave
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

typedef uint32_t u16v __attribute__((ext_vector_type(16)));
typedef uint32_t u8v __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// K chains of dependent 64-byte scalar loads in flight; every record's first word is the byte offset of the chain's next record.
template <int K>
__global__ void chase_scalar(const uint32_t* buf, const uint32_t* starts, int steps, uint64_t* out) {
  uint32_t off[K];
  for (int k = 0; k < K; k++) off[k] = __builtin_amdgcn_readfirstlane(starts[k]);
  uint64_t t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < steps; i++) {
    u16v r[K];
    for (int k = 0; k < K; k++) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(r[k]) : "s"(buf), "s"(off[k]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int k = 0; k < K; k++) { asm volatile("" : "+s"(r[k])); off[k] = r[k][0]; }
  }
  uint64_t t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = off[0]; }
}

// the same chain through the vector path: lane 0's address, one dword per lane of a 64-byte record
__global__ void chase_vector(const uint32_t* buf, uint32_t start, int steps, uint64_t* out) {
  uint32_t off = start;
  uint64_t t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < steps; i++) {
    uint32_t v = buf[off / 4 + (threadIdx.x & 15)];
    off = __builtin_amdgcn_readfirstlane(v);
  }
  uint64_t t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = off; }
}

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

__global__ void issue_rates(uint64_t* out, float seed, uint32_t sx) {
  float z = seed + 7.0f * threadIdx.x, a = seed + threadIdx.x, b = seed * 0.5f, c = 1.0f, d = 2.0f, e = 3.0f, f = 4.0f, g = 5.0f, h = 6.0f;
  uint64_t t[40];
  int n = 0;
  t[n++] = clock64();
  REP64(asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(b));)                       // dependent fused multiply-adds
  t[n++] = clock64();
  REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                    "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                    : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(z) : "v"(b));)   // independent
  t[n++] = clock64();
  REP64(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));)                           // dependent multiplies
  t[n++] = clock64();
  REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                    "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                    : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(z) : "v"(b));)
  t[n++] = clock64();
  {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p = {a, c}, q = {b, d}, r = {e, f}, s = {g, h};
    REP64(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));)                      // dependent packed multiplies
    t[n++] = clock64();
    REP8(asm volatile("v_pk_mul_f32 %0, %0, %3\n v_pk_mul_f32 %1, %1, %3\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %0, %0, %3\n"
                      "v_pk_mul_f32 %1, %1, %3\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %0, %0, %3\n v_pk_mul_f32 %1, %1, %3"
                      : "+v"(p), "+v"(r), "+v"(s) : "v"(q));)
    t[n++] = clock64();
    a += p.x + p.y + r.x + r.y + s.x + s.y;
  }
  {
    uint64_t m;
    REP64(asm volatile("v_cmp_lt_f32_e64 %0, %1, %2\n s_and_b64 %0, %0, exec\n s_cbranch_scc0 1\n s_nop 0" : "=s"(m) : "v"(a), "v"(b) : "scc");)
    t[n++] = clock64();                                                                     // compare, mask, branch (as at the end of a stage)
    a += (float)m;
  }
  {
    uint32_t x = sx | 1u, y;
    REP64(asm volatile("s_ff1_i32_b32 %1, %0\n s_lshl_b32 %1, %1, 5\n s_add_u32 %0, %0, %1" : "+s"(x), "=s"(y) :: "scc");)
    t[n++] = clock64();                                                                     // 3 dependent scalar instructions
    a += (float)x;
  }
  {
    uint32_t x;
    REP64(asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(x) : "v"(a));)
    t[n++] = clock64();                                                                     // lane reads
    a += (float)x;
  }
  {
    uint64_t m;
    REP64(asm volatile("v_cmp_lt_f32_e64 %0, %1, %2\n s_and_b64 %0, %0, exec" : "=s"(m) : "v"(a), "v"(b) : "scc");)
    t[n++] = clock64();                                                                     // compare into scalar registers, mask (no branch)
    a += (float)m;
  }
  {
    REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_cbranch_vccz 1\n s_nop 0" :: "v"(a), "v"(b) : "vcc");)
    t[n++] = clock64();                                                                     // compare into vcc, branch on vccz
  }
  {
    REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_cbranch_vccnz 1\n s_nop 0" :: "v"(a), "v"(b) : "vcc");)
    t[n++] = clock64();                                                                     // compare into vcc, branch on vccnz
  }
  {
    uint32_t x = sx;
    REP64(asm volatile("s_cmp_lg_u32 %0, 0\n s_cbranch_scc0 1\n s_nop 0" :: "s"(x) : "scc");)
    t[n++] = clock64();                                                                     // scalar compare, branch not taken
    REP64(asm volatile("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1\n s_nop 0" :: "s"(x) : "scc");)
    t[n++] = clock64();                                                                     // scalar compare, branch taken
  }
  {
    uint64_t m;
    REP64(asm volatile("v_cmp_lt_f32_e64 %0, %1, %2\n v_mul_f32 %3, %3, %2\n v_mul_f32 %4, %4, %2\n v_mul_f32 %5, %5, %2\n v_mul_f32 %6, %6, %2\n"
                       "s_and_b64 %0, %0, exec\n s_cbranch_scc0 1\n s_nop 0" : "=s"(m), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "v"(a), "v"(b) : "scc");)
    t[n++] = clock64();                                                                     // compare, 4 independent multiplies, mask, branch
    a += (float)m;
  }
  {
    uint64_t m;
    REP64(asm volatile("v_cmp_lt_f32_e64 %0, %1, %2\n v_mul_f32 %3, %3, %2\n v_mul_f32 %4, %4, %2\n v_mul_f32 %5, %5, %2\n v_mul_f32 %6, %6, %2\n"
                       "v_mul_f32 %3, %3, %2\n v_mul_f32 %4, %4, %2\n v_mul_f32 %5, %5, %2\n v_mul_f32 %6, %6, %2\n"
                       "s_and_b64 %0, %0, exec\n s_cbranch_scc0 1\n s_nop 0" : "=s"(m), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "v"(a), "v"(b) : "scc");)
    t[n++] = clock64();                                                                     // compare, 8 multiplies, mask, branch
    a += (float)m;
  }
  {
    REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)               // dependent fma, distinct operands
    t[n++] = clock64();
    REP64(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
    t[n++] = clock64();
    REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
    t[n++] = clock64();
    REP64(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %2, %2, %1" : "+v"(a), "+v"(d) : "v"(b));)  // two chains interleaved (per pair)
    t[n++] = clock64();
    REP64(asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a) : "s"(sx));)                          // dependent multiply with a scalar operand
    t[n++] = clock64();
  }
  {
    uint64_t m;
    REP64(asm volatile("v_cmp_lt_f32_e64 %0, %1, %2\n s_nop 3\n s_and_b64 %0, %0, exec\n s_cbranch_scc0 1\n s_nop 0" : "=s"(m) : "v"(a), "v"(b) : "scc");)
    t[n++] = clock64();                                                                     // compare, s_nop 3, mask, branch
    a += (float)m;
  }
  a += c + d + e + f;
  if (threadIdx.x == 0) for (int i = 0; i + 1 < n; i++) out[i] = t[i + 1] - t[i];
  if (a == 12345.f) out[39] = 1;
}

int main() {
  const size_t bytes = size_t(256) << 20, recs = bytes / 64;
  std::vector<uint32_t> h(bytes / 4, 0);
  uint32_t *d; uint64_t *o;
  CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&o, 512)); CHECK(hipMemset(o, 0, 512));
  uint64_t r[40];
  auto run = [&](const char* what, size_t span_recs) {
    // one random cycle through the records of the span
    std::vector<uint32_t> order(span_recs);
    std::iota(order.begin(), order.end(), 0u);
    std::mt19937 rng(7);
    std::shuffle(order.begin(), order.end(), rng);
    for (size_t i = 0; i < span_recs; i++) h[size_t(order[i]) * 16] = order[(i + 1) % span_recs] * 64;
    CHECK(hipMemcpy(d, h.data(), span_recs * 64, hipMemcpyHostToDevice));
    uint32_t starts[8]; for (int k = 0; k < 8; k++) starts[k] = order[(span_recs / 8) * k] * 64;
    uint32_t* ds; CHECK(hipMalloc(&ds, 32)); CHECK(hipMemcpy(ds, starts, 32, hipMemcpyHostToDevice));
    const int steps = 2000;
    for (int rep = 0; rep < 2; rep++) {
#define ONE(K) hipLaunchKernelGGL(chase_scalar<K>, 1, 64, 0, 0, d, ds, steps, o); CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(r, o, 24, hipMemcpyDeviceToHost)); \
      if (rep) printf("%-28s scalar x16, %d in flight: %7.1f cycles, %6.1f ns per round\n", what, K, double(r[0]) / steps, double(r[1]) * 10 / steps);
      ONE(1) ONE(2) ONE(4)
      hipLaunchKernelGGL(chase_vector, 1, 64, 0, 0, d, starts[0], steps, o); CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(r, o, 24, hipMemcpyDeviceToHost));
      if (rep) printf("%-28s vector 64 B             : %7.1f cycles, %6.1f ns per round\n", what, double(r[0]) / steps, double(r[1]) * 10 / steps);
    }
    CHECK(hipFree(ds));
  };
  run("span 4 KiB", 64);
  run("span 1 MiB", 16384);
  run("span 46 MiB", (size_t(46) << 20) / 64);
  run("span 256 MiB", recs);
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(issue_rates, 1, 64, 0, 0, o, 1.5f, 12345u); CHECK(hipDeviceSynchronize()); }
  CHECK(hipMemcpy(r, o, 30 * 8, hipMemcpyDeviceToHost));
  const char* names[] = {"64 dependent v_fma_f32", "64 independent v_fma_f32", "64 dependent v_mul_f32", "64 independent v_mul_f32", "64 dependent v_pk_mul_f32",
                         "64 independent v_pk_mul_f32", "64 x (v_cmp, s_and, s_cbranch)", "64 x 3 dependent scalar", "64 v_readlane",
                         "64 x (v_cmp to sgpr, s_and)", "64 x (v_cmp vcc, s_cbranch_vccz)", "64 x (v_cmp vcc, s_cbranch_vccnz)", "64 x (s_cmp, branch not taken)", "64 x (s_cmp, branch taken)",
                         "64 x (v_cmp, 4 v_mul, s_and, branch)", "64 x (v_cmp, 8 v_mul, s_and, branch)", "64 dependent v_fma a*b+c", "64 dependent v_max3", "64 dependent v_add", "64 x 2 interleaved v_mul chains",
                         "64 dependent v_mul with sgpr", "64 x (v_cmp, s_nop 3, s_and, branch)"};
  for (int i = 0; i < 22; i++) printf("%-34s %6.1f cycles each (s_memtime units)\n", names[i], double(r[i]) / 64);
  return 0;
}
