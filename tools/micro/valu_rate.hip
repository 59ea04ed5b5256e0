// Issue-rate probe for gfx950: wave64 v_mul_f32 vs v_pk_mul_f32 vs v_min_f32 vs s_mov (independent chains).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o build/valu_rate && build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float k) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
  const v2f kk = {k, k};
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                     "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                     "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(kk));
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        asm volatile("v_min3_f32 %0, %0, %8, %1\n v_min3_f32 %1, %1, %8, %2\n v_min3_f32 %2, %2, %8, %3\n v_min3_f32 %3, %3, %8, %4\n"
                     "v_min3_f32 %4, %4, %8, %5\n v_min3_f32 %5, %5, %8, %6\n v_min3_f32 %6, %6, %8, %7\n v_min3_f32 %7, %7, %8, %0\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
      }
    } else {   // alternate VALU and SALU from the same wave
#pragma unroll
      for (int u = 0; u < 8; u++) {
        asm volatile("v_mul_f32 %0, %0, %8\n s_add_u32 s20, s20, 1\n v_mul_f32 %1, %1, %8\n s_add_u32 s21, s21, 1\n v_mul_f32 %2, %2, %8\n s_add_u32 s22, s22, 1\n v_mul_f32 %3, %3, %8\n s_add_u32 s23, s23, 1\n"
                     "v_mul_f32 %4, %4, %8\n s_add_u32 s20, s20, 1\n v_mul_f32 %5, %5, %8\n s_add_u32 s21, s21, 1\n v_mul_f32 %6, %6, %8\n s_add_u32 s22, s22, 1\n v_mul_f32 %7, %7, %8\n s_add_u32 s23, s23, 1\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "s20", "s21", "s22", "s23", "scc");
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE>
static void run(const char* name, int waves_per_simd, int instr_per_iter) {
  const int blocks = 256 * waves_per_simd, iters = 20000;   // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
  float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<MODE><<<blocks, 256>>>(out, 100, 1.0001f);
  hipEventRecord(a); probe<MODE><<<blocks, 256>>>(out, iters, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double inst = (double)iters * instr_per_iter * waves_per_simd;   // per SIMD
  printf("%-28s %d wave(s)/SIMD: %.3f ms, %.2f ns per instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e6 / inst, ms * 1e6 / inst * 2.4);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    if (w == 1) { run<0>("v_mul_f32", 1, 64); run<1>("v_pk_mul_f32", 1, 64); run<2>("v_min3_f32", 1, 64); run<3>("v_mul_f32 + s_add_u32", 1, 128); }
    if (w == 2) { run<0>("v_mul_f32", 2, 64); run<1>("v_pk_mul_f32", 2, 64); run<2>("v_min3_f32", 2, 64); run<3>("v_mul_f32 + s_add_u32", 2, 128); }
    if (w == 4) { run<0>("v_mul_f32", 4, 64); run<1>("v_pk_mul_f32", 4, 64); run<2>("v_min3_f32", 4, 64); run<3>("v_mul_f32 + s_add_u32", 4, 128); }
  }
  return 0;
}
