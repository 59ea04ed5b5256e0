// Issue cost of the instruction forms the traversal loops are made of, on gfx950 (wave64), at the occupancies the render
// kernel runs at: 8 independent chains per wave, N waves per SIMD on all 1024 SIMDs.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/issue_rate2.hip -o build/issue_rate2 && build/issue_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

enum { M_MUL_VV, M_SUB_SV, M_FMA_VVV, M_FMA_SVV, M_FMAC_VV, M_FMAC_SV, M_MAX_VV, M_MAX3_VVV, M_CMP_VCC, M_CMP_SGPR, M_CNDMASK,
       M_PK_FMA, M_PK_MUL, M_SMOV, M_SAND64, M_READLANE, M_FMA_SVV_NEG, M_MIX_SUBMUL, M_MIX_FMA, M_MED3, M_XOR, M_COUNT };

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float k, float sk_in) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = k * 0.5f, c = k * 0.25f;
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
  const v2f kk = {k, k};
  const float sk = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sk_in)));
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (MODE == M_MUL_VV) {
#define X(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
        asm volatile(REP8(X) : OPS : "v"(k));
#undef X
      } else if (MODE == M_SUB_SV) {
#define X(n) "v_sub_f32 %" #n ", %8, %" #n "\n"
        asm volatile(REP8(X) : OPS : "s"(sk));
#undef X
      } else if (MODE == M_FMA_VVV) {
#define X(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
        asm volatile(REP8(X) : OPS : "v"(k), "v"(b));
#undef X
      } else if (MODE == M_FMA_SVV) {
#define X(n) "v_fma_f32 %" #n ", %8, %" #n ", %9\n"
        asm volatile(REP8(X) : OPS : "s"(sk), "v"(b));
#undef X
      } else if (MODE == M_FMA_SVV_NEG) {
#define X(n) "v_fma_f32 %" #n ", %8, %9, -%" #n "\n"
        asm volatile(REP8(X) : OPS : "s"(sk), "v"(b));
#undef X
      } else if (MODE == M_FMAC_VV) {
#define X(n) "v_fmac_f32 %" #n ", %8, %9\n"
        asm volatile(REP8(X) : OPS : "v"(k), "v"(b));
#undef X
      } else if (MODE == M_FMAC_SV) {
#define X(n) "v_fmac_f32 %" #n ", %8, %9\n"
        asm volatile(REP8(X) : OPS : "s"(sk), "v"(b));
#undef X
      } else if (MODE == M_MAX_VV) {
#define X(n) "v_max_f32 %" #n ", %" #n ", %8\n"
        asm volatile(REP8(X) : OPS : "v"(k));
#undef X
      } else if (MODE == M_MAX3_VVV) {
#define X(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
        asm volatile(REP8(X) : OPS : "v"(k), "v"(b));
#undef X
      } else if (MODE == M_MED3) {
#define X(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
        asm volatile(REP8(X) : OPS : "v"(k), "v"(b));
#undef X
      } else if (MODE == M_XOR) {
#define X(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
        asm volatile(REP8(X) : OPS : "v"(k));
#undef X
      } else if (MODE == M_CMP_VCC) {
#define X(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
        asm volatile(REP8(X) : OPS : "v"(k) : "vcc");
#undef X
      } else if (MODE == M_CMP_SGPR) {
#define X(n) "v_cmp_lt_f32_e64 s[20:21], %" #n ", %8\n"
        asm volatile(REP8(X) : OPS : "v"(k) : "s20", "s21");
#undef X
      } else if (MODE == M_CNDMASK) {
#define X(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
        asm volatile(REP8(X) : OPS : "v"(k) : "vcc");
#undef X
      } else if (MODE == M_PK_FMA) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(kk), "v"(kk));
      } else if (MODE == M_PK_MUL) {
        asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                     "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(kk));
      } else if (MODE == M_SMOV) {
        asm volatile("s_mov_b32 s20, s21\n s_mov_b32 s22, s23\n s_mov_b32 s24, s25\n s_mov_b32 s26, s27\n"
                     "s_mov_b32 s21, s20\n s_mov_b32 s23, s22\n s_mov_b32 s25, s24\n s_mov_b32 s27, s26\n" ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
      } else if (MODE == M_SAND64) {
        asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]\n s_andn2_b64 s[24:25], s[24:25], s[22:23]\n s_or_b64 s[26:27], s[26:27], s[22:23]\n s_and_b64 s[28:29], s[28:29], s[22:23]\n"
                     "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_andn2_b64 s[24:25], s[24:25], s[22:23]\n s_or_b64 s[26:27], s[26:27], s[22:23]\n s_and_b64 s[28:29], s[28:29], s[22:23]\n"
                     ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc");
      } else if (MODE == M_READLANE) {
#define X(n) "v_readlane_b32 s2" #n ", %" #n ", 3\n"
        asm volatile(REP8(X) : OPS : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
#undef X
      } else if (MODE == M_MIX_SUBMUL) {   // the scan loop's present form: 4 x (sub from SGPR, mul)
        asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n"
                     "v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %9\n v_mul_f32 %3, %3, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sk), "v"(k));
      } else if (MODE == M_MIX_FMA) {      // the same four results by fused multiply-adds on the SGPR operand
        asm volatile("v_fma_f32 %0, %8, %9, -%0\n v_fma_f32 %1, %8, %9, -%1\n v_fma_f32 %2, %8, %9, -%2\n v_fma_f32 %3, %8, %9, -%3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sk), "v"(k));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + c;
}

template <int MODE>
static void run(const char* name, int waves_per_simd, int instr_per_iter) {
  const int blocks = 256 * waves_per_simd, iters = 10000;   // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
  float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<MODE><<<blocks, 256>>>(out, 100, 1.0001f, 0.999f);
  hipEventRecord(a); probe<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.999f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double inst = (double)iters * instr_per_iter * waves_per_simd;   // per SIMD
  printf("%-34s %d waves/SIMD: %7.3f ms, %5.2f cycles per instruction per SIMD at 2.4 GHz\n", name, waves_per_simd, ms, ms * 1e6 / inst * 2.4);
  hipFree(out);
}

#define RUN(M, name, n) run<M>(name, w, n)
int main() {
  for (int w : {2, 6}) {
    RUN(M_MUL_VV, "v_mul_f32 v,v", 64); RUN(M_SUB_SV, "v_sub_f32 s,v", 64); RUN(M_MAX_VV, "v_max_f32 v,v", 64); RUN(M_XOR, "v_xor_b32 v,v", 64);
    RUN(M_FMA_VVV, "v_fma_f32 v,v,v", 64); RUN(M_FMA_SVV, "v_fma_f32 s,v,v", 64); RUN(M_FMA_SVV_NEG, "v_fma_f32 s,v,-v", 64);
    RUN(M_FMAC_VV, "v_fmac_f32 v,v", 64); RUN(M_FMAC_SV, "v_fmac_f32 s,v", 64);
    RUN(M_MAX3_VVV, "v_max3_f32 v,v,v", 64); RUN(M_MED3, "v_med3_f32 v,v,v", 64);
    RUN(M_CMP_VCC, "v_cmp_lt_f32 vcc", 64); RUN(M_CMP_SGPR, "v_cmp_lt_f32_e64 sgpr", 64); RUN(M_CNDMASK, "v_cndmask_b32", 64);
    RUN(M_PK_MUL, "v_pk_mul_f32", 64); RUN(M_PK_FMA, "v_pk_fma_f32", 64);
    RUN(M_SMOV, "s_mov_b32", 64); RUN(M_SAND64, "s_and/andn2/or_b64", 64); RUN(M_READLANE, "v_readlane_b32", 64);
    RUN(M_MIX_SUBMUL, "4x(v_sub s,v ; v_mul) = 8 instr", 64); RUN(M_MIX_FMA, "4x v_fma s,v,-v = same results", 32);
  }
  return 0;
}
