// rt_kernel_wave.hip -- wave-synchronous render kernel for gfx950.
//
// A 64-lane wave owns an 8x8 pixel tile (a 4x4 tile on small frames, or 64 / 16 queued bounce rays) and walks the
// octree ONCE for all of its rays.  The
// reference visits children in the fixed order 0..7 whatever the ray direction (ray.cu:40-50), so
// every ray of the wave meets the nodes of the union of their subtrees in the same order; the wave
// therefore keeps one traversal stack (branch, children still to test, 64-bit mask of the lanes that
// entered the branch) instead of 64.  Consequences on CDNA4:
//   * node and triangle records are addressed with wave-uniform indices, i.e. fetched through the
//     scalar cache into SGPRs once per wave, and the loop control runs on the scalar unit;
//   * a triangle is tested by all interested lanes in the same iteration, so the tests share
//     wave-level early-outs (ballot == 0 -> skip the rest of the solve);
//   * each lane still takes exactly the decisions of its own depth-first walk with its own running
//     closest t, which is what makes the result identical to the reference's.
// Neighbouring pixels of a tile traverse nearly the same subtree (measured on meshDragon at 4K: the
// union costs 1.02x the node tests and 0.80x the triangle iterations of 64 private stacks).
//
// Decisions are taken with the division-free predicates of rt_math.h; quotients are formed only for
// triangles that pass every sign / range test.  Built with -ffp-contract=off.
#include "rt_kernels.h"
#include "rt_trace_asm.inc"

namespace {

// Workgroup of this kernel: waves are independent of each other (no barrier, no shared data), so a workgroup is a
// single wave: the slot of a finished wave is then refilled at once instead of waiting for the other waves of a
// larger workgroup to finish (tiles differ a lot in cost).
#ifndef RT_WBLOCK
#define RT_WBLOCK 64
#endif
#define RT_WAVES_X (RT_WBLOCK >= 128 ? 2 : 1)
#define RT_WTILE_W (8 * RT_WAVES_X)
#define RT_WTILE_H (8 * (RT_WBLOCK / 64) / RT_WAVES_X)

#define RT_DIES 8u        // XCDs of an MI355X; workgroup i of a launch starts on die i % 8
#ifndef RT_STRIP_W
#define RT_STRIP_W 4u     // tile columns per strip
#endif

struct Tally { uint32_t rays, shadow, node, leaf, tri, sph, traced; uint32_t w[8]; };   // w: wave-level path counts (lane 0, instrumented variant)

// The wave's traversal stack lives in four VGPRs: entry i is held by lane i (depth <= 20 < 64) and is
// written with a lane-select and read with v_readlane, whose lane index is wave-uniform.  No LDS, no memory latency.
struct WaveStack {
  uint32_t first, cur, mask_lo, mask_hi;   // index of child 0 of the branch, bit mask of its children still to test, lanes that entered it
};

typedef float v2f __attribute__((ext_vector_type(2)));   // two floats handled by one packed-fp32 instruction

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Wave-level path counters (how often each part of the walk runs) are kept by the instrumented variant, and by the
// plain one too in a build with -DRT_PATH_COUNTS (measurements; off in the product).
#ifdef RT_PATH_COUNTS
#define RT_PATHS_ON true
#else
#define RT_PATHS_ON false
#endif


// Minimum of a value over the 64 lanes, the same in every lane (all lanes active): quad swaps and row mirrors, then the
// four rows through scalar registers.
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)));    // quad_perm [1,0,3,2]
  v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)));    // quad_perm [2,3,0,1]
  v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true)));   // row_half_mirror
  v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true)));   // row_mirror
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)),
              r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fminf(fminf(r0, r1), fminf(r2, r3)))));
}

// Maximum over the 64 lanes the same way.
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true)));
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)),
              r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fmaxf(fmaxf(r0, r1), fmaxf(r2, r3)))));
}
// Index of the lane within its wave, derived on the spot (the compiler may not keep the result of a volatile statement
// for a later use, which is the point: see render_wave).
__device__ __forceinline__ int fresh_lane_id() {
  int t;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(t));
  return t;
}

// Is this lane in the wave-uniform set `mask`?  (The mask becomes the lane predicate directly; a shift by the lane number
// would keep a 64-bit per-lane value alive wherever it is used.)
__device__ __forceinline__ bool in_mask(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

// The bundle of the rays in `cls` (rt_math.h: RtBundle, rt_cull_test) for the walk's triangle skipping.  The point the lines pass close
// to is taken from the first ray: its origin, or for a shadow ray the point at the light's distance (the rays towards a
// point light all pass through the light; primary rays share the eye; parallel rays only need some point).  Returns
// false when the bundle is too wide to rule anything out (rays that have little in common): the walk then tests every
// triangle as it comes.  All 64 lanes must be active.
__device__ __forceinline__ bool wave_bundle(uint64_t cls, const RtVec3& e, const RtVec3& d, float stop_t, RtBundle& out, float r0_max, float ksp_max) {
  const bool in = in_mask(cls);
  const float h = stop_t >= 0.f ? fminf(stop_t, RT_INF) : 0.f;
  const int first = __builtin_ctzll(cls), last = 63 - __builtin_clzll(cls);
  auto from = [&](float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); };
  const RtVec3 o = {from(fmaf(h, d.x, e.x), first), from(fmaf(h, d.y, e.y), first), from(fmaf(h, d.z, e.z), first)};
  // axis: between the first and the last ray of the class (opposite corners of a tile); any unit vector would do
  const RtVec3 sum = {from(d.x, first) + from(d.x, last), from(d.y, first) + from(d.y, last), from(d.z, first) + from(d.z, last)};
  const float len2 = rt_dot(sum, sum);
  if (!(len2 > 1.f)) return false;   // (more than 120 degrees apart: certainly not a bundle)
  const float rl = __builtin_amdgcn_rsqf(len2);   // (the axis is a unit vector to within 2^-21: covered by the slack of rt_bundle_finish)
  const RtVec3 a = {sum.x * rl, sum.y * rl, sum.z * rl};
  // rt_bundle_lane on squared lengths (the root is monotone: the largest square belongs to the largest length)
  const RtVec3 w = {o.x - e.x, o.y - e.y, o.z - e.z};
  const RtVec3 c = {w.y * d.z - w.z * d.y, w.z * d.x - w.x * d.z, w.x * d.y - w.y * d.x};
  const RtVec3 q = {d.x - a.x, d.y - a.y, d.z - a.z};
  const float r0 = __builtin_amdgcn_sqrtf(wave_max(in ? rt_dot(c, c) : 0.f)) * 1.000001f;
  const float bw = __builtin_amdgcn_sqrtf(wave_max(in ? rt_dot(w, w) : 0.f)) * 1.000001f;
  const float k = __builtin_amdgcn_sqrtf(wave_max(in ? rt_dot(q, q) : 0.f)) * 1.000001f;
  if (!(r0 <= r0_max && k <= ksp_max && bw < 1e6f)) return false;   // (also false for NaN)
  out = rt_bundle_finish(o, a, r0, bw, k);
  return true;
}

// The walk of a set of rays (`live`) that agree on their direction signs (`variant`: the copy of the node array whose plane
// pairs are in (near, far) order for them; zpat 0) or share the pattern `zpat` of zero direction components (variant 0),
// as one block of assembly - see the call in trace_wave for what the block does and which registers it owns.
// `nodes` is that variant's array; the root's children start at record top_first, `rem` names the ones to test.
// Returns the lanes that are still looking for a hit afterwards (shadow rays leave when they are known to be blocked).
// bary_lds != nullptr: the barycentrics of an accepted triangle go to LDS (word 0 and word RT_WBLOCK of the lane's column
// there) instead of into `hit` - they change a few times per walk and are read once per closest hit, and as part of `hit`
// they would be carried through the whole walk in two registers.
// CULL, cull != 0: `cull` is the byte offset of the RtCullRec array behind the triangle records and `bundle` the bundle of the
// rays in `live` (wave_bundle): on entering a leaf the walk leaves out the triangles that no ray of the bundle can be
// accepted by.  cull == 0, or the variant of the block without that code (!CULL: kernels whose rays never form a
// bundle): every triangle of every leaf entered is tested.
#ifdef RT_WAVE_TIMES_CLK
__device__ float* rt_dbg_ts();   // the `t` output of the launch (defined behind the kernel's argument block)
#endif
template <bool CULL, bool BARY>
__device__ __forceinline__ uint64_t walk_asm(const float4* nodes, uint32_t variant, uint32_t zpat, const float4* tri_leaf, uint32_t top_first, uint32_t rem,
                                             uint64_t live, const RtVec3& e, const RtVec3& d, const RtVec3& inv, const RtVec3& fc, float neg_eb,
                                             float stop_t, Hit& hit, WaveStack& st, float* bary_lds, uint32_t cull, const RtBundle& bundle) {
  const RtVec3 c2 = {-d.x, -d.y, -d.z};
  // (wave-uniform values, pinned to scalar registers once: the walk reads them on every leaf it enters)
  const uint32_t b0 = uni(__float_as_uint(bundle.ox)), b1 = uni(__float_as_uint(bundle.oy)), b2 = uni(__float_as_uint(bundle.oz)), b3 = uni(__float_as_uint(bundle.ax)),
                 b4 = uni(__float_as_uint(bundle.ay)), b5 = uni(__float_as_uint(bundle.az)), b6 = uni(__float_as_uint(bundle.r0)), b7 = uni(__float_as_uint(bundle.ksp)),
                 b8 = uni(__float_as_uint(bundle.k1)), b9 = uni(__float_as_uint(bundle.bw));                   // third column of the triangle solve (triangle.cu:118-120)
  const v2f e_xy = {e.x, e.y}, c_zx = {c2.z, c2.x}, c_yz = {c2.y, c2.z}, c_yx = {c2.y, c2.x};
  uint64_t a_blk = (uint64_t)(uintptr_t)(nodes + 2 * (size_t)top_first);
  uint32_t a_rem = rem | 0x100u, a_first = top_first, a_idx = 0u, a_rfirst = 0u, a_rmeta = 0u, a_res = 0u, a_sp = 0u;
  uint64_t a_tmask = live, a_mine = live, a_live = live, a_acc = 0ull, a_und = 0ull;
  const uint64_t a_tri = (uint64_t)(uintptr_t)tri_leaf;
  // (the loop with four record sets takes s[68:83] over while its leaf is open and puts the node array's address back
  //  afterwards: an operand that is written, like the rest of that range)
  uint64_t a_nodes = (uint64_t)(uintptr_t)nodes;
  // (the record of the child to be tested next: requested inside the block, carried across the exact node test outside it)
  uint64_t a_nx0 = 0ull, a_nx1 = 0ull, a_nx2 = 0ull, a_nx3 = 0ull;
  uint32_t a_work = 0u;   // triangles met in large leaves so far: the block treats a walk with many of them differently (see the register map at the call)
  const float cmax = rt_max3(fabsf(fc.x), fabsf(fc.y), fabsf(fc.z));
#ifdef RT_WAVE_TIMES_CLK
  uint32_t dbg_n1 = 0u, dbg_n2 = 0u; unsigned long long dbg_out = 0ull;
#endif
#ifdef RT_STAGE_COUNT
  // measurement builds (with -DRT_WAVE_TIMES_CLK and a block generated with RT_GEN_COUNT=1): s17 counts the triangle iterations
  // that went on to the second (low half) and third stage (high half), s14 the triangle (low half) and node iterations (high half)
  // s15 the blocks of children entered (low half) and the leaves (high half), s16 the children that came to the margin test (low half) and the pops
  uint32_t a_stage = 0u, a_iter = 0u, a_enter = 0u, a_tail = 0u;
#define RT_COUNT_OPERAND , "+{s17}"(a_stage), "+{s14}"(a_iter), "+{s15}"(a_enter), "+{s16}"(a_tail)
#else
#define RT_COUNT_OPERAND
#endif
  // (one copy of the block per kind of node test inside the one statement: sign-uniform rays (0), or the rays' common
  //  zero pattern (1..6), chosen by s18)
  // Operand lists of the four forms of the block (with / without the bundle test; barycentrics to registers / to LDS).
#define RT_WALK_OUTPUTS \
            "+{s[68:69]}"(a_blk), "+{s70}"(a_rem), "+{s71}"(a_first), "+{s[72:73]}"(a_tmask), "+{s[74:75]}"(a_mine), "+{s[76:77]}"(a_live), \
            "+{s78}"(a_idx), "+{s79}"(a_rfirst), "+{s80}"(a_rmeta), "+{s[34:35]}"(a_acc), "+{s[96:97]}"(a_und), \
            "+{s30}"(a_res), "+{m0}"(a_sp) RT_COUNT_OPERAND, \
            "+{v60}"(st.first), "+{v61}"(st.cur), "+{v62}"(st.mask_lo), "+{v63}"(st.mask_hi), \
            "+{s[82:83]}"(a_nodes), "+{s23}"(a_work), "+{v58}"(hit.t), [prim] "+v"(hit.prim), \
            "+{s[84:85]}"(a_nx0), "+{s[86:87]}"(a_nx1), "+{s[88:89]}"(a_nx2), "+{s[90:91]}"(a_nx3)
#define RT_WALK_INPUTS \
            "{s[32:33]}"(a_tri), "{s18}"(zpat), "{v[40:41]}"(e_xy), "{v42}"(e.z), "{v[44:45]}"(c_zx), "{v[46:47]}"(c_yz), \
            "{v[48:49]}"(c_yx), "{v50}"(inv.x), "{v51}"(inv.y), "{v52}"(inv.z), "{v53}"(fc.x), "{v54}"(fc.y), "{v55}"(fc.z), "{v56}"(cmax), \
            "{v57}"(neg_eb), "{v59}"(stop_t)
#define RT_WALK_BUNDLE \
            , "{s19}"(cull), [b0] "s"(b0), [b1] "s"(b1), [b2] "s"(b2), [b3] "s"(b3), [b4] "s"(b4), [b5] "s"(b5), [b6] "s"(b6), [b7] "s"(b7), [b8] "s"(b8), [b9] "s"(b9)
#define RT_WALK_CLOBBERS \
            "s20", "s21", "s22", "s24", "s25", "s26", "s27", "s28", "s29", "s31", "s92", "s93", "s94", "s95", "s98", "s99", "s81", \
            "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", \
            "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", \
            "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", \
            "v75", "v76", "v77", "v78", "v79", "vcc", "scc", "memory"
  // (barycentrics in LDS: the wave's two columns, lane by lane; the block adds the lane's offset itself)
  const uint32_t bary_addr = BARY ? uni((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)bary_lds) : 0u;
  for (;;) {
    if constexpr (CULL && BARY) {
      asm volatile(RT_TRACE_ASM_CULL_LDS : RT_WALK_OUTPUTS : RT_WALK_INPUTS RT_WALK_BUNDLE, [bary] "s"(bary_addr), [boff] "n"(RT_WBLOCK * 4) : RT_WALK_CLOBBERS);
    } else if constexpr (CULL) {
      asm volatile(RT_TRACE_ASM_CULL_REG : RT_WALK_OUTPUTS, [bet] "+v"(hit.bet), [gam] "+v"(hit.gam) : RT_WALK_INPUTS RT_WALK_BUNDLE : RT_WALK_CLOBBERS);
    } else if constexpr (BARY) {
      asm volatile(RT_TRACE_ASM_PLAIN_LDS : RT_WALK_OUTPUTS : RT_WALK_INPUTS, [bary] "s"(bary_addr), [boff] "n"(RT_WBLOCK * 4) : RT_WALK_CLOBBERS);
    } else {
      asm volatile(RT_TRACE_ASM_PLAIN_REG : RT_WALK_OUTPUTS, [bet] "+v"(hit.bet), [gam] "+v"(hit.gam) : RT_WALK_INPUTS : RT_WALK_CLOBBERS);
    }
    if (a_res == 0u) break;
#ifdef RT_WAVE_TIMES_CLK
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime();
    dbg_n1++;
#endif
    if (a_res == 1u) {
      // the exact test for the lanes in a_und: the planes of the child just tested are read again
      const uint32_t c_cur = (a_rmeta >> 24) & 7u;
      typedef float f4 __attribute__((ext_vector_type(4)));
      typedef const __attribute__((address_space(4))) f4* cnode_ptr;
      auto rec = [](cnode_ptr q) { const f4 v = *q; return float4{v.x, v.y, v.z, v.w}; };
      constexpr uint32_t NS = 2;
      const cnode_ptr cb = (cnode_ptr)a_blk;
      const float4 na = rec(cb + NS * c_cur), nb = rec(cb + NS * c_cur + 1);
      RtVec3 dd = d, ee = e; float T = hit.t;
      asm volatile("" : "+v"(dd.x), "+v"(dd.y), "+v"(dd.z), "+v"(ee.x), "+v"(ee.y), "+v"(ee.z), "+v"(T));
      const RtRayConst rc = rt_ray_const(dd);
      const bool wx = (variant & 1u) != 0u, wy = (variant & 2u) != 0u, wz = (variant & 4u) != 0u;
      const RtVec3 ini = {wx ? na.y : na.x, wy ? na.w : na.z, wz ? nb.x : nb.y}, end = {wx ? na.x : na.y, wy ? na.z : na.w, wz ? nb.y : nb.x};
      const float S = rt_node_score(ini, end, ee, dd, rc, rt_t_float(rc, T));
      uint64_t pos_m = __builtin_amdgcn_ballot_w64(S > 0.f);
      const uint64_t tie_m = __builtin_amdgcn_ballot_w64(S == 0.f) & a_und;
      if (tie_m) {   // a boundary case: settle it with the compare form
        const bool ex = rt_node_accept_exact(ini, end, ee, dd, rc, rt_t_const(rc, T));
        pos_m = (pos_m & ~tie_m) | (__builtin_amdgcn_ballot_w64(ex) & tie_m);
      }
      a_acc |= pos_m & a_und;
    }
#ifdef RT_WAVE_TIMES_CLK
    dbg_out += __builtin_amdgcn_s_memtime() - dbg_t0;
#endif
  }
#ifdef RT_WAVE_TIMES_CLK
  // diagnostic build: per workgroup, behind the waves' records in the `t` output: exits for the exact node test, exits for
  // triangle candidates, sixteenths of the cycles spent outside the block between exit and re-entry, walks
  if (float* const ts = rt_dbg_ts()) {
    if (fresh_lane_id() == 0) {
      uint32_t* const dbg = (uint32_t*)ts + (1u << 22) + 4u * blockIdx.x;
      atomicAdd(dbg + 0, dbg_n1); atomicAdd(dbg + 1, dbg_n2); atomicAdd(dbg + 2, (uint32_t)(dbg_out >> 4)); atomicAdd(dbg + 3, 1u);
#ifdef RT_STAGE_COUNT
      uint32_t* const stg = (uint32_t*)ts + (1u << 22) + (1u << 20) + 8u * blockIdx.x;
      atomicAdd(stg + 0, a_iter & 0xffffu); atomicAdd(stg + 1, a_stage & 0xffffu); atomicAdd(stg + 2, a_stage >> 16); atomicAdd(stg + 3, a_iter >> 16);
      atomicAdd(stg + 4, a_enter & 0xffffu); atomicAdd(stg + 5, a_enter >> 16); atomicAdd(stg + 6, a_tail & 0xffffu); atomicAdd(stg + 7, a_tail >> 16);
#if RT_STAGE_COUNT == 2
      // (block generated with RT_GEN_COUNT=leaf: the four registers count something else; totals over the launches of the frame)
      uint32_t* const lf = (uint32_t*)ts + (1u << 22) + (1u << 21) + (1u << 20);
      atomicAdd(lf + 0, a_iter); atomicAdd(lf + 1, a_enter); atomicAdd(lf + 2, a_tail); atomicAdd(lf + 3, a_stage); atomicAdd(lf + 4, 1u);
#endif
#endif
    }
  }
#endif
  return a_live;
}

// Closest hit (ray.cu:29-54) for all lanes of the wave at once.
//   want   : this lane has a ray to trace
//   stop_t : >= 0 for a shadow ray: the lane drops out as soon as its t falls below stop_t, because
//            only "t < light distance" is consumed (ray.cu:70) and t never grows; < 0 = never
template <bool COUNT, bool CULLABLE>
__device__ __forceinline__ void trace_wave(const float4* __restrict__ nodes, int n_wnodes, const float4* __restrict__ tri_leaf,
                                           const RtSphere* __restrict__ spheres, int n_spheres, bool has_oct, bool want,
                                           const RtVec3& e, const RtVec3& d, float stop_t, Hit& hit, WaveStack& st,
                                           int lane, Tally& tl, float* cull_lds, uint32_t cull_offset, float cull_r0_max, float cull_ksp_max) {
  // Lane sets are kept as wave-uniform 64-bit masks in SGPRs (who still has an undecided ray, who entered the
  // current branch, who accepted the current node, ...); they are combined on the scalar unit and only turned
  // back into a per-lane predicate on the rare paths that need one.
  constexpr bool PATHS = COUNT || RT_PATHS_ON;
  uint64_t live = __builtin_amdgcn_ballot_w64(want);
  if (!live) return;
  if (PATHS) tl.w[0]++;
  const RtVec3 inv = {1.f / d.x, 1.f / d.y, 1.f / d.z};   // only feeds the conservative classification
  const RtVec3 c2 = {-d.x, -d.y, -d.z};                   // third column of the triangle solve (triangle.cu:118-120)
  const v2f e_xy = {e.x, e.y}, c_zx = {c2.z, c2.x}, c_yz = {c2.y, c2.z}, c_yx = {c2.y, c2.x};
  const unsigned zmask = (d.x == 0.f ? 1u : 0u) | (d.y == 0.f ? 2u : 0u) | (d.z == 0.f ? 4u : 0u);
  const bool any_zero = (__builtin_amdgcn_ballot_w64(zmask != 0u) & live) != 0ull;
  // Zero components usually come from an axis-parallel directional light, the same for every ray of the wave: then the
  // pattern is wave-uniform and the node test specialised on it runs (0 = not uniform, or no zero at all).
  uint32_t zpat = 0u;
  if (any_zero) {
    const uint64_t zx = __builtin_amdgcn_ballot_w64(d.x == 0.f) & live, zy = __builtin_amdgcn_ballot_w64(d.y == 0.f) & live,
                   zz = __builtin_amdgcn_ballot_w64(d.z == 0.f) & live;
    if ((zx == 0ull || zx == live) && (zy == 0ull || zy == live) && (zz == 0ull || zz == live))
      zpat = (zx ? 1u : 0u) | (zy ? 2u : 0u) | (zz ? 4u : 0u);
    if (zpat == 7u) zpat = 0u;   // cannot happen for a direction; the general form copes with anything
  }
  zpat = uni(zpat);
  // Direction signs: when every ray of the wave agrees on all three (and none is zero), the wave reads the node
  // variant whose plane pairs are already in (near, far) order for those signs.  Otherwise variant 0 = (lo, hi).
  const uint64_t sx = __builtin_amdgcn_ballot_w64(d.x < 0.f) & live, sy = __builtin_amdgcn_ballot_w64(d.y < 0.f) & live,
                 sz = __builtin_amdgcn_ballot_w64(d.z < 0.f) & live;
  // (a component so small that its inverse overflows takes the general path: the scan loop's fused plane distances
  //  need finite c = e * inv)
  const float INF = __builtin_inff();
  const bool inv_finite = (__builtin_amdgcn_ballot_w64(!(fabsf(inv.x) < INF && fabsf(inv.y) < INF && fabsf(inv.z) < INF)) & live) == 0ull;
  const bool presorted = !any_zero && inv_finite && (sx == 0ull || sx == live) && (sy == 0ull || sy == live) && (sz == 0ull || sz == live);
  const uint32_t variant = uni(presorted ? ((sx ? 1u : 0u) | (sy ? 2u : 0u) | (sz ? 4u : 0u)) : 0u);   // uni(): keeps the node addresses scalar
  nodes += 2 * (size_t)variant * (size_t)n_wnodes;   // each variant is a dense copy of the whole array

  if (has_oct) {
    // The walk goes over the children of one branch at a time: `rem` holds the children still to be tested (bit c =
    // child c, lowest first = the reference's order 0..7).  Empty leaves are left out of `rem` from the start: testing
    // one has no effect on any result, only on the node-test count, so the instrumented variant keeps them.
    constexpr uint32_t NS = 2;   // float4 per node record
    // The node array is read through the constant address space: such loads are scalar loads whatever the compiler can
    // or cannot prove about stores elsewhere, which allows keeping a running pointer to the current branch's children.
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) f4* cnode_ptr;
    const cnode_ptr cnodes = (cnode_ptr)(uintptr_t)nodes;
    auto rec = [](cnode_ptr p) { const f4 v = *p; return float4{v.x, v.y, v.z, v.w}; };
    uint32_t top_first, rem;   // the root is entered without a test (ray.cu:36-38)
    { const float4 root_b = rec(cnodes + 1); top_first = uni(__float_as_uint(root_b.z)); rem = COUNT ? 0xffu : (uni(__float_as_uint(root_b.w)) & 0xffu); }
    uint64_t top_mask = live;
    int sp = 0;
    // Node records reach the SGPRs through the scalar cache.  The record of the child that will be tested next is
    // always requested as soon as it is known (next sibling: before the current child is classified; first child of
    // a branch: on descent or ascent), so its latency hides behind the classification.
    // (one 8-word value: the assembly loops keep it in a fixed block of eight scalar registers, and the loads around
    // them go straight into that block)
    typedef float f8 __attribute__((ext_vector_type(8)));
    typedef const __attribute__((address_space(4))) f8* cnode8_ptr;
    auto rec8 = [](cnode_ptr p) { return *(cnode8_ptr)p; };
    f8 nx = {0, 0, 0, 0, 0, 0, 0, 0};
    // margin that no node of this tree exceeds for this ray (the root's planes bound every node's)
    // The assembly scan loop forms a plane distance by one fused multiply-add, p * inv - c with c = e * inv (rt_math.h,
    // rt_slab_fused); the compiled loops of the instrumented variants keep the two-step form (same decisions either way:
    // both classifications only ever abstain).
    float neg_eb = 0.f;
    RtVec3 fc = {0.f, 0.f, 0.f};
    constexpr bool FUSED = !COUNT && !PATHS;
    if (presorted) {
      const float4 ra = rec(cnodes), rb = rec(cnodes + 1);
      if (FUSED) {
        const RtSlabFused fu = rt_slab_fused(e, inv);
        fc = fu.c;
        neg_eb = -rt_slab_margin_bound_fused(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, e, inv, fu, hit.t);
      } else neg_eb = -rt_slab_margin_bound(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, e, inv, hit.t);
    }
#ifdef RT_WAVE_CULL
    // Measurement build (off in the product; tools/build_variant.sh cull -DRT_WAVE_CULL): the children of a branch are
    // put in front of the wave's rays as a whole before they are scanned.  On the 4K dragon frame it takes 54 % of the
    // node iterations away (18.0 M -> 8.3 M) and about 1 % of the time: the iterations it removes are the cheap ones,
    // whose record was requested an iteration ahead, and each descent now waits for a vector load.
    // Bounds of the wave's rays (see the descent): per axis the origin that makes a plane distance smallest / largest
    // and the inverse direction of smallest / largest magnitude.
    bool cull_on = presorted && !COUNT && (!PATHS || RT_PATHS_ON);
    if (cull_on) {
      float cu_eA[3] = {0, 0, 0}, cu_eB[3] = {0, 0, 0}, cu_iS[3] = {0, 0, 0}, cu_iL[3] = {0, 0, 0}, cu_neb = 0.f;
      const float PINF = __builtin_inff();
      const float ev[3] = {e.x, e.y, e.z}, iv[3] = {inv.x, inv.y, inv.z};
      bool finite = true;
      for (int k = 0; k < 3; k++) {
        const float emin = wave_min(want ? ev[k] : PINF), emax = -wave_min(want ? -ev[k] : PINF);
        const float amin = wave_min(want ? fabsf(iv[k]) : PINF), amax = -wave_min(want ? -fabsf(iv[k]) : PINF);
        const bool neg = ((variant >> k) & 1u) != 0u;
        cu_iS[k] = neg ? -amin : amin; cu_iL[k] = neg ? -amax : amax;
        cu_eA[k] = neg ? emin : emax; cu_eB[k] = neg ? emax : emin;
        finite = finite && amax < PINF;
      }
      cu_neb = wave_min(want ? neg_eb : PINF);
      cull_on = finite && cu_neb > -PINF;
      // they wait in LDS for the descents (thirteen scalar registers across the walk would be spilled)
      for (int k = 0; k < 3; k++) { cull_lds[k] = cu_eA[k]; cull_lds[4 + k] = cu_eB[k]; cull_lds[8 + k] = cu_iS[k]; cull_lds[12 + k] = cu_iL[k]; }
      cull_lds[3] = cu_neb;
    }
#endif
#ifndef RT_NO_ASM_WALK
    // ---- The walk of a wave whose rays agree on their direction signs (nearly all primary and point-light shadow waves) or
    // share a pattern of zero direction components (shadow rays of axis-parallel lights), as one block of assembly (rt_trace_asm.inc, written by tools/gen_trace_asm.py): scan of a branch's children with
    // fused plane distances, margin and acceptance, descent (the stack entry is one v_writelane per word, the stack
    // pointer lives in M0), ascent, and the leaf loop with the three numerator stages of the triangle test on two sets
    // of record registers filled in turn.  What the compiler makes of the same walk spends most of its scalar
    // instructions moving records and lane masks between register sets; the scalar unit, which a compute unit's four
    // SIMDs share, is what this kernel keeps busiest.  The block leaves for one thing only and is entered again
    // afterwards: the exact node test of lanes the classification could not decide (reason 1: about 0.5 % of the node
    // tests).  Reason 0: the walk is over.  The candidates of a triangle (2.5 % of the triangle tests) are decided by a
    // routine of the block itself: the reference's three quotients as correctly rounded divisions, its seven compares,
    // and the lanes that pass take t, the triangle's id and the barycentrics (operands, or the wave's columns in LDS).
    // Registers (fixed, named in the generator): s[36:67] two triangle records, s[68:69] address of the current
    // branch's children, s70 children left (bit 8 always set: "child 8" is the record behind the block, requested but
    // never tested), s71 index of child 0, s[72:73] lanes that entered the branch, s[74:75] those of them still
    // undecided, s[76:77] undecided lanes of the wave, s78 child whose record is in s[84:91], s79/s80 links of the child
    // just tested, s81 byte offset, s[82:83] node array, s[32:33] triangle array, s[34:35] accepted lanes, s[96:97] undecided lanes,
    // s[28:29] triangle candidates, s20 / s21 byte offsets of the triangle record under test and the next one to be tested, s22 offset
    // behind the leaf's last record (in a leaf whose triangles were put in front of the wave's bundle first: scratch), s[24:25] lanes
    // whose hit changed in the leaf, s30 reason / re-entry, s[30:31] the address the decision routine comes back to,
    // s19 byte offset of the RtCullRec array (0: no bundle), s[98:99] within such a leaf the triangles still to be tested,
    // ten scalar operands for the bundle; v[40:59] the ray
    // (origin, negated direction in the pairs the first stage multiplies, inverse direction, e * inv, margin terms,
    // closest hit so far - which the decision routine moves -, shadow-ray length), v[60:63] the wave's stack, v32 v33 v35 v36
    // determinant and numerators of a candidate, v34 v37:v39 v64:v79 scratch.
    // A walk with a long life ahead - one that has met 64 triangles in leaves with six or more to test; s23 counts them -
    // takes such leaves through a loop with FOUR record sets: s[68:83] and s[84:99] hold records too, the traversal
    // state that lives there waits in lanes 32 .. 44 of v61, s20 .. s23 are the leaf-relative indices of the records in
    // the four sets, s[24:25] the triangles still to request, vcc_lo the leaf's first record.  Two records are tested
    // while the next two are on their way: a scalar load takes such a wave about 400 cycles, a test 150, loads return
    // out of order and the only wait is for all of them.
    if ((presorted || zpat != 0u) && FUSED && (rem & 0xffu) != 0u) {
      RtBundle bundle = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const uint32_t cull = (CULLABLE && cull_offset != 0u && wave_bundle(live, e, d, stop_t, bundle, cull_r0_max, cull_ksp_max)) ? cull_offset : 0u;
      live = walk_asm<CULLABLE, false>(nodes, variant, zpat, tri_leaf, top_first, rem, live, e, d, inv, fc, neg_eb, stop_t, hit, st, nullptr, cull, bundle);
      if (!live) return;
    } else {
#endif
    cnode_ptr blk = cnodes + NS * top_first;   // records of the current branch's children
    { const uint32_t c = (uint32_t)__builtin_ctz(rem | 0x100u); nx = rec8(blk + NS * c); }
    // `mine_m` = lanes that entered the current branch and are still undecided.  It only changes when the branch changes or
    // a lane gets decided; whenever it becomes empty the branch is abandoned by clearing `rem`, so that the loop head has
    // a single scalar test.
    uint64_t mine_m = top_mask;
    for (;;) {
      if (rem == 0u) {   // branch exhausted (or abandoned)
        if (sp == 0) break;
        sp--;
        top_first = __builtin_amdgcn_readlane(st.first, sp); rem = __builtin_amdgcn_readlane(st.cur, sp);
        top_mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(st.mask_hi, sp) << 32) | (uint32_t)__builtin_amdgcn_readlane(st.mask_lo, sp);
        mine_m = top_mask & live;
        if (mine_m == 0ull) { rem = 0u; continue; }
        blk = cnodes + NS * top_first;
        const uint32_t c = (uint32_t)__builtin_ctz(rem);   // entries are only pushed with children left
        nx = rec8(blk + NS * c);
        continue;
      }
      float4 na, nb;
      uint64_t acc_m, und_m;
      // conservative slab classification first; the exact test only where some lane is within rounding of a boundary
      if (presorted || zpat != 0u) {
        // Siblings are scanned in a tight loop until one is accepted by some lane (or needs the exact test): about two
        // node tests in three end in "no lane", and those never leave this loop.
        // A record is consumed by six subtractions (plane - e) and two scalar copies (links); the request for the next
        // sibling is tied to those results, so it is issued afterwards and its data may land in the very registers the
        // record occupied: nothing is copied from "next" to "current".  Whoever needs the planes again (the exact test,
        // rarely) reads the record once more.
        uint32_t rem_cur = 1u, r_first = 0u, r_meta = 0u;   // rem_cur: `rem` before the child just tested was taken out of it
        // (RT_SCAN_LOOP stamps the loop out once per kind of node test: CLASSIFY turns the six differences into acc_m / und_m)
#define RT_SCAN_LOOP(SIDE, ...)                                                                                                         \
        for (;;) {                                                                                                                      \
          rem_cur = rem;                                                                                                                \
          rem &= rem - 1u;                                                                                                              \
          const float dx0 = nx[0] - e.x, dx1 = nx[1] - e.x, dy0 = nx[2] - e.y, dy1 = nx[3] - e.y, dz0 = nx[4] - e.z, dz1 = nx[5] - e.z;             \
          SIDE;                                                                                                                         \
          r_first = uni(__float_as_uint(nx[6])); r_meta = uni(__float_as_uint(nx[7]));                                                  \
          {                                                                                                                             \
            /* next sibling ("child 8" after the last one is the first record of the next block or the pad record) */                  \
            uint32_t off = (uint32_t)__builtin_ctz(rem | 0x100u) * 32u;   /* byte offset of that record within the block */             \
            asm volatile("" : "+s"(off) : "v"(dx0), "v"(dx1), "v"(dy0), "v"(dy1), "v"(dz0), "v"(dz1), "v"(side), "s"(r_first), "s"(r_meta)); \
            const cnode_ptr rp = (cnode_ptr)((const __attribute__((address_space(4))) char*)blk + off);                                 \
            nx = rec8(rp);                                                                                         \
          }                                                                                                                             \
          if (PATHS && (r_meta & 0x800000ffu) != 0u) tl.w[1]++;                                                                                       \
          if (COUNT && ((mine_m >> lane) & 1ull)) tl.node++;   /* w[1]: iterations the plain variant also makes */                      \
          acc_m = 0ull; und_m = 0ull;                                                                                                   \
          __VA_ARGS__                                                                                                                   \
          if ((acc_m | und_m) != 0ull || rem == 0u) break;                                                                              \
        }
        // a direction component that is zero takes the node test written for that pattern (rt_node_slab_zero_d); the cube's
        // side is taken from the record while it is there
#define RT_CLASSIFY_ZERO(ZX, ZY, ZZ)                                                                                                    \
          {                                                                                                                             \
            const float dlo[3] = {dx0, dy0, dz0}, dhi[3] = {dx1, dy1, dz1};                                                             \
            /* most children lie beside every ray of the wave: settled without any slab arithmetic */                                  \
            if ((mine_m & ~__builtin_amdgcn_ballot_w64(rt_node_zero_outside<ZX, ZY, ZZ>(dlo, dhi, side))) != 0ull) {                   \
              bool c_acc, c_rej;                                                                                                        \
              rt_node_slab_zero_d<ZX, ZY, ZZ>(dlo, dhi, side, inv, hit.t, c_acc, c_rej);                                                \
              acc_m = __builtin_amdgcn_ballot_w64(c_acc) & mine_m;                                                                      \
              und_m = mine_m & ~(acc_m | __builtin_amdgcn_ballot_w64(c_rej));                                                           \
            }                                                                                                                           \
          }
#ifndef RT_NO_ASM_SCAN
        // (measurement builds: instructions without effect in the scan loop, to see what the loop's time follows)
#if defined(RT_JUNK_VALU)
#define RT_JUNK_ASM "v_mul_f32 %[x], %[ix], %[iy]\n v_mul_f32 %[r], %[ix], %[iy]\n v_mul_f32 %[tc], %[ix], %[iy]\n v_mul_f32 %[x], %[iz], %[iy]\n v_mul_f32 %[r], %[iz], %[iy]\n v_mul_f32 %[tc], %[ix], %[iz]\n"
#elif defined(RT_JUNK_SALU)
#define RT_JUNK_ASM "s_mov_b32 %[tmp], %[rem]\n s_mov_b32 %[tmp], %[prev]\n s_mov_b32 %[tmp], %[rem]\n s_mov_b32 %[tmp], %[prev]\n s_mov_b32 %[tmp], %[rem]\n s_mov_b32 %[tmp], %[prev]\n"
#else
#define RT_JUNK_ASM
#endif
        if (presorted && !COUNT && !PATHS) {
          // The same loop, its common part written out: children that every ray of the wave misses by more than the ray's
          // largest margin never leave the assembly (consume the record, request the next sibling into the same eight
          // registers, six products, head of the classification, one compare against the margin bound).  It ends at the
          // first child some lane may hit, or with the last child, and hands the plane distances and the head over.
          for (;;) {
            uint32_t prev, tmp;
            uint64_t notrej;
            float t0, t1, t2, t3, t4, t5, tn, tf, span, tc, x, r;
            const uint64_t blk_bits = (uint64_t)blk;
            asm volatile(
                "L_scan_%=:\n"
                "s_waitcnt lgkmcnt(0)\n"
                "v_fma_f32 %[t0], s84, %[ix], -%[cx]\n"
                "v_fma_f32 %[t1], s85, %[ix], -%[cx]\n"
                "v_fma_f32 %[t2], s86, %[iy], -%[cy]\n"
                "v_fma_f32 %[t3], s87, %[iy], -%[cy]\n"
                "v_fma_f32 %[t4], s88, %[iz], -%[cz]\n"
                "v_fma_f32 %[t5], s89, %[iz], -%[cz]\n"
                "s_mov_b32 %[first], s90\n"
                "s_mov_b32 %[meta], s91\n"
                "s_mov_b32 %[prev], %[rem]\n"
                "s_add_i32 %[tmp], %[rem], -1\n"
                "s_and_b32 %[rem], %[rem], %[tmp]\n"
                "s_or_b32 %[tmp], %[rem], 0x100\n"
                "s_ff1_i32_b32 %[tmp], %[tmp]\n"
                "s_lshl_b32 %[tmp], %[tmp], 5\n"
                "s_load_dwordx8 s[84:91], %[blk], %[tmp] offset:0x0\n"
                RT_JUNK_ASM
                "v_max3_f32 %[tn], %[t0], %[t2], %[t4]\n"
                "v_min3_f32 %[tf], %[t1], %[t3], %[t5]\n"
                "v_cmp_lt_f32 vcc, 0, %[tn]\n"
                "v_sub_f32 %[span], %[tf], %[tn]\n"
                "v_cndmask_b32 %[tc], %[tf], %[tn], vcc\n"
                "v_sub_f32 %[x], %[T], %[tc]\n"
                "v_min3_f32 %[r], %[span], %[tf], %[x]\n"
                "v_cmp_lt_f32_e64 %[nr], %[r], %[neb]\n"
                "s_andn2_b64 %[nr], %[mine], %[nr]\n"
                "s_cbranch_scc1 L_done_%=\n"
                "s_cmp_lg_u32 %[rem], 0\n"
                "s_cbranch_scc1 L_scan_%=\n"
                "L_done_%=:\n"
                "s_waitcnt lgkmcnt(0)\n"
                : [nx] "+{s[84:91]}"(nx), [rem] "+s"(rem), [first] "=&s"(r_first), [meta] "=&s"(r_meta), [prev] "=&s"(prev), [tmp] "=&s"(tmp),
                  [nr] "=&s"(notrej), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5),
                  [tn] "=&v"(tn), [tf] "=&v"(tf), [span] "=&v"(span), [tc] "=&v"(tc), [x] "=&v"(x), [r] "=&v"(r)
                : [blk] "s"(blk_bits), [mine] "s"(mine_m), [cx] "v"(fc.x), [cy] "v"(fc.y), [cz] "v"(fc.z), [ix] "v"(inv.x), [iy] "v"(inv.y),
                  [iz] "v"(inv.z), [T] "v"(hit.t), [neb] "v"(neg_eb)
                : "vcc", "scc");
            rem_cur = uni(prev);
            acc_m = 0ull; und_m = 0ull;
            if (notrej == 0ull) break;   // the last child, missed by everybody
            RtSlabHead hd;
            hd.tN = tn; hd.tF = tf; hd.span = span; hd.front = tn > 0.f; hd.tc = tc; hd.x = x; hd.r = r;
            bool c_acc, c_rej;
            rt_node_slab_tail_fused(hd, t0, t1, t2, t3, t4, t5, hit.t, rt_max3(fabsf(fc.x), fabsf(fc.y), fabsf(fc.z)), c_acc, c_rej);
            acc_m = __builtin_amdgcn_ballot_w64(c_acc) & mine_m;
            und_m = mine_m & ~(acc_m | __builtin_amdgcn_ballot_w64(c_rej));
            if ((acc_m | und_m) != 0ull || rem == 0u) break;
          }
        } else
#endif
        if (presorted) {
          RT_SCAN_LOOP(const float side = dx0, {
            const float tx0 = dx0 * inv.x, tx1 = dx1 * inv.x, ty0 = dy0 * inv.y, ty1 = dy1 * inv.y, tz0 = dz0 * inv.z, tz1 = dz1 * inv.z;
            const RtSlabHead hd = rt_node_slab_head(tx0, tx1, ty0, ty1, tz0, tz1, hit.t);
            /* Most children are missed by every ray of the wave, and by far: those are settled against the ray's largest
               possible margin, without working out the margin of this node. */
            if ((mine_m & ~__builtin_amdgcn_ballot_w64(hd.r < neg_eb)) != 0ull) {
              bool c_acc, c_rej;
              rt_node_slab_tail(hd, tx0, tx1, ty0, ty1, tz0, tz1, hit.t, c_acc, c_rej);
              acc_m = __builtin_amdgcn_ballot_w64(c_acc) & mine_m;
              und_m = mine_m & ~(acc_m | __builtin_amdgcn_ballot_w64(c_rej));
            }
          })
        }
#ifndef RT_NO_ASM_SCAN
        else if (!COUNT && !PATHS) {
          // Zero direction pattern, common part in assembly as above: a child leaves it only if some ray of the wave has its
          // origin within the child's extent on every zero axis (rt_node_zero_outside says no for everybody otherwise).
#define RT_ZERO_AXIS_ASM(D0, D1, COMBINE)                                                                                  \
              "v_min_f32_e64 %[mn], -%[" D0 "], %[" D1 "]\n"                                                               \
              "v_max_f32_e64 %[mx], -%[" D0 "], %[" D1 "]\n"                                                               \
              "v_cmp_gt_f32 vcc, 0, %[mn]\n"                                                                               \
              "v_cmp_gt_f32_e64 %[sq], %[mx], %[side]\n"                                                                   \
              "s_and_b64 %[sq], %[sq], vcc\n" COMBINE
#define RT_ZERO_FIRST "s_mov_b64 %[nr], %[sq]\n"
#define RT_ZERO_NEXT "s_or_b64 %[nr], %[nr], %[sq]\n"
#define RT_ZERO_SCAN(ZX, ZY, ZZ, AXES_ASM)                                                                                  \
          for (;;) {                                                                                                       \
            uint32_t prev, tmp;                                                                                            \
            uint64_t notrej, sq;                                                                                           \
            float d0, d1, d2, d3, d4, d5, side, mn, mx;                                                                    \
            const uint64_t blk_bits = (uint64_t)blk;                                                                       \
            asm volatile(                                                                                                  \
                "L_zscan_%=:\n"                                                                                            \
                "s_waitcnt lgkmcnt(0)\n"                                                                                   \
                "v_sub_f32 %[d0], s84, %[ex]\n"                                                                            \
                "v_sub_f32 %[d1], s85, %[ex]\n"                                                                            \
                "v_sub_f32 %[d2], s86, %[ey]\n"                                                                            \
                "v_sub_f32 %[d3], s87, %[ey]\n"                                                                            \
                "v_sub_f32 %[d4], s88, %[ez]\n"                                                                            \
                "v_sub_f32 %[d5], s89, %[ez]\n"                                                                            \
                "v_mov_b32 %[side], s84\n"                                                                                 \
                "v_sub_f32 %[side], s85, %[side]\n"                                                                        \
                "s_mov_b32 %[first], s90\n"                                                                                \
                "s_mov_b32 %[meta], s91\n"                                                                                 \
                "s_mov_b32 %[prev], %[rem]\n"                                                                              \
                "s_add_i32 %[tmp], %[rem], -1\n"                                                                           \
                "s_and_b32 %[rem], %[rem], %[tmp]\n"                                                                       \
                "s_or_b32 %[tmp], %[rem], 0x100\n"                                                                         \
                "s_ff1_i32_b32 %[tmp], %[tmp]\n"                                                                           \
                "s_lshl_b32 %[tmp], %[tmp], 5\n"                                                                           \
                "s_load_dwordx8 s[84:91], %[blk], %[tmp] offset:0x0\n"                                                     \
                AXES_ASM                                                                                                   \
                "s_andn2_b64 %[nr], %[mine], %[nr]\n"                                                                      \
                "s_cbranch_scc1 L_zdone_%=\n"                                                                              \
                "s_cmp_lg_u32 %[rem], 0\n"                                                                                 \
                "s_cbranch_scc1 L_zscan_%=\n"                                                                              \
                "L_zdone_%=:\n"                                                                                            \
                "s_waitcnt lgkmcnt(0)\n"                                                                                   \
                : [nx] "+{s[84:91]}"(nx), [rem] "+s"(rem), [first] "=&s"(r_first), [meta] "=&s"(r_meta), [prev] "=&s"(prev),   \
                  [tmp] "=&s"(tmp), [nr] "=&s"(notrej), [sq] "=&s"(sq), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2),      \
                  [d3] "=&v"(d3), [d4] "=&v"(d4), [d5] "=&v"(d5), [side] "=&v"(side), [mn] "=&v"(mn), [mx] "=&v"(mx)         \
                : [blk] "s"(blk_bits), [mine] "s"(mine_m), [ex] "v"(e.x), [ey] "v"(e.y), [ez] "v"(e.z)                      \
                : "vcc", "scc");                                                                                           \
            rem_cur = uni(prev);                                                                                           \
            acc_m = 0ull; und_m = 0ull;                                                                                    \
            if (notrej == 0ull) break;   /* the last child, beside everybody */                                            \
            {                                                                                                              \
              const float dlo[3] = {d0, d2, d4}, dhi[3] = {d1, d3, d5};                                                    \
              bool c_acc, c_rej;                                                                                           \
              rt_node_slab_zero_d<ZX, ZY, ZZ>(dlo, dhi, side, inv, hit.t, c_acc, c_rej);                                   \
              acc_m = __builtin_amdgcn_ballot_w64(c_acc) & mine_m;                                                         \
              und_m = mine_m & ~(acc_m | __builtin_amdgcn_ballot_w64(c_rej));                                              \
            }                                                                                                              \
            if ((acc_m | und_m) != 0ull || rem == 0u) break;                                                               \
          }
          switch (zpat) {
            case 1u: RT_ZERO_SCAN(true, false, false, RT_ZERO_AXIS_ASM("d0", "d1", RT_ZERO_FIRST)) break;
            case 2u: RT_ZERO_SCAN(false, true, false, RT_ZERO_AXIS_ASM("d2", "d3", RT_ZERO_FIRST)) break;
            case 3u: RT_ZERO_SCAN(true, true, false, RT_ZERO_AXIS_ASM("d0", "d1", RT_ZERO_FIRST) RT_ZERO_AXIS_ASM("d2", "d3", RT_ZERO_NEXT)) break;
            case 4u: RT_ZERO_SCAN(false, false, true, RT_ZERO_AXIS_ASM("d4", "d5", RT_ZERO_FIRST)) break;
            case 5u: RT_ZERO_SCAN(true, false, true, RT_ZERO_AXIS_ASM("d0", "d1", RT_ZERO_FIRST) RT_ZERO_AXIS_ASM("d4", "d5", RT_ZERO_NEXT)) break;
            default: RT_ZERO_SCAN(false, true, true, RT_ZERO_AXIS_ASM("d2", "d3", RT_ZERO_FIRST) RT_ZERO_AXIS_ASM("d4", "d5", RT_ZERO_NEXT)) break;
          }
#undef RT_ZERO_SCAN
#undef RT_ZERO_AXIS_ASM
#undef RT_ZERO_FIRST
#undef RT_ZERO_NEXT
        }
#endif
        else {
          switch (zpat) {
            case 1u: RT_SCAN_LOOP(const float side = nx[1] - nx[0], RT_CLASSIFY_ZERO(true, false, false)) break;
            case 2u: RT_SCAN_LOOP(const float side = nx[1] - nx[0], RT_CLASSIFY_ZERO(false, true, false)) break;
            case 3u: RT_SCAN_LOOP(const float side = nx[1] - nx[0], RT_CLASSIFY_ZERO(true, true, false)) break;
            case 4u: RT_SCAN_LOOP(const float side = nx[1] - nx[0], RT_CLASSIFY_ZERO(false, false, true)) break;
            case 5u: RT_SCAN_LOOP(const float side = nx[1] - nx[0], RT_CLASSIFY_ZERO(true, false, true)) break;
            default: RT_SCAN_LOOP(const float side = nx[1] - nx[0], RT_CLASSIFY_ZERO(false, true, true)) break;
          }
        }
#undef RT_SCAN_LOOP
#undef RT_CLASSIFY_ZERO
        na = float4{0.f, 0.f, 0.f, 0.f}; nb = float4{0.f, 0.f, 0.f, 0.f};
        if (und_m) {   // the exact test wants the planes again
          const uint32_t c_cur = (uint32_t)__builtin_ctz(rem_cur);
          na = rec(blk + NS * c_cur); nb = rec(blk + NS * c_cur + 1);
        }
        nb.z = __uint_as_float(r_first); nb.w = __uint_as_float(r_meta);
      } else {
        rem &= rem - 1u;
#ifdef RT_PATH_GENERIC
        tl.w[3]++;
#endif
        na = float4{nx[0], nx[1], nx[2], nx[3]}; nb = float4{nx[4], nx[5], nx[6], nx[7]};
        { const uint32_t c = (uint32_t)__builtin_ctz(rem | 0x100u); nx = rec8(blk + NS * c); }
        if (PATHS && (uni(__float_as_uint(nb.w)) & 0x800000ffu) != 0u) tl.w[1]++;
        if (COUNT && ((mine_m >> lane) & 1ull)) tl.node++;
        if (!any_zero) {
          bool c_acc, c_rej;
          rt_node_slab_fast(na.x, na.y, na.z, na.w, nb.x, nb.y, e, inv, hit.t, c_acc, c_rej);
          acc_m = __builtin_amdgcn_ballot_w64(c_acc) & mine_m;
          und_m = mine_m & ~(acc_m | __builtin_amdgcn_ballot_w64(c_rej));
        } else {
          bool c_acc, c_und;
          rt_node_slab_class2({na.x, na.z, nb.x}, {na.y, na.w, nb.y}, e, inv, hit.t, true, zmask, c_acc, c_und);
          acc_m = __builtin_amdgcn_ballot_w64(c_acc) & mine_m;
          und_m = __builtin_amdgcn_ballot_w64(c_und) & mine_m;
        }
      }
      if (und_m) {
#ifndef RT_PATH_HIST
        if (PATHS) tl.w[2]++;
#endif
        // Rare path (about 1 % of the node tests): the per-ray constants of the exact test are derived here, from
        // values the compiler cannot hoist, instead of occupying registers across the whole walk.
        RtVec3 dd = d, ee = e; float T = hit.t;
        asm volatile("" : "+v"(dd.x), "+v"(dd.y), "+v"(dd.z), "+v"(ee.x), "+v"(ee.y), "+v"(ee.z), "+v"(T));
        const RtRayConst rc = rt_ray_const(dd);
        // corners in the reference's convention (ini = lo.x lo.y hi.z, end = hi.x hi.y lo.z) out of the variant's pairs
        const bool wx = (variant & 1u) != 0u, wy = (variant & 2u) != 0u, wz = (variant & 4u) != 0u;
        const RtVec3 ini = {wx ? na.y : na.x, wy ? na.w : na.z, wz ? nb.x : nb.y}, end = {wx ? na.x : na.y, wy ? na.z : na.w, wz ? nb.y : nb.x};
        const float S = rt_node_score(ini, end, ee, dd, rc, rt_t_float(rc, T));
        uint64_t pos_m = __builtin_amdgcn_ballot_w64(S > 0.f);
        const uint64_t tie_m = __builtin_amdgcn_ballot_w64(S == 0.f) & und_m;
        if (tie_m) {   // a boundary case: settle it with the compare form
          if (PATHS) tl.w[3]++;
          const bool ex = rt_node_accept_exact(ini, end, ee, dd, rc, rt_t_const(rc, T));
          pos_m = (pos_m & ~tie_m) | (__builtin_amdgcn_ballot_w64(ex) & tie_m);
        }
        acc_m |= pos_m & und_m;
      }
      if (!acc_m) continue;
      const uint32_t first = uni(__float_as_uint(nb.z));
      const uint32_t meta = uni(__float_as_uint(nb.w));
      if (meta & 0x80000000u) {   // branch: descend; the parent is only kept if it has children left
        uint32_t crem = COUNT ? 0xffu : (meta & 0xffu);
#ifdef RT_WAVE_CULL
        if (cull_on) {
          // Child c in lane c (and in every eighth lane after it): bounds, over all rays of the wave, of the child's
          // near-plane distances from below and far-plane distances from above, by the lanes' own operations on the
          // interval ends (rounding is monotone).  If min(span, tF) of those bounds lies below every lane's -Eb, every
          // lane would reject the child in the scan loop: it is left out of the scan.
          uint32_t ones = ~0u;
          asm volatile("" : "+s"(ones));
          const uint32_t c = __builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u)) & 7u;
          typedef float f4v __attribute__((ext_vector_type(4)));
          const f4v* bl = (const f4v*)cull_lds;
          asm volatile("" : "+v"(bl));   // read here, every time: not carried in registers across the walk
          const f4v bA = bl[0], bB = bl[1], bS = bl[2], bL = bl[3];
          const float cu_eA[3] = {bA.x, bA.y, bA.z}, cu_eB[3] = {bB.x, bB.y, bB.z}, cu_iS[3] = {bS.x, bS.y, bS.z}, cu_iL[3] = {bL.x, bL.y, bL.z};
          const float cu_neb = bA.w;
          // the children's records by a vector load, eight lanes wide (deriving them from the parent's planes - a child
          // is a half of the parent per axis, octtree.cu:95-111 - costs a scalar reload and 18 more instructions: slower)
          const float4 ca = nodes[NS * (first + c)], cb = nodes[NS * (first + c) + 1];   // (near, far) per axis in this variant's copy
          const float cnx = ca.x, cfx = ca.y, cny = ca.z, cfy = ca.w, cnz = cb.x, cfz = cb.y;
          const float anx = cnx - cu_eA[0], afx = cfx - cu_eB[0], any_ = cny - cu_eA[1], afy = cfy - cu_eB[1], anz = cnz - cu_eA[2], afz = cfz - cu_eB[2];
          const float tn = rt_max3(fminf(anx * cu_iS[0], anx * cu_iL[0]), fminf(any_ * cu_iS[1], any_ * cu_iL[1]), fminf(anz * cu_iS[2], anz * cu_iL[2]));
          const float tf = rt_min3(fmaxf(afx * cu_iS[0], afx * cu_iL[0]), fmaxf(afy * cu_iS[1], afy * cu_iL[1]), fmaxf(afz * cu_iS[2], afz * cu_iL[2]));
          const float bound = fminf(tf - tn, tf);
          crem &= ~((uint32_t)__builtin_amdgcn_ballot_w64(bound < cu_neb) & 0xffu);
        }
#endif
        if (crem == 0u) continue;
        if (rem != 0u) {
          // lane sp keeps entry sp.  The lane number is counted afresh here (two instructions on an opaque mask) rather
          // than kept in a register, or fetched from a spill slot, during the whole walk.
          uint32_t ones = ~0u;
          asm volatile("" : "+s"(ones));
          const bool slot = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u)) == sp;
          st.first = slot ? top_first : st.first; st.cur = slot ? rem : st.cur;
          st.mask_lo = slot ? (uint32_t)top_mask : st.mask_lo; st.mask_hi = slot ? (uint32_t)(top_mask >> 32) : st.mask_hi;
          sp++;
        }
        top_first = first; rem = crem; top_mask = acc_m; mine_m = acc_m;
        blk = cnodes + NS * top_first;
        const uint32_t c = (uint32_t)__builtin_ctz(rem);
        nx = rec8(blk + NS * c);
        continue;
      }
      const int32_t count = (int32_t)(meta & 0xffu);   // (bits 24..26: the record's position among its siblings)
      // leaf: every triangle in insertion order (octtree.cu:264-274); all lanes in acc_m test the same triangle
      if (COUNT && ((acc_m >> lane) & 1ull)) { tl.leaf++; tl.tri += (uint32_t)count; }
      if (count == 0) continue;
#ifdef RT_PATH_HIST
      // measurement build: triangle iterations by how many lanes of the wave take part (w[5]: one lane, w[6]: 2-4, w[7]: 5-16;
      // the stage counters are not kept then)
      if (PATHS) { const int pc = __builtin_popcountll(acc_m); tl.w[pc == 1 ? 5 : (pc <= 4 ? 6 : (pc <= 16 ? 7 : 2))] += (uint32_t)count; }
#endif
      uint64_t changed_m = 0ull;
      // One triangle: the record is in SGPRs (q0..q3 as laid out in RtTriLeaf), every lane of acc_m tests it.
      auto test_triangle = [&](const float4& q0, const float4& q1, const float4& q2, const float4& q3) {
        if (PATHS) tl.w[4]++;
        // stage 1: determinant and beta numerator (triangle.cu:117-131, first column of cofactors); the operations and
        // their order are the reference's, paired two to an instruction
        const v2f bxy = e_xy - v2f{q0.x, q0.y};
        float bz = e.z - q0.z;
        asm volatile("" : "+v"(bz));   // keeps this subtraction from being paired with the one that forms m20 (the pairing costs three moves)
        const v2f p1 = v2f{q2.z, q2.w} * c_zx;          // e2.y c2.z , e2.z c2.x
        const v2f p2 = v2f{q3.x, q3.y} * c_yz;          // e2.z c2.y , e2.x c2.z
        const v2f m01 = p1 - p2;                        // m00 , m10
        const v2f p3 = v2f{q2.x, q2.y} * c_yx;          // e2.x c2.y , e2.y c2.x
        const float m20 = p3.x - p3.y;
        const v2f qd = v2f{q1.x, q1.y} * m01;
        const v2f qb = m01 * bxy;
        RtTriNum n;
        n.det = (qd.x + qd.y) + q1.z * m20;
        n.nb = (qb.x + qb.y) + m20 * bz;
        const unsigned s = rt_bits(n.det) & 0x80000000u;
        const float ad = fabsf(n.det);
        const float nbs = rt_from_bits(rt_bits(n.nb) ^ s);
        // 0 <= beta <= 1  <=>  0 <= nbs <= |det|   (det == 0 cannot survive stage 3: 0 < nts <= 0)
        uint64_t pass_m = __builtin_amdgcn_ballot_w64(nbs >= 0.f) & __builtin_amdgcn_ballot_w64(nbs <= ad) & acc_m;
        if (!pass_m) return;
#ifndef RT_PATH_HIST
        if (PATHS) tl.w[5]++;
#endif
        const RtVec3 e1 = {q1.x, q1.y, q1.z}, e2 = {q2.x, q2.y, q2.w};
        const RtVec3 b = {bxy.x, bxy.y, bz};
        // stage 2: gamma numerator
        const float m01s = -e1.y * c2.z + e1.z * c2.y;
        const float m11 = e1.x * c2.z - e1.z * c2.x;
        const float m21 = -e1.x * c2.y + e1.y * c2.x;
        n.ng = m01s * b.x + m11 * b.y + m21 * b.z;
        const float ngs = rt_from_bits(rt_bits(n.ng) ^ s);
        pass_m &= __builtin_amdgcn_ballot_w64(ngs >= 0.f) & __builtin_amdgcn_ballot_w64(ngs <= ad);
        if (!pass_m) return;
#ifndef RT_PATH_HIST
        if (PATHS) tl.w[6]++;
#endif
        // stage 3: t numerator; t > 0, and t clearly beyond the closest hit is rejected without dividing
        const float m02 = e1.y * e2.z - e1.z * e2.y;
        const float m12 = -e1.x * e2.z + e1.z * e2.x;
        const float m22 = e1.x * e2.y - e1.y * e2.x;
        n.nt = m02 * b.x + m12 * b.y + m22 * b.z;
        const float nts = rt_from_bits(rt_bits(n.nt) ^ s);
        pass_m &= __builtin_amdgcn_ballot_w64(nts > 0.f) & __builtin_amdgcn_ballot_w64(nts <= (hit.t * ad) * 1.000001f);
        if (!pass_m) return;
#ifndef RT_PATH_HIST
        if (PATHS) tl.w[7]++;
#endif
        if ((pass_m >> lane) & 1ull) {   // the reference's quotients decide (triangle.cu:131-137)
          float t, bet, gam;
          if (rt_triangle_finish(n, hit.t, t, bet, gam)) { hit.t = t; hit.prim = (int32_t)__float_as_uint(q0.w); hit.bet = bet; hit.gam = gam; }
        }
        changed_m |= pass_m;   // a superset of the lanes whose t changed is enough
      };
      // The leaf's records are streamed through two register sets in turn, each refilled with the record after next
      // while the other is being tested (both arrays end with a pad record, so reading one past the leaf is harmless).
      // One scalar request per record: its 64 bytes as one 16-word value through the constant address space (as three
      // loads of 8 + 4 + 2 words the frame was 0.6 % slower: the scalar caches count requests).
      typedef float f16v __attribute__((ext_vector_type(16)));
      typedef const __attribute__((address_space(4))) f16v* ctri_ptr;
      ctri_ptr tp = (ctri_ptr)(uintptr_t)(tri_leaf + 4 * (size_t)first);
      auto test16 = [&](const f16v& r) {
        test_triangle(float4{r[0], r[1], r[2], r[3]}, float4{r[4], r[5], r[6], r[7]}, float4{r[8], r[9], r[10], r[11]}, float4{r[12], r[13], r[14], r[15]});
      };
      f16v ra = tp[0];
      for (int32_t k = count;;) {
        const f16v rb = tp[1];
        test16(ra);
        if (--k == 0) break;
        tp += 2;
        ra = tp[0];
        test16(rb);
        if (--k == 0) break;
      }
      if (changed_m) {
        const bool done = stop_t >= 0.f && hit.t < stop_t;
        live &= ~__builtin_amdgcn_ballot_w64(done);
        if (!live) return;
        mine_m = top_mask & live;
        if (mine_m == 0ull) rem = 0u;   // nobody left in this branch
      }
    }
#ifndef RT_NO_ASM_WALK
    }
#endif
  }
  if (n_spheres > 0) {
    bool mine_live = ((live >> lane) & 1ull) != 0ull;
    for (int32_t s = 0; s < n_spheres; s++) {   // ray.cu:52-53
      if (mine_live) {
        if (COUNT) tl.sph++;
        if (rt_sphere_test(ld3(spheres[s].c), spheres[s].r, e, d, hit.t)) {
          hit.prim = -2 - s;
          if (stop_t >= 0.f && hit.t < stop_t) mine_live = false;
        }
      }
    }
  }
}

// Closest hit through the assembly walk for one class of the wave's pending rays (the production variants without bounce
// rays).  The walk wants rays that agree on their direction signs, or share a pattern of zero direction components; of a
// wave whose rays do not (the tiles on the picture's centre lines, a few shadow waves) only the class of the first
// pending lane is traced.  Returns the lanes that were traced; the others keep their rays for the caller's next round.
// A lane's result never depends on which other lanes walk with it, so the hits are those of any other grouping.
// The barycentrics of a triangle hit are left in LDS (see walk_asm), not in `hit`.
// A direction component so small that its inverse overflows (or a NaN) cannot feed the fused plane distances: such a
// lane walks with a NaN inverse direction, which the conservative classification can decide nothing about - every node
// test of the lane goes to the exact test, which does not use the inverse.
template <bool CULLABLE>
__device__ __forceinline__ uint64_t trace_class(const float4* __restrict__ nodes, int n_wnodes, const float4* __restrict__ tri_leaf,
                                                const RtSphere* __restrict__ spheres, int n_spheres, bool has_oct, bool want,
                                                const RtVec3& e, const RtVec3& d, float stop_t, Hit& hit, WaveStack& st, float* bary_lds,
                                                uint32_t cull_offset, float cull_r0_max, float cull_ksp_max) {
  const uint64_t pending = __builtin_amdgcn_ballot_w64(want);
  if (!pending) return 0ull;
  uint64_t cls = pending, after = pending;
  if (has_oct) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) f4* cnode_ptr;
    const f4 root_b = *((cnode_ptr)(uintptr_t)nodes + 1);   // (links and child mask of the root are the same in every variant)
    const uint32_t top_first = uni(__float_as_uint(root_b.z)), rem = uni(__float_as_uint(root_b.w)) & 0xffu;
    if (rem != 0u) {
      RtVec3 inv = {1.f / d.x, 1.f / d.y, 1.f / d.z};   // only feeds the conservative classification
      unsigned zmask = (d.x == 0.f ? 1u : 0u) | (d.y == 0.f ? 2u : 0u) | (d.z == 0.f ? 4u : 0u);
      const float INF = __builtin_inff();
      const bool unusable = zmask == 7u || (!(zmask & 1u) && !(fabsf(inv.x) < INF)) || (!(zmask & 2u) && !(fabsf(inv.y) < INF)) ||
                            (!(zmask & 4u) && !(fabsf(inv.z) < INF));
      if (unusable) { const float nan = __builtin_nanf(""); inv = {nan, nan, nan}; if (zmask == 7u) zmask = 0u; }
      // class of a lane: its zero pattern if it has one (the signs of the other components may differ within a class),
      // else its three direction signs
      const uint32_t key = zmask ? (8u | zmask) : ((d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u));
      const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, __builtin_ctzll(pending));
      cls = __builtin_amdgcn_ballot_w64(key == k0) & pending;
      const uint32_t zpat = (k0 & 8u) ? (k0 & 7u) : 0u, variant = (k0 & 8u) ? 0u : k0;
      const float4* const vnodes = nodes + 2 * (size_t)variant * (size_t)n_wnodes;   // each variant is a dense copy of the whole array
      const RtSlabFused fu = rt_slab_fused(e, inv);
      float neg_eb = 0.f;
      if (zpat == 0u) {   // margin that no node of this tree exceeds for this ray (the root's planes bound every node's)
        const f4 ra = *(cnode_ptr)(uintptr_t)vnodes, rb = *((cnode_ptr)(uintptr_t)vnodes + 1);
        neg_eb = -rt_slab_margin_bound_fused(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, e, inv, fu, hit.t);
      }
      RtBundle bundle = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const uint32_t cull = (CULLABLE && cull_offset != 0u && wave_bundle(cls, e, d, stop_t, bundle, cull_r0_max, cull_ksp_max)) ? cull_offset : 0u;
      after = walk_asm<CULLABLE, true>(vnodes, variant, zpat, tri_leaf, top_first, rem, cls, e, d, inv, fu.c, neg_eb, stop_t, hit, st, bary_lds, cull, bundle);
    }
  }
  if (n_spheres > 0 && after) {
    bool mine_live = in_mask(after);
    for (int32_t s = 0; s < n_spheres; s++) {   // ray.cu:52-53
      if (mine_live && rt_sphere_test(ld3(spheres[s].c), spheres[s].r, e, d, hit.t)) {
        hit.prim = -2 - s;
        if (stop_t >= 0.f && hit.t < stop_t) mine_live = false;
      }
    }
  }
  return cls;
}

// Closest hit with every lane walking its OWN ray (measurement builds only, -DRT_LANE_WALK: see render_wave for what it
// measured).  The idea: for rays that have little in common (a queue of bounce rays that scatter inside a mesh) the
// shared walk costs the union of the 64 subtrees per wave; here the lanes walk side by side, each with its own node and
// triangle loads.  No stack: the reference visits children in the fixed order 0..7 whatever the ray, so where a walk
// goes when it is done with a node's subtree is a property of the node (RtDeviceView::node_skip, built by the host) -
// an accepted branch leads to its first child that is not an empty leaf, everything else to the link.  A lane is either
// at a node or inside a leaf's triangle run; one round of the loop takes every lane one step (a node test or a triangle
// test), with the lane's own running closest t, so the decisions are those of the lane's own depth-first walk in the
// reference's order.  The predicates are the ones the shared walk uses in compiled form (rt_math.h: conservative slab
// classification, exact score / compare form where it abstains, numerators + quotients for triangles).
__device__ __forceinline__ void trace_lanes(const float4* __restrict__ nodes, const uint32_t* __restrict__ node_skip, const float4* __restrict__ tri_leaf,
                                            const RtSphere* __restrict__ spheres, int n_spheres, bool has_oct, bool want,
                                            const RtVec3& e, const RtVec3& d, float stop_t, Hit& hit) {
  constexpr uint32_t END = 0xffffffffu;
  bool live = want;   // (a shadow ray leaves as soon as it is known to be blocked)
  if (has_oct) {
    const RtVec3 inv = {1.f / d.x, 1.f / d.y, 1.f / d.z};   // only feeds the conservative classification
    const bool zx = d.x == 0.f, zy = d.y == 0.f, zz = d.z == 0.f, any_zero = zx | zy | zz;
    uint32_t node = END, tri = 0u, tri_end = 0u;   // the lane is inside a leaf while tri < tri_end
    {   // the root is entered without a test (ray.cu:36-38)
      const float4 root_b = nodes[1];
      const uint32_t first = __float_as_uint(root_b.z), occupied = __float_as_uint(root_b.w) & 0xffu;
      if (want && occupied) node = first + (uint32_t)__builtin_ctz(occupied);
    }
    while (__builtin_amdgcn_ballot_w64(node != END || tri < tri_end)) {
      if (tri < tri_end) {   // one triangle of the leaf (octtree.cu:264-274: all of them, in insertion order)
        const float4* q = tri_leaf + 4 * (size_t)tri;
        const float4 q0 = q[0], q1 = q[1], q2 = q[2];
        const RtTriNum n = rt_triangle_numerators({q0.x, q0.y, q0.z}, {q1.x, q1.y, q1.z}, {q2.x, q2.y, q2.w}, e, d);
        tri++;
        float t, bet, gam;
        if (rt_triangle_candidate(n) && rt_triangle_finish(n, hit.t, t, bet, gam)) {
          hit.t = t; hit.prim = (int32_t)__float_as_uint(q0.w); hit.bet = bet; hit.gam = gam;
          if (stop_t >= 0.f && t < stop_t) { live = false; node = END; tri = tri_end; }   // ray.cu:70 only asks whether something is closer than the light
        }
      } else if (node != END) {
        const float4 a = nodes[2 * (size_t)node], b = nodes[2 * (size_t)node + 1];
        const uint32_t after = node_skip[node];
        const RtVec3 lo = {a.x, a.z, b.x}, hi = {a.y, a.w, b.y};
        const int cls = rt_node_slab_class(lo, hi, e, inv, hit.t, any_zero, zx, zy, zz);
        bool accepted = cls > 0;
        if (cls == 0) {   // within rounding of a boundary: the exact form decides (octtree.cu:276-321)
          const RtRayConst rc = rt_ray_const(d);
          const RtVec3 ini = {lo.x, lo.y, hi.z}, end = {hi.x, hi.y, lo.z};
          const float S = rt_node_score(ini, end, e, d, rc, rt_t_float(rc, hit.t));
          accepted = S > 0.f;
          if (S == 0.f) accepted = rt_node_accept_exact(ini, end, e, d, rc, rt_t_const(rc, hit.t));
        }
        const uint32_t first = __float_as_uint(b.z), meta = __float_as_uint(b.w);
        node = after;
        if (accepted) {
          if (meta & 0x80000000u) { const uint32_t occupied = meta & 0xffu; if (occupied) node = first + (uint32_t)__builtin_ctz(occupied); }
          else { tri = first; tri_end = first + (meta & 0xffu); }   // a leaf: its triangles, then on to `after`
        }
      }
    }
  }
  if (n_spheres > 0) {
    for (int32_t s = 0; s < n_spheres; s++) {   // ray.cu:52-53
      if (live && rt_sphere_test(ld3(spheres[s].c), spheres[s].r, e, d, hit.t)) {
        hit.prim = -2 - s;
        if (stop_t >= 0.f && hit.t < stop_t) live = false;
      }
    }
  }
}

enum { PH_CLOSEST = 0, PH_SHADOW = 1, PH_DONE = 2, PH_SHADE = 3 };

#ifndef RT_WAVES_PER_SIMD_BOUNCE
#define RT_WAVES_PER_SIMD_BOUNCE 4   // the bounce variant carries more state; at 5 it spills inside the triangle loop
#endif
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 5
#endif
#ifndef RT_WAVES_PER_SIMD_QPRIMARY
#define RT_WAVES_PER_SIMD_QPRIMARY 4   // (glass bunny 4K: 5.18 ms at five waves, 5.11 at four: the dielectric shading code wants the registers)
#endif
#ifndef RT_WAVES_PER_SIMD_SPHERES
#define RT_WAVES_PER_SIMD_SPHERES 5          // the variants for scenes without a mesh
#endif
#ifndef RT_WAVES_PER_SIMD_SPHERES_BOUNCE
#define RT_WAVES_PER_SIMD_SPHERES_BOUNCE 4
#endif
#define RT_WAVES_PER_SIMD_QPRIMARY_OR_PLAIN(q) ((q) ? RT_WAVES_PER_SIMD_QPRIMARY : RT_WAVES_PER_SIMD)

// Shading state of a lane.  It is dead while the wave traverses the octree, and the traversal is where
// the registers are needed, so it is parked in LDS ([word][thread], conflict-free) around every trace.
template <bool BOUNCE>
struct ShadeState {
  RtVec3 P, N, view;                 // hit point, shading normal, direction of the ray being shaded
  RtVec3 colr, color;                // colour of the ray being shaded, colour of the pixel so far
  float hit_t;                       // distance of the hit being shaded (Beer attenuation)
  int32_t mfirst, mcount, mi, li;    // material list of the hit primitive, (material, light) cursor
  uint32_t vis_known, vis_occluded;  // per light (first 32): shadow test already traced for this hit point
  float tu, tv;                      // texture coordinates of the hit (variants for scenes with texture materials only)
  RtVec3 att; float level0, level1, level2; int32_t nlevel, depth;   // BOUNCE only (no arrays: the state must stay in registers)
  __device__ __forceinline__ float level_at(int i) const {
    float a = level0, b = level1, c = level2;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c));   // keeps this a select of values: a select of addresses would pin the struct in scratch memory
    return i == 0 ? a : (i == 1 ? b : c);
  }
  __device__ __forceinline__ void set_level(int i, float v) { level0 = i == 0 ? v : level0; level1 = i == 1 ? v : level1; level2 = i == 2 ? v : level2; }
};

// One statement per word (no temporary array: the state has to stay in registers on both sides).
#define RT_SHADE_FLOATS(X) X(0, P.x) X(1, P.y) X(2, P.z) X(3, N.x) X(4, N.y) X(5, N.z) X(6, view.x) X(7, view.y) X(8, view.z) \
  X(9, colr.x) X(10, colr.y) X(11, colr.z) X(12, hit_t)
#define RT_SHADE_INTS(X) X(13, mfirst) X(14, mcount) X(15, mi) X(16, li) X(17, vis_known) X(18, vis_occluded)
#define RT_SHADE_BOUNCE_FLOATS(X) X(19, att.x) X(20, att.y) X(21, att.z) X(22, level0) X(23, level1) X(24, level2) X(25, color.x) X(26, color.y) X(27, color.z)
#define RT_SHADE_BOUNCE_INTS(X) X(28, nlevel) X(29, depth)
// QPRIMARY (the primary-ray pass of a frame whose bounce rays go to queues): of the bounce words only the refraction
// levels are kept (a dielectric hit writes them, the hit's further materials read them); attenuation, depth and level
// count are those of a primary ray throughout, and the pixel's colour is finished by its one ray or by the term slots.
template <bool BOUNCE, bool QPRIMARY> struct ParkedWords { static constexpr int N = !BOUNCE ? 19 : (QPRIMARY ? 22 : 30); };
template <bool BOUNCE, bool TEX, bool QPRIMARY>
__device__ __forceinline__ void park(const ShadeState<BOUNCE>& s, float* lds) {
  uint32_t* ldu = (uint32_t*)lds;
  constexpr int N = ParkedWords<BOUNCE, QPRIMARY>::N;
#define PUT_F(k, f) lds[(k) * RT_WBLOCK] = s.f;
#define PUT_I(k, f) ldu[(k) * RT_WBLOCK] = (uint32_t)s.f;
  RT_SHADE_FLOATS(PUT_F) RT_SHADE_INTS(PUT_I)
  if (BOUNCE && !QPRIMARY) { RT_SHADE_BOUNCE_FLOATS(PUT_F) RT_SHADE_BOUNCE_INTS(PUT_I) }
  if (BOUNCE && QPRIMARY) { PUT_F(19, level0) PUT_F(20, level1) PUT_F(21, level2) }
  if (TEX) { PUT_F(N, tu) PUT_F(N + 1, tv) }
#undef PUT_F
#undef PUT_I
}
template <bool BOUNCE, bool TEX, bool QPRIMARY>
__device__ __forceinline__ void unpark(ShadeState<BOUNCE>& s, const float* lds) {
  const uint32_t* ldu = (const uint32_t*)lds;
  constexpr int N = ParkedWords<BOUNCE, QPRIMARY>::N;
#define GET_F(k, f) s.f = lds[(k) * RT_WBLOCK];
#define GET_I(k, f) s.f = (decltype(s.f))ldu[(k) * RT_WBLOCK];
  RT_SHADE_FLOATS(GET_F) RT_SHADE_INTS(GET_I)
  if (BOUNCE && !QPRIMARY) { RT_SHADE_BOUNCE_FLOATS(GET_F) RT_SHADE_BOUNCE_INTS(GET_I) }
  if (BOUNCE && QPRIMARY) { GET_F(19, level0) GET_F(20, level1) GET_F(21, level2) s.att = {1.f, 1.f, 1.f}; s.nlevel = 0; s.depth = 0; }
  if (TEX) { GET_F(N, tu) GET_F(N + 1, tv) }
#undef GET_F
#undef GET_I
}

// BOUNCE: the scene has reflective or dielectric materials; COUNT: instrumented; QUEUES (with BOUNCE): bounce rays leave
// through / arrive from the generation queues instead of living on the lane's stack (a variant of its own, so that
// the stacked kernel does not carry the queue code in its registers).
// The arguments of the kernel apart from the two arrays of the walk, as one block in the kernel-argument segment (behind
// those two pointers, which are parameters of their own so that their loads are known to be read-only).  What the walk needs is taken
// from it on entry.  Everything else (shading tables, output pointers, queues, frame geometry) is read where it is
// used, through LATE(): a load from the argument segment behind an opaque copy of its address, which the compiler can
// neither move in front of the walk nor keep in registers across it - the walk needs every scalar register it can get.
struct RtWaveArgs {
  RtCamera cam; RowMap rows;
  const float4* tri_geom; const float4* tri_shade;
  const RtSphere* spheres; const RtLight* lights; const RtMaterial* materials; const int32_t* mat_index;
  const int32_t* mesh_mat_first; const int32_t* mesh_mat_count;
  const float* tri_uv; const RtTexture* textures; const uint32_t* texels;   // scenes with texture materials only
  const uint32_t* node_skip;   // links of the stackless walk (trace_lanes)
  int n_spheres, n_lights, has_oct, n_wnodes;
  uint32_t cull_offset, pad_;   // bytes from tri_leaf to its RtCullRec array; 0: the walk tests every triangle (RT_AMD_NO_CULL, measurements)
  float cull_r0_max, cull_ksp_max;   // widest bundle of a wave's rays that is still put in front of a leaf's triangles (wave_bundle)
  uint8_t* rgb; int32_t* ids; float* ts; unsigned long long* counters_all; int tally;
  // Tiles of the launch and how they are dealt to the chip's dies (see tile_of): tiles per row / column, strips of
  // RT_STRIP_W tile columns per die (0: a plain two-dimensional grid, one workgroup per tile)
  uint32_t tiles_x, tiles_y, strips_per_die, centre_first;
  // ceil(2^32 / divisor) for the two divisions by launch constants that every wave makes (rows per band, tiles per strip):
  // n / d = the high word of n * magic for every n the launch can ask about (checked on the host); 0: divide
  uint32_t band_magic, strip_magic;
  RtQueues wf;
};
#define RT_WAVE_ARGS_OFFSET 16   // two pointers in front of the block
static_assert(alignof(RtWaveArgs) == 8, "the block follows the two pointers without padding");
typedef const __attribute__((address_space(4))) char* rt_kernarg_ptr;
__device__ __forceinline__ rt_kernarg_ptr late_args() {
  rt_kernarg_ptr p = (rt_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + RT_WAVE_ARGS_OFFSET;
  asm volatile("" : "+s"(p));
  return p;
}
#define LATE_AT(ka, field) (*(const __attribute__((address_space(4))) decltype(RtWaveArgs::field)*)((ka) + offsetof(RtWaveArgs, field)))
#define LATE(field) LATE_AT(late_args(), field)
// (the same for a whole member struct, word by word)
template <typename T>
__device__ __forceinline__ T late_struct(size_t offset) {
  static_assert(sizeof(T) % 4 == 0, "whole words");
  const __attribute__((address_space(4))) uint32_t* src = (const __attribute__((address_space(4))) uint32_t*)(late_args() + offset);
  uint32_t w[sizeof(T) / 4];
  for (size_t i = 0; i < sizeof(T) / 4; i++) w[i] = src[i];
  T v;
  __builtin_memcpy(&v, w, sizeof(T));
  return v;
}
#define LATE_STRUCT(field) late_struct<decltype(RtWaveArgs::field)>(offsetof(RtWaveArgs, field))
#ifdef RT_WAVE_TIMES_CLK
__device__ __forceinline__ float* rt_dbg_ts() { return LATE(ts); }
#endif

// DENSE (plain variant only): compiled for one more wave per SIMD, at the price of spilled registers in the shading code.
// It pays for big frames of a heavy mesh, whose walks dominate (meshDragon 4K: 2.71 -> 2.61 ms), not for small frames
// (1920x1080: 1.24 -> 1.28 ms) nor where shading weighs (sphere scenes: + 50 %); the launch chooses.
// TEX: the scene has texture materials (two more words of shading state, a texel fetch per material and light: variants
// of their own, so that every other scene runs the kernels it ran before).
// QPRIMARY (with QUEUES): the launch casts the primary rays; without it a QUEUES launch renders a queue.
// MESH = false: variants for scenes without a mesh (no octree): no walk is compiled in, and the shading state is not parked.
template <bool BOUNCE, bool COUNT, bool QUEUES, bool DENSE = false, bool TEX = false, bool QPRIMARY = false, bool MESH = true>
__global__ __launch_bounds__(RT_WBLOCK, !MESH ? (BOUNCE ? RT_WAVES_PER_SIMD_SPHERES_BOUNCE : RT_WAVES_PER_SIMD_SPHERES)
                                              : ((BOUNCE && !QPRIMARY) ? RT_WAVES_PER_SIMD_BOUNCE : (DENSE ? RT_WAVES_PER_SIMD + 1 : RT_WAVES_PER_SIMD_QPRIMARY_OR_PLAIN(QPRIMARY)))) void render_wave(
    const float4* __restrict__ nodes, const float4* __restrict__ tri_leaf, const RtWaveArgs A) {
  const RtSphere* __restrict__ spheres = A.spheres;
  const int n_spheres = A.n_spheres, has_oct = A.has_oct, n_wnodes = A.n_wnodes;
  auto counters_ptr = [&]() { return LATE(counters_all) + ((blockIdx.x + blockIdx.y) & (RT_COUNTER_COPIES - 1)) * RT_COUNTER_STRIDE; };
  // LDS of a workgroup: the parked shading state, the lane's three ray counters, its finished colour and the barycentrics of its hit; the packing area
  // of the epilogue lies over the first word of the state (nothing is parked any more then).  25 words per lane = 6 400 B
  // per wave (without bounce rays) leaves room for 24 waves per compute unit (six per SIMD) in the 160 KB.
  // The counters and the finished colour only ever live there: they are touched a few times per pixel, and as values that
  // every round of the main loop carries along they cost six registers (in the six-wave variant: six spill stores per
  // round) for nothing.
  constexpr int SW = ParkedWords<BOUNCE, QPRIMARY>::N + (TEX ? 2 : 0);   // words of parked shading state
  __shared__ float parked[(SW + 6) * RT_WBLOCK];
  uint32_t (*pack)[48] = (uint32_t (*)[48])parked;   // [wave][48]
  // The thread's index within its workgroup.  It is handed over in a register at the start and cannot be asked for again;
  // a workgroup of one wave derives it anew wherever it is needed (two instructions) instead of keeping it - and what is
  // derived from it, the thread's LDS addresses - alive across the walk.
#if RT_WBLOCK == 64
#define TID() fresh_lane_id()
#define LANE() fresh_lane_id()
#else
  const int tid_kept = threadIdx.x;
#define TID() tid_kept
#define LANE() (tid_kept & 63)
#endif
#define MY_PARK() (&parked[TID()])
#define MY_COUNTS() ((uint32_t*)MY_PARK() + SW * RT_WBLOCK)   /* rays, shadow rays, rays traced, finished colour */
  MY_COUNTS()[0] = 0u; MY_COUNTS()[RT_WBLOCK] = 0u; MY_COUNTS()[2 * RT_WBLOCK] = 0u; MY_COUNTS()[3 * RT_WBLOCK] = 0u;
  enum { N_RAYS = 0, N_SHADOW = 1, N_TRACED = 2 };
  auto count = [&](int which) { __hip_atomic_fetch_add(MY_COUNTS() + which * RT_WBLOCK, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  __shared__ float cull_sh[16 * (RT_WBLOCK / 64)];   // bounds of the wave's rays during a walk (trace_wave)
  WaveStack st = {0u, 0u, 0u, 0u};
  // Pixel of a thread: wave w of the block owns the 8x8 tile (w % RT_WAVES_X, w / RT_WAVES_X) of the block's pixels.  The
  // coordinates are needed at the start, after the primary ray and at the end; they are derived again each time
  // from an opaque copy of the thread index, so that none of them occupies a register (or a spill slot) in between.
  struct Pixel { int col, lrow, row, orow; bool valid; };   // (derived per use, see below)
  // Tile of this workgroup.  Workgroups go to the eight dies (XCDs) of the chip in turn, each die with its own L2: so
  // that a die keeps meeting the same part of the scene, workgroup i renders a tile of die i % 8's own strips - strips of
  // RT_STRIP_W tile columns, dealt to the dies round-robin across the picture (balanced, yet each die sees an eighth of
  // it) - one strip after the other, each from the top down, beginning with the die's strip nearest the middle of the
  // picture and going outwards on alternating sides: what is looked at tends to sit in the middle, its tiles hold the
  // longest waves, and a long wave that starts late is what the end of the launch waits for (headline frame: the longest
  // wave, 0.94 ms, started at 1.84 of 2.87 ms with the strips walked left to right; 2.56 -> 2.50-2.52 ms; the rows of a plain grid taken the same way: 4 % less at 1280x720, nothing at 1920x1080, slower at 512x512).  A slot beyond
  // the picture (ragged widths) has no tile: its lanes are invalid.  (Measured on the headline frame: 2.88 ms with a plain grid, 2.72 ms with the strips walked
  // row by row, 2.70 ms strip by strip; strips 2 or 8 columns wide and square blocks of 4 to 16 tiles: 2.71 to 2.84 ms.)
  struct Tile { uint32_t x, y; };
  auto const_div = [](uint32_t n, uint32_t d, uint32_t magic) -> uint32_t { return magic ? __umulhi(n, magic) : n / d; };
  auto band_row = [&](const RowMap& rows, int lrow) -> int {   // image row of local row lrow (rt_kernels.h, RowMap)
    const int band = (int)const_div((uint32_t)lrow, (uint32_t)rows.band_rows, LATE(band_magic));
    return rows.row0 + band * rows.band_stride + (lrow - band * rows.band_rows);
  };
  auto tile_of = [&]() {
    Tile t = {blockIdx.x, blockIdx.y};
    const uint32_t spd = LATE(strips_per_die);
    if (spd != 0u) {
      const uint32_t die = blockIdx.x % RT_DIES, j = blockIdx.x / RT_DIES;
      const uint32_t per_strip = RT_STRIP_W * LATE(tiles_y);
      uint32_t strip = const_div(j, per_strip, LATE(strip_magic));
      const uint32_t in_strip = j - strip * per_strip;
      if (LATE(centre_first)) { const uint32_t off = (strip + 1u) >> 1; strip = (strip & 1u) ? (spd >> 1) - off : (spd >> 1) + off; }
      t.y = in_strip / RT_STRIP_W;
      t.x = (strip * RT_DIES + die) * RT_STRIP_W + in_strip % RT_STRIP_W;
    }
    return t;
  };
  auto pixel_of = [&](int t) {
    asm volatile("" : "+v"(t));
    const int w = t >> 6, l = t & 63;
    const Tile tile = tile_of();
    const RowMap rows = LATE_STRUCT(rows);
    const int W = LATE(cam).width, H = LATE(cam).height;
    Pixel p;
    if (rows.small) {
      p.col = tile.x * (RT_WTILE_W / 2) + (w % RT_WAVES_X) * 4 + (l & 3);
      p.lrow = tile.y * (RT_WTILE_H / 2) + (w / RT_WAVES_X) * 4 + ((l >> 2) & 3);
    } else {
      p.col = tile.x * RT_WTILE_W + (w % RT_WAVES_X) * 8 + (l & 7);
      p.lrow = tile.y * RT_WTILE_H + (w / RT_WAVES_X) * 8 + (l >> 3);
    }
    p.row = band_row(rows, p.lrow);
    p.orow = rows.compact ? p.lrow : p.row;
    p.valid = p.col < W && p.lrow < rows.n_local && p.row < H && (!rows.small || l < 16);
    return p;
  };
  Tally tl = {0, 0, 0, 0, 0, 0, 0, {0, 0, 0, 0, 0, 0, 0, 0}};
  const RtCamera& c = A.cam;   // (entry only: primary ray set-up)
#ifdef RT_WAVE_TIMES
  // Diagnostic build: every wave leaves (start, end) of its life on the 100 MHz clock and where it ran in the `t`
  // output, four words per wave in launch order (tools/wave_times.py); the per-pixel t is not written then.
  const unsigned long long wt_start = __builtin_amdgcn_s_memrealtime();
#ifdef RT_WAVE_TIMES_CLK
  const unsigned long long wt_clk0 = __builtin_amdgcn_s_memtime();   // (shader clock cycles: life in cycles / life in 10 ns = the clock the wave ran at)
#endif
#endif

  // view handed to the helpers of rt_kernels.h

  // ---- registers that live across the traversal: the ray being traced and a few flags -----------------
  RtVec3 tr_e, tr_d;        // ray being traced
  float tr_stop = -1.f;     // >= 0: shadow ray, length to the light
  int phase;
  constexpr bool wavefront = BOUNCE && QUEUES;                // bounce rays leave through the queue
  // every ray through the assembly walk, class by class (trace_class)?  The variants with bounce rays on the lanes' stacks keep
  // the general walk for their mixed waves (measured with all variants class by class, before the queues had launches of
  // their own: glass bunny 1920x1080 3.23 -> 3.55 ms, 4K and the mirror scenes within 1 %); the launches that render a
  // queue go class by class - their rays are ordered by direction class, so few waves are mixed (glass bunny 512x512
  // 1.94 -> 1.87 ms, 1920x1080 2.83 -> 2.77, 4K unchanged).
#ifdef RT_NO_ASM_WALK
  constexpr bool CLASSES = false;
#else
  constexpr bool CLASSES = (!BOUNCE || wavefront) && !COUNT && !RT_PATHS_ON;
#endif
  constexpr bool queued = wavefront && !QPRIMARY;   // this launch renders a queue of bounce rays, one per lane
  // Measurement build (tools/build_variant.sh lanes -DRT_LANE_WALK): the rays of a queue walked lane by lane (trace_lanes)
  // instead of as a wave.  Same bytes; slower: the generations of the glass bunny at 4K take 1.8-2.0 ms each against
  // 1.2-1.4 ms (frame 6.9 against 5.1 ms; 512x512 2.3 against 2.0 ms) - sorted, the rays of a wave still share most of their walk.
#ifdef RT_LANE_WALK
  constexpr bool LANES = queued && !COUNT && !RT_PATHS_ON;
#else
  constexpr bool LANES = false;
#endif
  if (!queued) {
    const Pixel px0 = pixel_of(TID());
    phase = px0.valid ? PH_CLOSEST : PH_DONE;
    const int W = A.cam.width, H = A.cam.height;
    const int i = px0.col, j = H - 1 - px0.row;   // camera.cu:94,118: j counts from the bottom
    // camera.cu:96-99; (i+0.5) is a double, so product and sum are double before the store to float
    const float ic = (float)((double)c.left + (double)((c.right - c.left) / (float)(unsigned short)W) * ((double)i + 0.5));
    const float jc = (float)((double)c.bottom + (double)((c.top - c.bottom) / (float)(unsigned short)H) * ((double)j + 0.5));
    const float kz = -RT_NEAR;
    const RtVec3 d0 = {ic * c.u[0] + jc * c.v[0] + kz * c.w[0], ic * c.u[1] + jc * c.v[1] + kz * c.w[1], ic * c.u[2] + jc * c.v[2] + kz * c.w[2]};
    tr_d = rt_normalized(d0);
    tr_e = {c.pos[0], c.pos[1], c.pos[2]};
  }
  bool primary = !queued;
  int verdict = -1;   // result of the last shadow ray, until consumed

  // ---- shading state (parked during traversal) and the pixel's ray stack (scratch, BOUNCE only) --------
  ShadeState<BOUNCE> sh;
  sh.P = {0, 0, 0}; sh.N = {0, 0, 1}; sh.view = tr_d;
  sh.colr = {0.f, 0.f, 0.f}; sh.color = {0.f, 0.f, 0.f}; sh.level1 = 0.f; sh.level2 = 0.f;
  sh.hit_t = RT_INF; sh.mfirst = 0; sh.mcount = 0; sh.mi = 0; sh.li = 0; sh.vis_known = 0; sh.vis_occluded = 0; sh.tu = 0.f; sh.tv = 0.f;
  sh.att = {1.f, 1.f, 1.f}; sh.level0 = c.scene_refr; sh.nlevel = 0; sh.depth = 0;
  BounceRay rstack[BOUNCE ? RT_RAY_STACK : 1];
  int rsp = 0;
  // queue mode: the term block of the lane's pixel, the ray's node in the pixel's ray tree, how many rays it has pushed
  uint32_t q_block = 0u, q_slot = 0u; int q_pushes = 0;
  // A ray pushed by the hit being shaded (ray.cuh:22-27): onto the lane's stack, or into the queue of the next generation.
  auto push_ray = [&](const RtVec3& eye, const RtVec3& dir, const RtVec3& att, int nlevel) {
    if (BOUNCE && wavefront) {
      const RtQueues wf = LATE_STRUCT(wf);
      // the lanes active here count themselves; the first of them reserves a run for all (one atomic per wave)
      const uint64_t m = __builtin_amdgcn_ballot_w64(true);
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (!queued && q_pushes == 0) {   // a primary ray's first push: its pixel gets a block of term slots
        const Pixel px = pixel_of(TID());
        const uint64_t mb = __builtin_amdgcn_ballot_w64(true);
        const uint32_t rb = __builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u));
        uint32_t bb = 0u;
        if (rb == 0u) bb = atomicAdd(&wf.state->n_blocks, (uint32_t)__builtin_popcountll(mb));
        bb = __builtin_amdgcn_readfirstlane(bb);
        q_block = bb + rb; q_slot = 0u;
        if (q_block < wf.block_cap) wf.block_pixel[q_block] = (uint32_t)((size_t)px.orow * LATE(cam).width + px.col);
        else wf.state->overflow = 1u;   // (the host then renders the frame the other way)
      }
      uint32_t base = 0u;
      if (rank == 0u) base = atomicAdd(&wf.state->count[wf.out_index], (uint32_t)__builtin_popcountll(m));
      base = __builtin_amdgcn_readfirstlane(base);
      const uint32_t at = base + rank;
      if (at < wf.queue_cap) {
        RtQRay& q = wf.out[at];
        q.e[0] = eye.x; q.e[1] = eye.y; q.e[2] = eye.z; q.d[0] = dir.x; q.d[1] = dir.y; q.d[2] = dir.z;
        q.att[0] = att.x; q.att[1] = att.y; q.att[2] = att.z;
        q.level[0] = sh.level0; q.level[1] = sh.level1; q.level[2] = sh.level2; q.nlevel = nlevel; q.depth = sh.depth + 1;
        q.block = q_block; q.slot = q_slot * wf.pushes + 1u + (uint32_t)q_pushes;
      } else wf.state->overflow = 1u;
      q_pushes++;
    } else if (BOUNCE) {
      if (rsp < RT_RAY_STACK) {
        BounceRay& r = rstack[rsp++];
        r.eye = eye; r.dir = dir; r.att = att;
        r.level[0] = sh.level0; r.level[1] = sh.level1; r.level[2] = sh.level2;
        r.nlevel = nlevel; r.depth = sh.depth + 1;
      } else atomicAdd(&counters_ptr()[C_UB_STACK], 1ull);
    }
  };

  // A launch that renders a queue: one chunk of the queue per wave.  How long the queue is, is only known on the device; the
  // launch has a wave for every chunk of a queue as long as the host expects at most (a full queue on a context's first frame,
  // the last frame's longest generation and a margin afterwards; a frame whose queue turns out longer is rendered again), and a
  // wave whose chunk lies beyond the queue's end leaves at once.
  // (Waves that stay and take chunk after chunk by ticket were measured too: the glass bunny at 4K took 7.6 ms that way
  // against 5.3 ms, at 1920x1080 3.8 against 3.2 ms.)
  if (queued) {
    const RtQueues wf = LATE_STRUCT(wf);
    uint32_t n_in = wf.state->count[wf.in_index];
    n_in = n_in < wf.queue_cap ? n_in : wf.queue_cap;
    n_in = __builtin_amdgcn_readfirstlane(n_in);
    // a short queue: shorter walks in four times as many waves (measured break-even)
    const uint32_t per_wave = n_in < wf.small_below ? 16u : 64u;
    const uint32_t chunk = blockIdx.x * (RT_WBLOCK / 64) + (uint32_t)(TID() >> 6);
    if ((unsigned long long)chunk * per_wave >= n_in) return;
    const uint32_t gid = chunk * per_wave + LANE();
    phase = PH_DONE; tr_e = {0.f, 0.f, 0.f}; tr_d = {0.f, 0.f, 1.f};
    if ((uint32_t)LANE() < per_wave && gid < n_in) {
      const RtQRay& q = wf.in[(wf.order && n_in >= RT_QUEUE_SORT_MIN) ? wf.order[gid] : gid];
      tr_e = ld3(q.e); tr_d = ld3(q.d); sh.att = ld3(q.att);
      sh.level0 = q.level[0]; sh.level1 = q.level[1]; sh.level2 = q.level[2]; sh.nlevel = q.nlevel; sh.depth = q.depth;
      q_block = q.block; q_slot = q.slot;
      phase = PH_CLOSEST; count(N_RAYS);
    }
    sh.view = tr_d;
  }
#ifdef RT_WAVE_TIMES_CLK
  unsigned long long wt_walk = 0ull; uint32_t wt_rounds = 0u;   // cycles inside the walks, rounds of the loop
#endif
  while (__builtin_amdgcn_ballot_w64(phase != PH_DONE)) {
    // ---- trace the pending rays of the wave ---------------------------------------------------------
    Hit h = {RT_INF, -1, 0.f, 0.f};
#ifdef RT_WAVE_TIMES_CLK
    const unsigned long long wt_w0 = __builtin_amdgcn_s_memtime(); wt_rounds++;
#endif
    bool traced = phase != PH_DONE;   // (trace_class may leave some of the wave's rays for the next round)
    {
      const bool want = phase != PH_DONE;
      // the instrumented variant walks shadow rays to the end, as the reference does, so that its test counts match
      const float stop_t = (phase == PH_SHADOW && !COUNT) ? tr_stop : -1.f;
      if constexpr (!MESH) {
        // A scene without a mesh: the closest hit is the loop over the spheres (ray.cu:52-53) and nothing else.  No walk needs
        // the registers, so the shading state stays where it is instead of going to LDS and back around every ray
        // (36 words each way for the variants with bounce rays: a third of what a round of a sphere scene issued).
        if ((COUNT || RT_PATHS_ON) && __builtin_amdgcn_ballot_w64(want)) tl.w[0]++;
        bool mine_live = want;
        for (int32_t s = 0; s < n_spheres; s++) {
          if (mine_live) {
            if (COUNT) tl.sph++;
            if (rt_sphere_test(ld3(spheres[s].c), spheres[s].r, tr_e, tr_d, h.t)) {
              h.prim = -2 - s;
              if (stop_t >= 0.f && h.t < stop_t) mine_live = false;
            }
          }
        }
      } else {
      park<BOUNCE, TEX, QPRIMARY>(sh, MY_PARK());
      asm volatile("" ::: "memory");
      if (LANES) trace_lanes(nodes, LATE(node_skip), tri_leaf, spheres, n_spheres, has_oct != 0, want, tr_e, tr_d, stop_t, h);
      else if (CLASSES) {
        float* const bary = parked + (SW + 4) * RT_WBLOCK + (RT_WBLOCK == 64 ? 0 : (TID() & ~63));   // the wave's columns of the two barycentric words
        traced = in_mask(trace_class<!queued>(nodes, n_wnodes, tri_leaf, spheres, n_spheres, has_oct != 0, want, tr_e, tr_d, stop_t, h, st, bary, LATE(cull_offset), LATE(cull_r0_max), LATE(cull_ksp_max)));
        asm volatile("" ::: "memory");
        h.bet = bary[LANE()]; h.gam = bary[RT_WBLOCK + LANE()];   // (only meaningful, and only used, when a triangle was hit)
      }
      else trace_wave<COUNT, !queued && !COUNT>(nodes, n_wnodes, tri_leaf, spheres, n_spheres, has_oct != 0, want, tr_e, tr_d, stop_t, h, st, LANE(), tl, cull_sh + 16 * (TID() >> 6), COUNT ? 0u : LATE(cull_offset), LATE(cull_r0_max), LATE(cull_ksp_max));
      asm volatile("" ::: "memory");
      unpark<BOUNCE, TEX, QPRIMARY>(sh, MY_PARK());
      }
    }
#ifdef RT_WAVE_TIMES_CLK
    wt_walk += __builtin_amdgcn_s_memtime() - wt_w0;
#endif
    if (traced) count(N_TRACED);
    const rt_kernarg_ptr ka = late_args();   // one opaque copy for the whole shading phase: its loads go out together
    if (!traced) {
    } else if (phase == PH_CLOSEST) {
      if (primary) {
        primary = false;
        int32_t* const ids = LATE_AT(ka, ids); float* const ts = LATE_AT(ka, ts);
        if (ids || ts) {
          const Pixel px = pixel_of(TID());
          const size_t p = (size_t)px.orow * LATE(cam).width + px.col;
          if (ids) ids[p] = h.prim;
#ifndef RT_WAVE_TIMES
          if (ts) ts[p] = h.t;
#endif
        }
      }
      sh.colr = {0.f, 0.f, 0.f};
      sh.mi = 0; sh.li = 0; sh.vis_known = 0; sh.vis_occluded = 0; sh.hit_t = h.t; sh.view = tr_d;
      if (h.prim == -1) {
        sh.colr = {LATE_AT(ka, cam).bg[0], LATE_AT(ka, cam).bg[1], LATE_AT(ka, cam).bg[2]};   // ray.cu:86
        sh.mcount = 0;
      } else {
        RtDeviceView S;   // view handed to the helpers of rt_kernels.h
        S.tri_geom = LATE_AT(ka, tri_geom); S.tri_shade = LATE_AT(ka, tri_shade); S.spheres = spheres;
        surface(S, tr_e, tr_d, h, sh.P, sh.N);
        if (TEX) hit_uv(LATE_AT(ka, tri_uv), h, sh.N, sh.tu, sh.tv);
        if (h.prim >= 0) {
          const int32_t mesh = (int32_t)__float_as_uint(S.tri_geom[3 * h.prim + 2].y);
          sh.mfirst = LATE_AT(ka, mesh_mat_first)[mesh]; sh.mcount = LATE_AT(ka, mesh_mat_count)[mesh];
        } else { sh.mfirst = spheres[-2 - h.prim].mat_first; sh.mcount = spheres[-2 - h.prim].mat_count; }
      }
    } else if (phase == PH_SHADOW) {
      verdict = tr_stop > h.t ? 1 : 0;   // ray.cu:70; consumed by the (material, light) step that asked for it
    }
    // ---- shade until the lane needs another ray or its pixel is finished --------------------------------
    if (traced) {
      phase = PH_SHADE;
      while (phase == PH_SHADE) {
        if (sh.mi >= sh.mcount) {   // every material of the hit handled: camera.cu:110-111, then next ray of the pixel
          if (BOUNCE && wavefront && (queued || q_pushes > 0)) {
            // the ray's contribution goes to its slot of the pixel's block; the pixel is summed up from there, in the reference's order
            const RtQueues wf = LATE_STRUCT(wf);
            if (q_block < wf.block_cap)
              wf.terms[(size_t)q_slot * wf.block_cap + q_block] = RtTermRGB{sh.att.x * sh.colr.x, sh.att.y * sh.colr.y, sh.att.z * sh.colr.z, wf.tag};
            phase = PH_DONE;
            break;
          }
          if (!BOUNCE || QPRIMARY) {   // the pixel's only ray: its bytes wait in LDS for the end of the wave
            const float cx = 0.f + sh.att.x * sh.colr.x, cy = 0.f + sh.att.y * sh.colr.y, cz = 0.f + sh.att.z * sh.colr.z;
            MY_COUNTS()[3 * RT_WBLOCK] = (uint32_t)rt_quantise(cx) | ((uint32_t)rt_quantise(cy) << 8) | ((uint32_t)rt_quantise(cz) << 16);
            phase = PH_DONE;
            break;
          }
          sh.color.x = sh.color.x + sh.att.x * sh.colr.x;
          sh.color.y = sh.color.y + sh.att.y * sh.colr.y;
          sh.color.z = sh.color.z + sh.att.z * sh.colr.z;
          if (BOUNCE && rsp > 0) {
            const BounceRay& r = rstack[--rsp];
            tr_e = r.eye; tr_d = r.dir; sh.att = r.att;
            sh.level0 = r.level[0]; sh.level1 = r.level[1]; sh.level2 = r.level[2];
            sh.nlevel = r.nlevel; sh.depth = r.depth;
            phase = PH_CLOSEST; count(N_RAYS);
          } else phase = PH_DONE;
          break;
        }
        RtMaterial M = LATE_AT(ka, materials)[LATE_AT(ka, mat_index)[sh.mfirst + sh.mi]];
        bool seen_by_ambient = true;
        if (TEX && M.type >= RT_TEX_LAMBERT) {   // a texture material: Lambert / Blinn-Phong with the colour looked up (rt_kernels.h)
          const uint32_t tflags = __float_as_uint(M.att[1]);
          const RtVec3 tc = texture_colour(LATE_AT(ka, textures), LATE_AT(ka, texels), __float_as_uint(M.att[0]), tflags, sh.tu, sh.tv);
          M.col[0] = tc.x; M.col[1] = tc.y; M.col[2] = tc.z;
          M.type -= RT_TEX_LAMBERT; seen_by_ambient = (tflags & RT_TEX_AMBIENT) != 0u;
        }
        const RtVec3 P = sh.P, N = sh.N;
        if (M.type == RT_LAMBERT || M.type == RT_PHONG) {   // material.cu:15-24, lights.cu:260-367
          const int li = sh.li;
          if (li >= LATE_AT(ka, n_lights)) { sh.mi++; sh.li = 0; continue; }
          const RtLight L = LATE_AT(ka, lights)[li];
          if (L.type == RT_AMBIENT) {
            if (M.type == RT_LAMBERT && seen_by_ambient) {
              sh.colr.x = sh.colr.x + M.col[0] * L.col[0]; sh.colr.y = sh.colr.y + M.col[1] * L.col[1]; sh.colr.z = sh.colr.z + M.col[2] * L.col[2];
            }
            sh.li++; continue;
          }
          RtVec3 l, sP; float ldist;
          if (MESH && verdict >= 0) {
            // the step comes round again with the verdict of the trace it asked for: the direction to the light is the ray's,
            // worked out then (a root and three divisions for a point light), and its origin and length have served
            l = tr_d; sP = tr_e; ldist = tr_stop;
          } else {
            if (L.type == RT_DIRECTIONAL) { l = {L.dir[0] * -1, L.dir[1] * -1, L.dir[2] * -1}; ldist = RT_INF; }
            else l = rt_normalized({L.pos[0] - P.x, L.pos[1] - P.y, L.pos[2] - P.z});
            sP = {P.x + l.x * RT_EPS, P.y + l.y * RT_EPS, P.z + l.z * RT_EPS};
            if (L.type != RT_DIRECTIONAL) ldist = rt_len({L.pos[0] - sP.x, L.pos[1] - sP.y, L.pos[2] - sP.z});
          }
          if (LATE_AT(ka, cam).shadows) {   // ray.cu:56-72
            bool occluded;
            if (verdict >= 0) {   // the trace requested by this very step has come back
              occluded = verdict != 0; verdict = -1;
              if (!COUNT && li < 32) { sh.vis_known |= 1u << li; if (occluded) sh.vis_occluded |= 1u << li; }
            } else if (!COUNT && li < 32 && ((sh.vis_known >> li) & 1u)) {
              // origin, direction and length depend on the hit point and the light only, so every further
              // material of this hit gets the verdict of the ray already traced (the instrumented variant
              // re-traces, to count what the reference does)
              occluded = (sh.vis_occluded >> li) & 1u;
            } else if constexpr (!MESH) {
              // no mesh: the shadow ray is the loop over the spheres, traced on the spot (ray.cu:56-72) instead of in a round of
              // its own (which would work the light's direction and distance out a second time when the verdict comes back)
              float st = RT_INF;
              bool looking = true;
              for (int32_t sp = 0; sp < n_spheres; sp++) {
                if (looking) {
                  if (COUNT) tl.sph++;
                  if (rt_sphere_test(ld3(spheres[sp].c), spheres[sp].r, sP, l, st) && !COUNT && st < ldist) looking = false;   // (the instrumented variant tests them all, as the reference does)
                }
              }
              occluded = ldist > st;   // ray.cu:70
              count(N_TRACED);
              if (!COUNT && li < 32) { sh.vis_known |= 1u << li; if (occluded) sh.vis_occluded |= 1u << li; }
            } else {
              tr_e = sP; tr_d = l; tr_stop = ldist; phase = PH_SHADOW;
              break;
            }
            count(N_RAYS); count(N_SHADOW);   // the reference casts this ray for every (material, light) pair
            if (occluded) { sh.li++; continue; }
          }
          if (L.type == RT_SPOT && !(-rt_dot(l, ld3(L.dir)) > L.cos_half)) { sh.li++; continue; }
          float w;
          if (M.type == RT_LAMBERT) { const float nl = rt_dot(N, l); w = nl > 0.f ? nl : 0.f; }
          else {
            const RtVec3 hv = rt_normalized({sh.view.x * -1 + l.x, sh.view.y * -1 + l.y, sh.view.z * -1 + l.z});
            const float nh = rt_dot(N, hv);
            w = powf(nh > 0.f ? nh : 0.f, M.param);
          }
          sh.colr.x = sh.colr.x + (M.col[0] * L.col[0]) * w;
          sh.colr.y = sh.colr.y + (M.col[1] * L.col[1]) * w;
          sh.colr.z = sh.colr.z + (M.col[2] * L.col[2]) * w;
          sh.li++;
        } else {
          if (BOUNCE && sh.depth < LATE_AT(ka, cam).max_depth) {
            const RtVec3 vd = sh.view;
            const float s2 = 2 * rt_dot(vd, N);
            const RtVec3 rl = rt_normalized({vd.x - N.x * s2, vd.y - N.y * s2, vd.z - N.z * s2});
            if (M.type == RT_REFLECTIVE) {   // material.cu:25-35
              push_ray({P.x + rl.x * RT_EPS, P.y + rl.y * RT_EPS, P.z + rl.z * RT_EPS}, rl,
                       {M.col[0] * sh.att.x, M.col[1] * sh.att.y, M.col[2] * sh.att.z}, sh.nlevel);
            } else {   // RT_DIELECTRIC, material.cu:36-99
              const float refr = sh.level_at(sh.nlevel);
              const RtVec3 rlsP = {P.x + N.x * RT_EPS, P.y + N.y * RT_EPS, P.z + N.z * RT_EPS};
              const float q = (refr - M.param) / (refr + M.param);
              const float R0 = q * q;
              const float dn = rt_dot(vd, N);
              const bool entering = dn < 0;
              float rOrI, cs; RtVec3 nn; float K[3];
              if (entering) { rOrI = refr / M.param; cs = -dn; nn = N; K[0] = K[1] = K[2] = 1.0f; }
              else {
                rOrI = M.param / refr; cs = dn; nn = {N.x * -1, N.y * -1, N.z * -1};
                for (int k = 0; k < 3; k++) K[k] = expf(M.att[k] * -sh.hit_t);
              }
              const float totRefl = 1 - (rOrI * rOrI) * (1 - cs * cs);
              const float a3[3] = {sh.att.x, sh.att.y, sh.att.z};
              float attl[3], attf[3];
              for (int k = 0; k < 3; k++) { attl[k] = (M.col[k] * K[k]) * a3[k]; attf[k] = attl[k]; }
              if (totRefl >= 0) {
                const RtVec3 rf = refract_dir(vd, nn, rOrI);
                const float fresnel = (float)((double)R0 + (double)(1 - R0) * pow(1.0 - (double)cs, 5.0));
                for (int k = 0; k < 3; k++) attf[k] = attf[k] * (1 - fresnel);
                int nl;
                if (entering) {
                  nl = sh.nlevel + 1;
                  if (nl > 2) { nl = 2; atomicAdd(&counters_ptr()[C_UB_OVER], 1ull); }
                  sh.set_level(nl, M.param);
                } else {
                  nl = sh.nlevel - 1;
                  if (nl < 0) { nl = 0; atomicAdd(&counters_ptr()[C_UB_LEAVE], 1ull); }
                }
                push_ray({P.x + rf.x * RT_EPS, P.y + rf.y * RT_EPS, P.z + rf.z * RT_EPS}, rf, {attf[0], attf[1], attf[2]}, nl);
                for (int k = 0; k < 3; k++) attl[k] = attl[k] * fresnel;
              }
              push_ray(rlsP, rl, {attl[0], attl[1], attl[2]}, sh.nlevel);
            }
          }
          sh.mi++; sh.li = 0;
        }
      }
    }
  }

  // ---- quantise (camera.cu:113-122; NaN < 1 is false, so NaN -> 1 -> 255) and store -------------------------
  // A tile row is 24 contiguous bytes.  The wave packs its 192 bytes through LDS and stores them as 48 aligned
  // dwords instead of 192 single bytes (each byte store is its own partial write to memory).
  // (Pixels whose rays went through the queues are written again, with their sum, by the resolve kernel.)
  if (!queued) {
    uint8_t q[3];
    if (BOUNCE && !QPRIMARY) { q[0] = rt_quantise(sh.color.x); q[1] = rt_quantise(sh.color.y); q[2] = rt_quantise(sh.color.z); }
    else {
      asm volatile("" ::: "memory");
      const uint32_t packed = MY_COUNTS()[3 * RT_WBLOCK];   // (0 for a lane without a pixel: the bytes of colour 0)
      q[0] = (uint8_t)packed; q[1] = (uint8_t)(packed >> 8); q[2] = (uint8_t)(packed >> 16);
    }
    const Pixel pxe = pixel_of(TID());
    const RowMap rows = LATE_STRUCT(rows);
    const int W = LATE(cam).width, H = LATE(cam).height;
    uint8_t* const rgb = LATE(rgb);
    int wave = TID(); asm volatile("" : "+v"(wave)); wave >>= 6;
    if (pxe.valid) count(N_RAYS);   // the primary ray
    const int col0 = pxe.col & ~7, lrow0 = pxe.lrow & ~7;   // tile origin (wave-uniform)
    const int first_row = band_row(rows, lrow0), last_row = band_row(rows, lrow0 + 7);
    const bool whole = !rows.small && (W & 3) == 0 && col0 + 8 <= W && lrow0 + 8 <= rows.n_local && last_row < H &&
                       last_row == first_row + 7;   // (the tile's rows lie in one band; wave-uniform)
    if (whole) {
      uint8_t* pk = (uint8_t*)pack[wave];
      pk[LANE() * 3 + 0] = q[0]; pk[LANE() * 3 + 1] = q[1]; pk[LANE() * 3 + 2] = q[2];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (LANE() < 48) {
        const int r = LANE() / 6, j = LANE() - 6 * r;
        const int orow0 = rows.compact ? lrow0 : first_row;
        uint32_t* dst = (uint32_t*)(rgb + ((size_t)(orow0 + r) * W + col0) * 3) + j;
        *dst = pack[wave][r * 6 + j];
      }
    } else if (pxe.valid) {
      const size_t p = ((size_t)pxe.orow * W + pxe.col) * 3;
      rgb[p] = q[0]; rgb[p + 1] = q[1]; rgb[p + 2] = q[2];
    }
  }
#ifdef RT_WAVE_TIMES
  float* const ts = LATE(ts);
  // (the waves of the queue launches log behind the tiles' records: a counter word, then four words per wave in the order they ended)
  uint32_t wt_sum = 0u, wt_max = 0u;
  if (COUNT && ts && queued) {
    wt_sum = tl.node + tl.tri; wt_max = wt_sum;
    for (int off = 32; off > 0; off >>= 1) { wt_sum += __shfl_xor(wt_sum, off, 64); const uint32_t o2 = __shfl_xor(wt_max, off, 64); wt_max = o2 > wt_max ? o2 : wt_max; }
  }
  if (ts && queued && LANE() == 0) {
    const unsigned long long wt_end = __builtin_amdgcn_s_memrealtime();
    uint32_t* const log = (uint32_t*)ts + 4 * (size_t)(((LATE(cam).width + 7) / 8) * ((LATE(cam).height + 7) / 8));
    const uint32_t at = atomicAdd(log, 1u);
    if (4ull * (at + 2u) + 4ull * (((LATE(cam).width + 7) / 8) * ((LATE(cam).height + 7) / 8)) <= (unsigned long long)LATE(cam).width * LATE(cam).height) {
      uint32_t* o = log + 4 + 4 * (size_t)at;
      o[0] = (uint32_t)wt_start; o[1] = (uint32_t)wt_end; o[2] = (uint32_t)sh.depth; o[3] = blockIdx.x;
      if (COUNT) { o[2] |= wt_sum << 2; o[3] = wt_max; }   // (with the counting variant: the wave's node + triangle tests, and those of its busiest lane)
    }
  }
  if (ts && !queued && blockIdx.x == 0 && TID() == 0) ((uint32_t*)ts)[4 * (size_t)(((LATE(cam).width + 7) / 8) * ((LATE(cam).height + 7) / 8))] = 0u;
  if (ts && !queued && LANE() == 0) {
    const unsigned long long wt_end = __builtin_amdgcn_s_memrealtime();
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    uint32_t* o = (uint32_t*)ts + 4 * (size_t)((tile_of().y * LATE(tiles_x) + tile_of().x) * (RT_WBLOCK / 64) + (TID() >> 6));
    o[0] = (uint32_t)wt_start; o[1] = (uint32_t)wt_end; o[2] = hw; o[3] = xcc;
#ifdef RT_WAVE_TIMES_CLK
    // (-DRT_WAVE_TIMES_CLK, tools/exp_lone_wave_parts.py: the wave's life in shader clock cycles; sixteenths of the cycles it spent inside walks and the rounds of its main loop)
    o[2] = (uint32_t)(__builtin_amdgcn_s_memtime() - wt_clk0);
    o[3] = (uint32_t)(wt_walk >> 4) | (wt_rounds << 24);
#endif
  }
#endif

  // work counters: wave reduction, one atomic per wave and counter
  if (!LATE(tally)) return;   // nobody asked for statistics
  unsigned long long* const counters = counters_ptr();
  asm volatile("" ::: "memory");
  const uint32_t vals[7] = {MY_COUNTS()[0], MY_COUNTS()[RT_WBLOCK], tl.node, tl.leaf, tl.tri, tl.sph, MY_COUNTS()[2 * RT_WBLOCK]};
  const int slot[7] = {C_RAYS, C_SHADOW, C_NODE, C_LEAF, C_TRI, C_SPH, C_TRACED};
  for (int k = 0; k < 7; k++) {
    if (!COUNT && k >= 2 && k <= 5) continue;
    uint32_t v = vals[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (LANE() == 0 && v) atomicAdd(&counters[slot[k]], (unsigned long long)v);
  }
  if ((COUNT || RT_PATHS_ON) && LANE() == 0) for (int k = 0; k < 8; k++) if (tl.w[k]) atomicAdd(&counters[C_W_TRACE + k], (unsigned long long)tl.w[k]);
}

#undef TID
#undef LANE
#undef MY_PARK
#undef MY_COUNTS

}  // namespace

int rt_preload_wave_kernels() {
  hipFuncAttributes attr;   // asking for a kernel's attributes makes the runtime load the code object it is in
  // (one code object holds every variant; the variants a frame of any kind launches are all asked for, so that none of them
  //  is looked up for the first time inside a render call)
  const void* const kernels[] = {
      reinterpret_cast<const void*>(&render_wave<false, false, false, true>), reinterpret_cast<const void*>(&render_wave<false, false, false>),
      reinterpret_cast<const void*>(&render_wave<false, true, false>), reinterpret_cast<const void*>(&render_wave<true, false, false>),
      reinterpret_cast<const void*>(&render_wave<true, true, false>), reinterpret_cast<const void*>(&render_wave<true, false, true>),
      reinterpret_cast<const void*>(&render_wave<true, true, true>), reinterpret_cast<const void*>(&render_wave<true, false, true, false, false, true>),
      reinterpret_cast<const void*>(&render_wave<true, true, true, false, false, true>), reinterpret_cast<const void*>(&render_wave<false, false, false, false, true>),
      reinterpret_cast<const void*>(&render_wave<false, true, false, false, true>), reinterpret_cast<const void*>(&render_wave<true, false, false, false, true>),
      reinterpret_cast<const void*>(&render_wave<true, true, false, false, true>),
      reinterpret_cast<const void*>(&render_wave<false, false, false, false, false, false, false>), reinterpret_cast<const void*>(&render_wave<false, true, false, false, false, false, false>),
      reinterpret_cast<const void*>(&render_wave<true, false, false, false, false, false, false>), reinterpret_cast<const void*>(&render_wave<true, true, false, false, false, false, false>)};
  for (const void* k : kernels) { const hipError_t e = hipFuncGetAttributes(&attr, k); if (e != hipSuccess) return (int)e; }
  return (int)hipSuccess;
}

void rt_launch_wave_kernel(const RtDeviceView& V, const RowMap& rows, bool bounce, bool count, uint8_t* rgb, int32_t* ids,
                           float* ts, unsigned long long* counters, bool tally, const RtQueues& wf, hipStream_t stream) {
  const int tw = rows.small ? RT_WTILE_W / 2 : RT_WTILE_W, th = rows.small ? RT_WTILE_H / 2 : RT_WTILE_H;
  dim3 grid((V.cam.width + tw - 1) / tw, (rows.n_local + th - 1) / th), block(RT_WBLOCK);
  if (wf.in) {
    // one chunk per wave, for as many chunks as a queue of wf.launch_rays rays has (of a short queue, cut into chunks of 16 rays, only
    // RT_QUEUE_SMALL / 16 exist); a wave whose chunk lies beyond the queue's end leaves at once
    const uint32_t cover = wf.launch_rays;
    const uint32_t chunks = cover < wf.small_below ? (cover + 15u) / 16u : std::max(wf.small_below / 16u, (cover + 63u) / 64u);
    grid = dim3((chunks + RT_WBLOCK / 64 - 1) / (RT_WBLOCK / 64), 1); bounce = true;
  }
  RtWaveArgs A;
  A.tiles_x = grid.x; A.tiles_y = grid.y; A.strips_per_die = 0u;
  // n / d as the high word of n * ceil(2^32 / d): exact for every n <= n_max when n_max * (ceil(2^32 / d) * d - 2^32) < 2^32
  auto magic_for = [](uint64_t d, uint64_t n_max) -> uint32_t {
    if (d < 2) return 0u;
    const uint64_t m = ((1ull << 32) + d - 1) / d, e = m * d - (1ull << 32);
    return (m >> 32) == 0 && n_max * e < (1ull << 32) ? (uint32_t)m : 0u;
  };
  A.band_magic = magic_for((uint64_t)rows.band_rows, (uint64_t)rows.n_local + 64u); A.strip_magic = 0u;
  static const bool strips_in_order = getenv("RT_AMD_STRIPS_IN_ORDER") != nullptr;   // measurements
  A.centre_first = strips_in_order ? 0u : 1u;
  static const bool plain_grid = getenv("RT_AMD_PLAIN_GRID") != nullptr;   // measurements
  static const uint64_t strip_min_tiles = getenv("RT_AMD_STRIP_MIN_TILES") ? strtoull(getenv("RT_AMD_STRIP_MIN_TILES"), nullptr, 10) : 4000u;
  // (measured with the threshold lowered: meshDragon 1920x1080 1.20 -> 1.27 ms, 2560x1440 1.55 -> 1.56 ms; the bands of one GPU
  // of 8 / 4 / 2 with two frames in flight 0.47 -> 0.52, 0.75 -> 0.74, 1.36 -> 1.35 ms: the sixth wave only pays on a full frame)
  static const uint64_t dense_min_tiles = getenv("RT_AMD_DENSE_MIN_TILES") ? strtoull(getenv("RT_AMD_DENSE_MIN_TILES"), nullptr, 10) : 60000u;
  // Tiles dealt to the dies in strips: a one-dimensional grid of 8 x strips x rows slots, for launches of 4 000 tiles or
  // more, or on request (rows.strips, RT_RENDER_STRIPS).  Since each die walks its strips from the middle outwards the
  // strips pay from about 1024x768 on, mostly through that order: a small frame ends when its few long waves end, and
  // those now start first (meshDragon 1024x768 1.23 -> 0.98 ms, 1600x900 1.18 -> 0.92, 1920x1080 1.22 -> 1.11, 2560x1440
  // 1.55 -> 1.35; meshTorus 1280x720 1.00 -> 0.90; lambertBPHSpheres 1920x1080 0.30 -> 0.27; meshMedBunny within 5 % either
  // way).  With the strips walked left to right they only paid from 60 000 tiles on.
  if (!wf.in && RT_WBLOCK == 64 && !plain_grid && ((uint64_t)grid.x * grid.y >= strip_min_tiles || rows.strips)) {
    const uint32_t strips = (grid.x + RT_STRIP_W - 1u) / RT_STRIP_W;
    A.strips_per_die = (strips + RT_DIES - 1u) / RT_DIES;
    A.strip_magic = magic_for((uint64_t)RT_STRIP_W * grid.y, (uint64_t)A.strips_per_die * RT_STRIP_W * grid.y + 1u);
    grid = dim3(RT_DIES * A.strips_per_die * RT_STRIP_W * grid.y, 1);
  }
  A.cam = V.cam; A.rows = rows; A.tri_geom = V.tri_geom; A.tri_shade = V.tri_shade;
  A.spheres = V.spheres; A.lights = V.lights; A.materials = V.materials; A.mat_index = V.mat_index;
  A.mesh_mat_first = V.mesh_mat_first; A.mesh_mat_count = V.mesh_mat_count;
  A.tri_uv = V.tri_uv; A.textures = V.textures; A.texels = V.texels; A.node_skip = V.node_skip;
  A.n_spheres = V.n_spheres; A.n_lights = V.n_lights; A.has_oct = V.has_oct; A.n_wnodes = V.n_wnodes;
  static const bool no_cull = getenv("RT_AMD_NO_CULL") != nullptr;
  // (small meshes: what a wave saves in their few leaves does not pay for working out its bundle - boxedSpheres, 12 triangles: + 16 %;
  //  rays from the queues: scattered by construction, their bundles rule nothing out)
  static const int cull_min_nodes = getenv("RT_AMD_CULL_MIN_NODES") ? atoi(getenv("RT_AMD_CULL_MIN_NODES")) : 512;
  A.cull_offset = (no_cull || rows.every_triangle || V.n_wnodes < cull_min_nodes || wf.in != nullptr) ? 0u : V.cull_offset; A.pad_ = 0u;
  static const float cull_r0 = getenv("RT_AMD_CULL_R0") ? (float)atof(getenv("RT_AMD_CULL_R0")) : 1.0f;      // measurements
  static const float cull_ksp = getenv("RT_AMD_CULL_KSP") ? (float)atof(getenv("RT_AMD_CULL_KSP")) : 0.25f;
  A.cull_r0_max = cull_r0; A.cull_ksp_max = cull_ksp;
  A.rgb = rgb; A.ids = ids; A.ts = ts; A.counters_all = counters; A.tally = tally ? 1 : 0; A.wf = wf;
#define LAUNCH(B, C, Q) hipLaunchKernelGGL((render_wave<B, C, Q>), grid, block, 0, stream, V.wnodes, V.tri_leaf, A)
  if (V.textures) {   // (never from queues: rt_device.hip keeps scenes with texture materials on the stacks)
#define LAUNCH_TEX(B, C) hipLaunchKernelGGL((render_wave<B, C, false, false, true>), grid, block, 0, stream, V.wnodes, V.tri_leaf, A)
    if (bounce && count) LAUNCH_TEX(true, true); else if (bounce) LAUNCH_TEX(true, false); else if (count) LAUNCH_TEX(false, true); else LAUNCH_TEX(false, false);
#undef LAUNCH_TEX
  }
  else if (wf.enabled && !wf.in) {   // the primary-ray pass of a frame rendered from queues
    if (count) hipLaunchKernelGGL((render_wave<true, true, true, false, false, true>), grid, block, 0, stream, V.wnodes, V.tri_leaf, A);
    else hipLaunchKernelGGL((render_wave<true, false, true, false, false, true>), grid, block, 0, stream, V.wnodes, V.tri_leaf, A);
  }
  else if (wf.enabled) { if (count) LAUNCH(true, true, true); else LAUNCH(true, false, true); }
  else if (!V.has_oct) {   // a scene of spheres only: the variants without a walk
#define LAUNCH_SPHERES(B, C) hipLaunchKernelGGL((render_wave<B, C, false, false, false, false, false>), grid, block, 0, stream, V.wnodes, V.tri_leaf, A)
    if (bounce && count) LAUNCH_SPHERES(true, true); else if (bounce) LAUNCH_SPHERES(true, false); else if (count) LAUNCH_SPHERES(false, true); else LAUNCH_SPHERES(false, false);
#undef LAUNCH_SPHERES
  }
  else if (bounce && count) LAUNCH(true, true, false);
  else if (bounce) LAUNCH(true, false, false);
  else if (count) LAUNCH(false, true, false);
  // (a caller that keeps several launches in flight, rows.strips, fills the chip between them: the sixth wave then pays from
  //  launches half as large - the bands of one GPU of 4, 32 400 tiles, three in flight: 0.333 -> 0.306 ms per frame; of one GPU
  //  of 8, 16 200 tiles: no gain)
  else if (((uint64_t)A.tiles_x * A.tiles_y >= (rows.strips ? dense_min_tiles * 2 / 5 : dense_min_tiles)) && V.has_oct && V.n_wnodes >= 512 && !rows.small)   // a big frame of a big mesh: one more wave per SIMD
    hipLaunchKernelGGL((render_wave<false, false, false, true>), grid, block, 0, stream, V.wnodes, V.tri_leaf, A);
  else LAUNCH(false, false, false);
#undef LAUNCH
}
