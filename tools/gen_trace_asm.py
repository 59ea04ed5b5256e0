#!/usr/bin/env python3
"""Writes raytracer_amd/csrc/rt_trace_asm.inc: the octree walk of a wave whose rays agree on their direction signs, as ONE
block of gfx950 assembly (scan of a branch's children, descent, ascent, leaf loop with the three numerator stages of the
triangle test).  The block is generated because the triangle stages exist twice (two sets of record registers filled in
turn) and every register is named explicitly; rt_kernel_wave.hip includes the result and documents the register map.

   python3 tools/gen_trace_asm.py        (from the repository root; the output is checked in)
"""
import os
import sys

# ---- scalar registers owned by the block (see RT_TRACE_* operand list in rt_kernel_wave.hip) ----
TA, TB = 36, 52                  # s[36:51], s[52:67]: two triangle records (RtTriLeaf, 16 words)
BLK = "s[68:69]"                 # address of the current branch's children
REM, FIRST = "s70", "s71"        # children still to test (bit 8 always set: "child 8" is the record behind the block), index of child 0
TMASK, MINE, LIVE = "s[72:73]", "s[74:75]", "s[76:77]"
IDX, RFIRST, RMETA = "s78", "s79", "s80"   # child whose record is in (or on its way into) NX; links of the child just tested
NODES = ("s82", "s83")
OFF = "s81"
NX = 84                          # s[84:91]: node record
UND, NR = "s[96:97]", "s[98:99]"  # (TRI and ACC: below, outside the range that the loop with four record sets takes over)
# The triangle records of a leaf are addressed by their byte offset from the start of the record array (a scene's records stay
# below 4 GiB: rt_device.hip checks): TCUR / TNXT take turns as "offset of the record under test" and "offset of the record
# behind it, which is on its way", TEND is the offset behind the leaf's last record.
TCUR, TNXT, TEND = "s20", "s21", "s22"
# Which of a leaf's triangles are still to be tested is a bit mask (SURV; bit 63 is always set while the leaf is open: "record
# 63", which no leaf has, marks the end, so that "the next one" can be requested without asking whether there is one).
# The mask starts as the triangles that the wave's bundle of rays can possibly be accepted by (cull(): lane k looks at
# triangle k of the leaf; rt_cull_test in rt_math.h), or as all of them when the caller passes no bundle (CULL = 0).
# Outside the block the mask waits in lanes 62 / 63 of the stack register ST_FIRST, whose lanes 32 .. 41 hold the bundle.
SURV, JT = "s[98:99]", "s22"       # (SURV shares the registers of NR, which only lives within the scan; in such a leaf TEND only carries the triangle count into the mask and is the scratch index JT afterwards)
CULL = "s19"                     # byte offset of the RtCullRec array from the triangle records, 0: test every triangle
CHG, T64, PASS = "s[24:25]", "s[26:27]", "s[28:29]"
RES, RPRIM = "s30", "s31"

# The leaf loop of a wave with a long life keeps FOUR records in flight (two being tested while the next two are on their way).  A scalar
# load takes a lone wave some 400 cycles and its test 150, and loads return out of order, so that the only wait there is
# waits for ALL of them: what hides the latency is the number of requests between two waits.  The two further record sets
# are s[68:83] and s[84:99]: the traversal state that lives there is parked in lanes 32 .. 44 of ST_CUR while a leaf with
# at least BIG triangles to test is open (smaller leaves keep the loops with two sets: parking costs what two tests cost).
# TRI and ACC move out of that range for good; inside that loop SURVB is the mask of triangles still to request (bit 63: the
# end marker), the J registers hold the leaf-relative index of the record in each set (63: none), RF = vcc_lo the leaf's first
# record index, bit 30: the walk left this leaf for the quotients at least once (what CHG says in the other loops).
BATCH = os.environ.get("RT_GEN_BATCH", "1") != "0"        # (RT_GEN_BATCH=0: the block without that loop, same registers)
TRI = ("s32", "s33")
ACC = "s[34:35]"
SURVB, SURVB_LO, SURVB_HI = "s[24:25]", "s24", "s25"     # (CHG's registers: "some candidate was met in this leaf" is bit 30 of RF there)
BIG = int(os.environ.get("RT_GEN_BATCH_MIN", "6"))       # leaves with at least so many triangles to test take the loop with four sets ...
WORK0 = int(os.environ.get("RT_GEN_BATCH_WORK", "64"))  # ... once the walk has met so many triangles in such leaves (WORK counts them): only
                                                         # a wave with a long life ahead waits for its records alone on its SIMD
J = ("s20", "s21", "s22", "s23")
RSET = (36, 52, 68, 84)
RF = "vcc_lo"
TMPP, TMP0, TMP1 = "s[30:31]", "s30", "s31"
PARKED = list(range(68, 79)) + [82, 83]

# ---- vector registers ----
E_XY, EZ = "v[40:41]", "v42"
C_ZX, C_YZ, C_YX = "v[44:45]", "v[46:47]", "v[48:49]"
C2X, C2Y, C2Z = "v45", "v46", "v44"
IX, IY, IZ, FCX, FCY, FCZ, CMAX, NEB, T, STOP = "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59"
ST_FIRST, ST_CUR, ST_LO, ST_HI = "v60", "v61", "v62", "v63"


WORK = "s23"                      # triangles the walk has met in leaves with BIG or more to test (an operand, zero when the walk starts; inside the
                                  # loop with four sets the register is J[3], and WORK0 again behind it)


def counting():
    """RT_GEN_COUNT=1: the block counts its steps in s14 .. s17 (measurement builds, -DRT_STAGE_COUNT); RT_GEN_COUNT=leaf: it counts
    instead how many triangles leaves hold that few rays of the wave enter (-DRT_STAGE_COUNT=2)."""
    return os.environ.get("RT_GEN_COUNT") not in (None, "", "0", "leaf")


def s(n):
    return "s%d" % n


def s2(n):
    return "s[%d:%d]" % (n, n + 1)


def scan_sorted():
    t = ["v%d" % (64 + i) for i in range(6)]
    tn, tf, span, tc, x, r = "v70", "v71", "v72", "v73", "v74", "v75"
    L = []
    # (scalar loads return out of order: before a record is requested into registers that an earlier request - the sibling
    #  behind the one that was accepted, the triangle behind a leaf's last - may still be on its way into, that one is awaited)
    L += ["LT_enter_%=:"] + (["s_add_u32 %s, %s, 1" % ("s15", "s15")] if counting() else []) + [
          "s_ff1_i32_b32 %s, %s" % (IDX, REM),
          "s_lshl_b32 %s, %s, 5" % (OFF, IDX),
          "s_waitcnt lgkmcnt(0)",
          "s_load_dwordx8 s[84:91], %s, %s" % (BLK, OFF),
          "LT_scan_%=:",
          "s_bitset0_b32 %s, %s" % (REM, IDX),
          "s_waitcnt lgkmcnt(0)"]
    if counting():
        L.append("s_add_u32 %s, %s, 0x10000" % ("s14", "s14"))
    for i, (inv, fc) in enumerate(((IX, FCX), (IX, FCX), (IY, FCY), (IY, FCY), (IZ, FCZ), (IZ, FCZ))):
        L.append("v_fma_f32 %s, %s, %s, -%s" % (t[i], s(NX + i), inv, fc))
    L += ["s_mov_b32 %s, %s" % (RFIRST, s(NX + 6)),
          "s_mov_b32 %s, %s" % (RMETA, s(NX + 7)),
          "s_ff1_i32_b32 %s, %s" % (IDX, REM),
          "s_lshl_b32 %s, %s, 5" % (OFF, IDX),
          "s_load_dwordx8 s[84:91], %s, %s" % (BLK, OFF),
          "v_max3_f32 %s, %s, %s, %s" % (tn, t[0], t[2], t[4]),
          "v_min3_f32 %s, %s, %s, %s" % (tf, t[1], t[3], t[5]),
          "v_cmp_lt_f32 vcc, 0, %s" % tn,
          "v_sub_f32 %s, %s, %s" % (span, tf, tn),
          "v_cndmask_b32 %s, %s, %s, vcc" % (tc, tf, tn),
          "v_sub_f32 %s, %s, %s" % (x, T, tc),
          "v_min3_f32 %s, %s, %s, %s" % (r, span, tf, x),
          "v_cmp_lt_f32_e64 %s, %s, %s" % (NR, r, NEB),
          "s_andn2_b64 %s, %s, %s" % (NR, MINE, NR),
          "s_cbranch_scc1 LT_tail_%=",
          "LT_next_%=:",
          "s_cmp_lt_u32 %s, 8" % IDX,
          "s_cbranch_scc1 LT_scan_%=",
          # ---- the branch is exhausted: back to the youngest entry of the stack that still has undecided rays
          "LT_pop_%=:",
          "s_cmp_eq_u32 m0, 0",
          "s_cbranch_scc1 LT_done_%=",
          "s_add_u32 m0, m0, -1", "@COUNT_POP",
          "s_nop 1",
          "v_readlane_b32 %s, %s, m0" % (FIRST, ST_FIRST),
          "v_readlane_b32 %s, %s, m0" % (REM, ST_CUR),
          "v_readlane_b32 s72, %s, m0" % ST_LO,
          "v_readlane_b32 s73, %s, m0" % ST_HI,
          "s_nop 0",
          "s_and_b64 %s, %s, %s" % (MINE, TMASK, LIVE),
          "s_cbranch_scc0 LT_pop_%=",
          "s_lshl_b32 %s, %s, 5" % (OFF, FIRST),
          "s_add_u32 s68, %s, %s" % (NODES[0], OFF),
          "s_addc_u32 s69, %s, 0" % NODES[1],
          "s_branch LT_enter_%=",
          # ---- some ray of the branch may hit the child: margin and acceptance (rt_node_slab_tail_fused)
          "LT_tail_%=:", "@COUNT_TAIL",
          "v_med3_f32 v76, %s, %s, %s" % (t[0], t[2], t[4]),
          "v_med3_f32 v77, %s, %s, %s" % (t[1], t[3], t[5]),
          "v_min3_f32 v78, %s, %s, %s" % (t[0], t[2], t[4]),
          "v_max3_f32 v79, %s, %s, %s" % (t[1], t[3], t[5]),
          "v_max3_f32 v78, v79, -v78, %s" % CMAX,
          "v_add_f32 v78, v78, %s" % T,
          "v_mul_f32 v78, 0x37000000, v78",
          "v_sub_f32 v76, %s, v76" % tn,
          "v_sub_f32 v77, v77, %s" % tf,
          "v_cndmask_b32 v76, v77, v76, vcc",
          "v_min3_f32 v76, %s, v76, %s" % (x, span),
          "v_min_f32 v76, v76, %s" % tc,
          "v_cmp_gt_f32_e64 %s, v76, v78" % ACC,
          "v_cmp_lt_f32_e64 %s, %s, -v78" % (UND, r),
          "s_and_b64 %s, %s, %s" % (ACC, ACC, MINE),
          "s_or_b64 %s, %s, %s" % (UND, UND, ACC),
          "s_andn2_b64 %s, %s, %s" % (UND, MINE, UND),
          "s_cbranch_scc1 LT_exit_exact_%=",
          "LT_after_exact_%=:",
          "s_cmp_eq_u64 %s, 0" % ACC,
          "s_cbranch_scc1 LT_next_%=",
          "s_bitcmp1_b32 %s, 31" % RMETA,
          "s_cbranch_scc0 LT_leaf_%=",
          # ---- a branch: its children become the current block; the parent is kept only if it has children left
          "s_and_b32 %s, %s, 0xff" % (OFF, RMETA),
          "s_cbranch_scc0 LT_next_%=",
          "s_cmp_lt_u32 %s, 8" % IDX,
          "s_cbranch_scc0 LT_nopush_%=",
          "v_writelane_b32 %s, %s, m0" % (ST_FIRST, FIRST),
          "v_writelane_b32 %s, %s, m0" % (ST_CUR, REM),
          "v_writelane_b32 %s, s72, m0" % ST_LO,
          "v_writelane_b32 %s, s73, m0" % ST_HI,
          "s_add_u32 m0, m0, 1",
          "LT_nopush_%=:",
          "s_mov_b32 %s, %s" % (FIRST, RFIRST),
          "s_or_b32 %s, %s, 0x100" % (REM, OFF),
          "s_mov_b64 %s, %s" % (TMASK, ACC),
          "s_mov_b64 %s, %s" % (MINE, ACC),
          "s_lshl_b32 %s, %s, 5" % (OFF, RFIRST),
          "s_add_u32 s68, %s, %s" % (NODES[0], OFF),
          "s_addc_u32 s69, %s, 0" % NODES[1],
          "s_branch LT_enter_%="]
    return L


def scan_zero(pattern):
    """Scan and classification for a wave whose rays share the zero pattern `pattern` of their directions (bit k: component k
    is exactly zero): rt_node_zero_outside as the quick rejection, rt_node_slab_zero_d as the full classification (rt_math.h).
    The records are the (lo, hi) variant; the non-zero components may differ in sign from lane to lane."""
    Z = [k for k in range(3) if (pattern >> k) & 1]
    NZ = [k for k in range(3) if not (pattern >> k) & 1]
    d = ["v%d" % (64 + i) for i in range(6)]      # lo - e, hi - e per axis
    ev = ("v40", "v41", EZ)
    inv = (IX, IY, IZ)
    side = "v74"
    mnmx = [("v70", "v71"), ("v72", "v73")]
    SQ = T64
    INALL = "s[98:99]"   # (NR is consumed before the tail needs this)
    L = ["LT_enter_%=:"] + (["s_add_u32 %s, %s, 1" % ("s15", "s15")] if counting() else []) + [
         "s_ff1_i32_b32 %s, %s" % (IDX, REM),
         "s_lshl_b32 %s, %s, 5" % (OFF, IDX),
         "s_waitcnt lgkmcnt(0)",
         "s_load_dwordx8 s[84:91], %s, %s" % (BLK, OFF),
         "LT_scan_%=:",
         "s_bitset0_b32 %s, %s" % (REM, IDX),
         "s_waitcnt lgkmcnt(0)"]
    if counting():
        L.append("s_add_u32 %s, %s, 0x10000" % ("s14", "s14"))
    for i in range(6):
        L.append("v_sub_f32 %s, %s, %s" % (d[i], s(NX + i), ev[i // 2]))
    L += ["v_mov_b32 %s, %s" % (side, s(NX)),
          "v_sub_f32 %s, %s, %s" % (side, s(NX + 1), side),
          "s_mov_b32 %s, %s" % (RFIRST, s(NX + 6)),
          "s_mov_b32 %s, %s" % (RMETA, s(NX + 7)),
          "s_ff1_i32_b32 %s, %s" % (IDX, REM),
          "s_lshl_b32 %s, %s, 5" % (OFF, IDX),
          "s_load_dwordx8 s[84:91], %s, %s" % (BLK, OFF)]
    for j, k in enumerate(Z):
        mn, mx = mnmx[j]
        L += ["v_min_f32_e64 %s, -%s, %s" % (mn, d[2 * k], d[2 * k + 1]),      # A = e - lo = -(lo - e), B = hi - e
              "v_max_f32_e64 %s, -%s, %s" % (mx, d[2 * k], d[2 * k + 1]),
              "v_cmp_gt_f32 vcc, 0, %s" % mn,
              "v_cmp_gt_f32_e64 %s, %s, %s" % (SQ, mx, side),
              "s_and_b64 %s, %s, vcc" % (SQ, SQ),
              ("s_mov_b64 %s, %s" if j == 0 else "s_or_b64 %s, %s, " + NR) % ((NR, SQ) if j == 0 else (NR, SQ))]
    L += ["s_andn2_b64 %s, %s, %s" % (NR, MINE, NR),        # lanes of the branch whose origin is not outside the child's extent on a zero axis
          "s_cbranch_scc1 LT_tail_%=",
          "LT_next_%=:",
          "s_cmp_lt_u32 %s, 8" % IDX,
          "s_cbranch_scc1 LT_scan_%=",
          "LT_pop_%=:",
          "s_cmp_eq_u32 m0, 0",
          "s_cbranch_scc1 LT_done_%=",
          "s_add_u32 m0, m0, -1", "@COUNT_POP",
          "s_nop 1",
          "v_readlane_b32 %s, %s, m0" % (FIRST, ST_FIRST),
          "v_readlane_b32 %s, %s, m0" % (REM, ST_CUR),
          "v_readlane_b32 s72, %s, m0" % ST_LO,
          "v_readlane_b32 s73, %s, m0" % ST_HI,
          "s_nop 0",
          "s_and_b64 %s, %s, %s" % (MINE, TMASK, LIVE),
          "s_cbranch_scc0 LT_pop_%=",
          "s_lshl_b32 %s, %s, 5" % (OFF, FIRST),
          "s_add_u32 s68, %s, %s" % (NODES[0], OFF),
          "s_addc_u32 s69, %s, 0" % NODES[1],
          "s_branch LT_enter_%=",
          "LT_tail_%=:", "@COUNT_TAIL",
          "s_mov_b64 %s, %s" % (UND, NR)]                       # N = mine & ~outside, kept while NR's registers hold `inside on every zero axis`
    for j, k in enumerate(Z):
        mn, mx = mnmx[j]
        L += ["v_cmp_lt_f32 vcc, 0, %s" % mn,
              "v_cmp_lt_f32_e64 %s, %s, %s" % (SQ, mx, side),
              ("s_and_b64 %s, %s, vcc" % (INALL, SQ)) if j == 0 else ("s_and_b64 %s, %s, vcc" % (SQ, SQ))]
        if j == 1:
            L.append("s_and_b64 %s, %s, %s" % (INALL, INALL, SQ))
    # slab arithmetic on the non-zero axes: (plane - e) * inv, ordered by min / max
    n = ["v75", "v77"]
    f = ["v76", "v78"]
    for j, k in enumerate(NZ):
        L += ["v_mul_f32 v32, %s, %s" % (d[2 * k], inv[k]),
              "v_mul_f32 v33, %s, %s" % (d[2 * k + 1], inv[k]),
              "v_min_f32 %s, v32, v33" % n[j],
              "v_max_f32 %s, v32, v33" % f[j]]
    two = len(NZ) == 2
    tN, tF, tN2, tF2, M, span, tc, x, a, r = "v34", "v35", "v36", "v37", "v38", "v39", "v32", "v33", "v79", "v70"
    if two:
        L += ["v_max_f32 %s, %s, %s" % (tN, n[0], n[1]), "v_min_f32 %s, %s, %s" % (tN2, n[0], n[1]),
              "v_min_f32 %s, %s, %s" % (tF, f[0], f[1]), "v_max_f32 %s, %s, %s" % (tF2, f[0], f[1]),
              "v_max_f32_e64 %s, %s, -%s" % (M, tF2, tN2)]
    else:
        tN, tF = n[0], f[0]
        L += ["v_max_f32_e64 %s, %s, -%s" % (M, f[0], n[0])]
    L += ["v_add_f32 %s, %s, %s" % (M, M, T),
          "v_mul_f32 %s, 0x37000000, %s" % (M, M),              # E
          "v_sub_f32 %s, %s, %s" % (span, tF, tN),
          "v_cmp_lt_f32 vcc, 0, %s" % tN,
          "v_cndmask_b32 %s, %s, %s, vcc" % (tc, tF, tN),
          "v_sub_f32 %s, %s, %s" % (x, T, tc),
          "v_min3_f32 %s, %s, %s, %s" % (r, span, tF, x)]
    if two:
        L += ["v_sub_f32 v71, %s, %s" % (tN, tN2),
              "v_sub_f32 v72, %s, %s" % (tF2, tF),
              "v_cndmask_b32 v71, v72, v71, vcc",
              "v_min3_f32 %s, %s, %s, v71" % (a, x, span)]
    else:
        L += ["v_min_f32 %s, %s, %s" % (a, x, span)]
    L += ["v_min_f32 %s, %s, %s" % (a, a, tc),
          "v_cmp_gt_f32_e64 %s, %s, %s" % (ACC, a, M),
          "v_cmp_lt_f32_e64 %s, %s, -%s" % (SQ, r, M),            # rejected by the slab arithmetic
          "s_and_b64 %s, %s, %s" % (ACC, ACC, INALL),
          "s_andn2_b64 %s, %s, %s" % (UND, UND, SQ),               # N & ~rejected
          "s_and_b64 %s, %s, %s" % (ACC, ACC, UND),
          "s_andn2_b64 %s, %s, %s" % (UND, UND, ACC),
          "s_cbranch_scc1 LT_exit_exact_%=",
          "LT_after_exact_%=:",
          "s_cmp_eq_u64 %s, 0" % ACC,
          "s_cbranch_scc1 LT_next_%=",
          "s_bitcmp1_b32 %s, 31" % RMETA,
          "s_cbranch_scc0 LT_leaf_%=",
          "s_and_b32 %s, %s, 0xff" % (OFF, RMETA),
          "s_cbranch_scc0 LT_next_%=",
          "s_cmp_lt_u32 %s, 8" % IDX,
          "s_cbranch_scc0 LT_nopush_%=",
          "v_writelane_b32 %s, %s, m0" % (ST_FIRST, FIRST),
          "v_writelane_b32 %s, %s, m0" % (ST_CUR, REM),
          "v_writelane_b32 %s, s72, m0" % ST_LO,
          "v_writelane_b32 %s, s73, m0" % ST_HI,
          "s_add_u32 m0, m0, 1",
          "LT_nopush_%=:",
          "s_mov_b32 %s, %s" % (FIRST, RFIRST),
          "s_or_b32 %s, %s, 0x100" % (REM, OFF),
          "s_mov_b64 %s, %s" % (TMASK, ACC),
          "s_mov_b64 %s, %s" % (MINE, ACC),
          "s_lshl_b32 %s, %s, 5" % (OFF, RFIRST),
          "s_add_u32 s68, %s, %s" % (NODES[0], OFF),
          "s_addc_u32 s69, %s, 0" % NODES[1],
          "s_branch LT_enter_%="]
    return L


def count_instr(tag):
    """Measurement builds (RT_GEN_COUNT=1, -DRT_STAGE_COUNT): the counter registers are the compiler's, named as operands in rt_kernel_wave.hip."""
    return COUNTERS[tag]


def triangle(q, other, tag, cur, nxt, masked):
    """The numerator stages on the record in s[q : q+15] (at offset `cur`); first the next record to be tested (offset `nxt`) is
    requested into s[other : other+15]: the one behind it (`cur` + 64) in a leaf walked from end to end, the one of the lowest bit
    left in SURV in a leaf whose triangles were put in front of the wave's bundle first (masked)."""
    bxy, bz = "v[68:69]", "v74"
    bx, by = "v68", "v69"
    det, nb, nbs, ng, nt, sg = "v32", "v33", "v34", "v35", "v36", "v39"
    L = ["LT_tri%s_%%=:" % tag,
         "s_waitcnt lgkmcnt(0)",
         # (the request for the record behind this one goes out first: the whole test lies between it and the wait for it)
         "@NEXT",
         "s_load_dwordx16 s[%d:%d], s[%s:%s], %s" % (other, other + 15, TRI[0][1:], TRI[1][1:], nxt),
         "@COUNT_TRI",
         # stage 1: determinant and beta numerator (triangle.cu:117-131), operations paired two to an instruction
         "v_pk_mul_f32 v[64:65], %s, %s" % (C_ZX, s2(q + 10)),                                  # e2.y c2.z , e2.z c2.x
         "v_pk_mul_f32 v[66:67], %s, %s neg_lo:[0,1] neg_hi:[0,1]" % (C_YZ, s2(q + 12)),       # -(e2.z c2.y) , -(e2.x c2.z)
         "v_pk_add_f32 %s, %s, %s neg_lo:[0,1] neg_hi:[0,1]" % (bxy, E_XY, s2(q + 0)),         # e.xy - v0.xy
         "v_pk_add_f32 v[70:71], v[64:65], v[66:67]",                                            # m00 , m10
         "v_pk_mul_f32 v[72:73], %s, %s" % (C_YX, s2(q + 8)),                                   # e2.x c2.y , e2.y c2.x
         "v_subrev_f32 %s, %s, %s" % (bz, s(q + 2), EZ),                                        # e.z - v0.z
         "v_sub_f32 v75, v72, v73",                                                              # m20
         "v_pk_mul_f32 v[76:77], %s, v[70:71]" % s2(q + 4),                                     # e1.x m00 , e1.y m10
         "v_pk_mul_f32 v[78:79], %s, v[70:71]" % bxy,                                           # b.x m00 , b.y m10
         "v_add_f32 v76, v76, v77",
         "v_mul_f32 v77, %s, v75" % s(q + 6),                                                   # e1.z m20
         "v_add_f32 %s, v76, v77" % det,
         "v_add_f32 v78, v78, v79",
         "v_mul_f32 v79, v75, %s" % bz,                                                         # m20 b.z
         "v_add_f32 %s, v78, v79" % nb,
         # 0 <= nb / det <= 1 can only hold when nb lies between 0 and det (either sign): the middle one of (nb, 0, det) is nb itself.
         # (A filter: it lets det == 0 and zeros of either sign through; the quotients decide.)
         "v_med3_f32 %s, %s, 0, %s" % (nbs, nb, det),
         "v_cmp_eq_f32_e64 %s, %s, %s" % (PASS, nbs, nb),
         "s_and_b64 %s, %s, %s" % (PASS, PASS, ACC),
         "s_cbranch_scc0 LT_end%s_%%=" % tag,
         # stage 2: gamma numerator
         "@COUNT_S2",
         "v_mul_f32 v37, %s, %s" % (s(q + 6), C2Y),
         "v_mul_f32 v38, %s, %s" % (s(q + 5), C2Z),
         "v_sub_f32 v37, v37, v38",                                                              # m01
         "v_mul_f32 v38, %s, %s" % (s(q + 4), C2Z),
         "v_mul_f32 %s, %s, %s" % (sg, s(q + 6), C2X),
         "v_sub_f32 v38, v38, %s" % sg,                                                          # m11
         "v_mul_f32 %s, %s, %s" % (sg, s(q + 5), C2X),
         "v_mul_f32 %s, %s, %s" % (ng, s(q + 4), C2Y),
         "v_sub_f32 %s, %s, %s" % (sg, sg, ng),                                                  # m21
         "v_mul_f32 v37, v37, %s" % bx,
         "v_mul_f32 v38, v38, %s" % by,
         "v_add_f32 v37, v37, v38",
         "v_mul_f32 v38, %s, %s" % (sg, bz),
         "v_add_f32 %s, v37, v38" % ng,
         "v_med3_f32 v38, %s, 0, %s" % (ng, det),                                                # the same filter for gamma
         "v_cmp_eq_f32_e64 %s, v38, %s" % (T64, ng),
         "s_and_b64 %s, %s, %s" % (PASS, PASS, T64),
         "s_cbranch_scc0 LT_end%s_%%=" % tag,
         # stage 3: t numerator from the record's e1 x e2; t > 0, and t clearly beyond the closest hit is rejected without dividing
         "@COUNT_S3",
         "v_and_b32 %s, 0x80000000, %s" % (sg, det),
         "v_mul_f32 v37, %s, %s" % (s(q + 7), bx),
         "v_mul_f32 v38, %s, %s" % (s(q + 14), by),
         "v_add_f32 v37, v37, v38",
         "v_mul_f32 v38, %s, %s" % (s(q + 15), bz),
         "v_add_f32 %s, v37, v38" % nt,
         "v_xor_b32 v38, %s, %s" % (sg, nt),
         "v_mul_f32_e64 v37, %s, |%s|" % (T, det),
         "v_mul_f32 v37, 0x3f800008, v37",
         "v_cmp_lt_f32 vcc, 0, v38",
         "v_cmp_ge_f32_e64 %s, v37, v38" % T64,
         "s_and_b64 %s, %s, vcc" % (T64, T64),
         "s_and_b64 %s, %s, %s" % (PASS, PASS, T64),
         "s_cbranch_scc0 LT_end%s_%%=" % tag,
         # candidates: the reference's quotients decide (LT_finish); the walk goes on behind this record
         "s_or_b64 %s, %s, %s" % (CHG, CHG, PASS)] + call_finish(q) + [
         "LT_end%s_%%=:" % tag]
    out = []
    for line in L:
        if line == "@NEXT":
            out += (["s_ff1_i32_b64 %s, %s" % (JT, SURV), "s_bitset0_b64 %s, %s" % (SURV, JT), "s_add_u32 %s, %s, %s" % (JT, JT, RFIRST),
                     "s_lshl_b32 %s, %s, 6" % (nxt, JT)] if masked else ["s_add_u32 %s, %s, 0x40" % (nxt, cur)])
        elif line.startswith("@COUNT"):
            if counting():
                out.append(count_instr(line))
        else:
            out.append(line)
    return out


def call_finish(q):
    """Candidates in PASS (SCC is 1: the mask operation before found some): the lanes of the wave become the candidates, the
    routine below decides them with the reference's quotients and comes back to the branch here with SCC = 0."""
    return ["s_mov_b64 %s, exec" % T64,
            "s_mov_b64 exec, %s" % PASS,
            "s_mov_b32 s28, vcc_lo",                                       # (the loop with four sets keeps the leaf's first record index there)
            "s_mov_b32 s29, %s" % s(q + 3),                                # the triangle's id
            "s_getpc_b64 %s" % TMPP,
            "s_cbranch_scc1 LT_finish_%="]


def divide(out, num, den):
    """out = num / den, correctly rounded (the sequence the compiler emits for a float division: scale, reciprocal, two
    refinements of the reciprocal's error and of the quotient, fix-up).  v34 v37 v38 v39 are its scratch."""
    A, B, C, D = "v37", "v38", "v39", "v34"
    return ["v_div_scale_f32 %s, vcc, %s, %s, %s" % (A, den, den, num),
            "v_rcp_f32 %s, %s" % (B, A),
            "s_nop 0",                                                      # (gfx950: one wait state between a transcendental and an instruction of another kind that reads its result; the assembler adds none in inline assembly)
            "v_fma_f32 %s, -%s, %s, 1.0" % (C, A, B),
            "v_fma_f32 %s, %s, %s, %s" % (B, C, B, B),
            "v_div_scale_f32 %s, vcc, %s, %s, %s" % (D, num, den, num),
            "v_mul_f32 %s, %s, %s" % (out, D, B),
            "v_fma_f32 %s, -%s, %s, %s" % (C, A, out, D),
            "v_fma_f32 %s, %s, %s, %s" % (out, C, B, out),
            "v_fma_f32 %s, -%s, %s, %s" % (C, A, out, D),
            "v_div_fmas_f32 %s, %s, %s, %s" % (out, C, B, out),             # (four instructions behind the scale that wrote vcc: the wait it needs)
            "v_div_fixup_f32 %s, %s, %s, %s" % (out, out, den, num)]


def finish(bary):
    """The final decision for the candidates of a triangle (triangle.cu:131-137, rt_triangle_finish in rt_math.h): beta, gamma and
    t as quotients by the determinant, 0 < t < closest hit so far, 0 <= beta, gamma <= 1, 0 <= 1 - beta - gamma.  The lanes that
    pass take t, the triangle's id and the barycentrics (registers of the caller, or the wave's two columns in LDS)."""
    det, nb, ng, nt = "v32", "v33", "v35", "v36"
    x0, x1, x2, alp = "v64", "v65", "v66", "v67"
    L = ["LT_finish_%=:"]
    L += divide(x0, nb, det) + divide(x1, ng, det) + divide(x2, nt, det)
    L += ["v_sub_f32 %s, 1.0, %s" % (alp, x0),
          "v_sub_f32 %s, %s, %s" % (alp, alp, x1),
          "v_cmpx_lt_f32 vcc, 0, %s" % x2,
          "v_cmpx_gt_f32 vcc, %s, %s" % (T, x2),
          "v_cmpx_le_f32 vcc, 0, %s" % x0,
          "v_cmpx_ge_f32 vcc, 1.0, %s" % x0,
          "v_cmpx_le_f32 vcc, 0, %s" % x1,
          "v_cmpx_ge_f32 vcc, 1.0, %s" % x1,
          "v_cmpx_le_f32 vcc, 0, %s" % alp,
          "v_mov_b32 %s, %s" % (T, x2),
          "v_mov_b32 %[prim], s29"]
    if bary:
        L += ["v_mbcnt_lo_u32_b32 v68, -1, 0",
              "v_mbcnt_hi_u32_b32 v68, -1, v68",
              "v_lshlrev_b32 v68, 2, v68",
              "v_add_u32 v68, %[bary], v68",
              "ds_write_b32 v68, %s" % x0,
              "ds_write_b32 v68, %s offset:%%[boff]" % x1]
    else:
        L += ["v_mov_b32 %%[bet], %s" % x0,
              "v_mov_b32 %%[gam], %s" % x1]
    L += ["s_mov_b32 vcc_lo, s28",
          "s_mov_b64 exec, %s" % T64,
          "s_cmp_lg_u32 0, 0",                                             # SCC = 0: the branch that called falls through
          "s_setpc_b64 %s" % TMPP]
    return L


def cull():
    """Lane k of the wave (all 64 lanes, whatever the walk's lane sets say) and triangle k of the leaf: rt_cull_test of rt_math.h.
    Leaves the triangles worth testing in SURV, or goes on to the next child when there is none."""
    P = ["%%[b%d]" % i for i in range(10)]         # ox oy oz ax ay az r0 ksp k1 bw: scalar operands of the statement (wherever the compiler keeps them)
    ox, oy, oz, ax, ay, az, r0, ksp, k1, bw = P
    L = ["s_waitcnt lgkmcnt(0)",                                      # (a request of the leaf before may still be on its way into those registers)
         "s_mov_b64 %s, exec" % SURV,
         "s_mov_b64 exec, -1",
         "v_mbcnt_lo_u32_b32 v72, -1, 0",
         "v_mbcnt_hi_u32_b32 v72, -1, v72",
         "v_add_u32 v72, %s, v72" % RFIRST,
         "v_lshlrev_b32 v72, 5, v72",
         "v_add_u32 v72, %s, v72" % CULL,
         "global_load_dwordx4 v[64:67], v72, s[%s:%s]" % (TRI[0][1:], TRI[1][1:]),
         "global_load_dwordx4 v[68:71], v72, s[%s:%s] offset:16" % (TRI[0][1:], TRI[1][1:])]
    L += ["s_mov_b32 s46, 0xb6000000",                                # RT_CULL_P2_SLACK
          "s_waitcnt vmcnt(0)",
          "v_subrev_f32 v64, %s, v64" % ox,
          "v_subrev_f32 v65, %s, v65" % oy,
          "v_subrev_f32 v66, %s, v66" % oz,
          "v_mul_f32 v73, v64, v64",
          "v_fma_f32 v73, v65, v65, v73",
          "v_fma_f32 v73, v66, v66, v73",                            # vv
          "v_mul_f32 v74, %s, v64" % ax,
          "v_fma_f32 v74, %s, v65, v74" % ay,
          "v_fma_f32 v74, %s, v66, v74" % az,                        # tc
          "v_fma_f32 v75, -v74, v74, v73",
          "v_fma_f32 v75, s46, v73, v75",
          "v_max_f32 v75, 0, v75",
          "v_sqrt_f32 v75, v75",                                     # p
          "v_sqrt_f32 v76, v73",                                     # |v|
          "v_mul_f32 v77, %s, v68" % ax,
          "v_fma_f32 v77, %s, v69, v77" % ay,
          "v_fma_f32 v77, %s, v70, v77" % az,
          "v_sub_f32_e64 v77, |v77|, %s" % ksp,                      # cosn
          "v_mov_b32 v78, %s" % r0,
          "v_fma_f32 v78, v67, %s, v78" % k1,
          "v_fma_f32 v78, %s, v76, v78" % ksp,                       # reach
          "v_sub_f32 v75, v75, v78",
          "v_mul_f32 v75, v75, v77",                                 # lhs
          "v_add_f32 v79, %s, v76" % bw,
          "v_fma_f32 v79, 4.0, v67, v79",
          "v_mul_f32 v79, v71, v79",
          "v_mul_f32 v79, %s, v79" % k1,                             # rhs
          "v_cmp_gt_f32_e64 %s, v75, v79" % T64,
          "v_cmp_lt_f32 vcc, 0x3a800000, v77",                       # RT_CULL_GUARD < cosn
          "s_and_b64 %s, %s, vcc" % (T64, T64),
          "s_mov_b64 exec, %s" % SURV,
          "s_bfm_b64 %s, %s, 0" % (SURV, TEND),
          "s_andn2_b64 %s, %s, %s" % (SURV, SURV, T64),
          "s_cbranch_scc0 LT_next_%="]
    return L


def leaf(with_cull):
    L = ["LT_leaf_%=:", "@COUNT_LEAF",
         "s_and_b32 %s, %s, 0xff" % (TEND, RMETA),
         "s_cbranch_scc0 LT_next_%="]
    if os.environ.get("RT_GEN_COUNT") == "leaf":
        # (-DRT_STAGE_COUNT=2) how many triangles leaves hold that few rays of the wave enter: s14 all, s15 / s16 entered by at most 2 / 4 rays, s17 triangles x rays
        L += ["s_bcnt1_i32_b64 s30, %s" % ACC,
              "s_add_u32 s14, s14, %s" % TEND,
              "s_cmp_le_u32 s30, 2", "s_cselect_b32 s31, %s, 0" % TEND, "s_add_u32 s15, s15, s31",
              "s_cmp_le_u32 s30, 4", "s_cselect_b32 s31, %s, 0" % TEND, "s_add_u32 s16, s16, s31",
              "s_mul_i32 s31, s30, %s" % TEND, "s_add_u32 s17, s17, s31"]
    if with_cull:
        L += leaf_masked()
    L += leaf_all()
    return L


def leaf_masked():
    """A leaf entered with a bundle: the triangles that survive it, lowest bit first."""
    L = ["s_cmp_eq_u32 %s, 0" % CULL,
         "s_cbranch_scc1 LT_leaf_all_%=",
         "s_bitcmp1_b32 %s, 8" % RMETA,                              # RT_LEAF_ASK: the host found the leaf's triangles worth asking about
         "s_cbranch_scc0 LT_leaf_all_%="]
    L += cull()
    if BATCH:
        L += ["s_bcnt1_i32_b64 %s, %s" % (JT, SURV),
              "s_cmp_ge_u32 %s, %d" % (JT, BIG),
              "s_cbranch_scc1 LT_big_masked_%=",
              "LT_small_masked_%=:"]
    L += ["s_bitset1_b64 %s, 63" % SURV,
          "s_mov_b64 %s, 0" % CHG,
          "@PRIO",
          "s_ff1_i32_b64 %s, %s" % (JT, SURV),
          "s_bitset0_b64 %s, %s" % (SURV, JT),
          "s_add_u32 %s, %s, %s" % (JT, JT, RFIRST),
          "s_lshl_b32 %s, %s, 6" % (TCUR, JT),
          "LT_leaf_load_%=:",
          "s_waitcnt lgkmcnt(0)",
          "s_load_dwordx16 s[%d:%d], s[%s:%s], %s" % (TA, TA + 15, TRI[0][1:], TRI[1][1:], TCUR)]
    L += triangle(TA, TB, "A", TCUR, TNXT, True)
    L += ["s_cmp_eq_u64 %s, 0" % SURV,                                # the record requested meanwhile is the end marker's
          "s_cbranch_scc1 LT_leaf_done_%="]
    L += triangle(TB, TA, "B", TNXT, TCUR, True)
    L += ["s_cmp_eq_u64 %s, 0" % SURV,
          "s_cbranch_scc0 LT_triA_%=",
          "s_branch LT_leaf_done_%="]
    return L


def leaf_all():
    """Without a bundle: every triangle of the leaf, one record after the other."""
    L = ["LT_leaf_all_%=:"]
    if BATCH:
        L += ["s_cmp_ge_u32 %s, %d" % (TEND, BIG),
              "s_cbranch_scc1 LT_big_all_%=",
              "LT_small_all_%=:"]
    L += ["s_lshl_b32 %s, %s, 6" % (TCUR, RFIRST),
          "s_lshl_b32 %s, %s, 6" % (TEND, TEND),
          "s_add_u32 %s, %s, %s" % (TEND, TEND, TCUR),
          "s_mov_b64 %s, 0" % CHG,
          "@PRIO",
          "LT_leaf_load_all_%=:",
          "s_waitcnt lgkmcnt(0)",
          "s_load_dwordx16 s[%d:%d], s[%s:%s], %s" % (TA, TA + 15, TRI[0][1:], TRI[1][1:], TCUR)]
    L += triangle(TA, TB, "C", TCUR, TNXT, False)
    L += ["s_cmp_lt_u32 %s, %s" % (TNXT, TEND),
          "s_cbranch_scc0 LT_leaf_done_%="]
    L += triangle(TB, TA, "D", TNXT, TCUR, False)
    L += ["s_cmp_lt_u32 %s, %s" % (TCUR, TEND),
          "s_cbranch_scc1 LT_triC_%=",
          "LT_leaf_done_%=:",
          "s_cmp_eq_u64 %s, 0" % CHG,
          "s_cbranch_scc1 LT_next_%=",
          # some ray's closest hit moved: shadow rays that are now known to be occluded drop out (ray.cu:70)
          "LT_leaf_changed_%=:",
          "v_cmp_le_f32 vcc, 0, %s" % STOP,
          "v_cmp_lt_f32_e64 %s, %s, %s" % (T64, T, STOP),
          "s_and_b64 %s, %s, vcc" % (T64, T64),
          "s_andn2_b64 %s, %s, %s" % (LIVE, LIVE, T64),
          "s_cbranch_scc0 LT_done_%=",
          "s_and_b64 %s, %s, %s" % (MINE, TMASK, LIVE),
          "s_cbranch_scc1 LT_next_%=",
          "s_branch LT_pop_%="]
    return L



def batch_issue(slot, tmp):
    """The next triangle to test of the open leaf: its index into J[slot], its record on its way into record set `slot`."""
    q = RSET[slot]
    return ["s_ff1_i32_b64 %s, %s" % (J[slot], SURVB),
            "s_bitset0_b64 %s, %s" % (SURVB, J[slot]),
            "s_bitset1_b64 %s, 63" % SURVB,                                 # (the end marker stays: asked again it answers 63 again)
            "s_add_u32 %s, %s, %s" % (tmp, J[slot], RF),
            "s_lshl_b32 %s, %s, 6" % (tmp, tmp),                            # (bit 30 of RF leaves with the shift)
            "s_load_dwordx16 s[%d:%d], s[%s:%s], %s" % (q, q + 15, TRI[0][1:], TRI[1][1:], tmp)]


def batch_test(slot):
    """The three numerator stages on record set `slot`; candidates are decided by LT_finish."""
    q = RSET[slot]
    bxy, bz = "v[68:69]", "v74"
    bx, by = "v68", "v69"
    det, nb, nbs, ng, nt, sg = "v32", "v33", "v34", "v35", "v36", "v39"
    tag = "%d" % slot
    L = ["LT_t%s_%%=:" % tag,
         "s_cmp_eq_u32 %s, 63" % J[slot],
         "s_cbranch_scc1 LT_unpark_%=",
         "@COUNT_TRI",
         "v_pk_mul_f32 v[64:65], %s, %s" % (C_ZX, s2(q + 10)),
         "v_pk_mul_f32 v[66:67], %s, %s neg_lo:[0,1] neg_hi:[0,1]" % (C_YZ, s2(q + 12)),
         "v_pk_add_f32 %s, %s, %s neg_lo:[0,1] neg_hi:[0,1]" % (bxy, E_XY, s2(q + 0)),
         "v_pk_add_f32 v[70:71], v[64:65], v[66:67]",
         "v_pk_mul_f32 v[72:73], %s, %s" % (C_YX, s2(q + 8)),
         "v_subrev_f32 %s, %s, %s" % (bz, s(q + 2), EZ),
         "v_sub_f32 v75, v72, v73",
         "v_pk_mul_f32 v[76:77], %s, v[70:71]" % s2(q + 4),
         "v_pk_mul_f32 v[78:79], %s, v[70:71]" % bxy,
         "v_add_f32 v76, v76, v77",
         "v_mul_f32 v77, %s, v75" % s(q + 6),
         "v_add_f32 %s, v76, v77" % det,
         "v_add_f32 v78, v78, v79",
         "v_mul_f32 v79, v75, %s" % bz,
         "v_add_f32 %s, v78, v79" % nb,
         "v_med3_f32 %s, %s, 0, %s" % (nbs, nb, det),
         "v_cmp_eq_f32_e64 %s, %s, %s" % (PASS, nbs, nb),
         "s_and_b64 %s, %s, %s" % (PASS, PASS, ACC),
         "s_cbranch_scc0 LT_e%s_%%=" % tag,
         "@COUNT_S2",
         "v_mul_f32 v37, %s, %s" % (s(q + 6), C2Y),
         "v_mul_f32 v38, %s, %s" % (s(q + 5), C2Z),
         "v_sub_f32 v37, v37, v38",
         "v_mul_f32 v38, %s, %s" % (s(q + 4), C2Z),
         "v_mul_f32 %s, %s, %s" % (sg, s(q + 6), C2X),
         "v_sub_f32 v38, v38, %s" % sg,
         "v_mul_f32 %s, %s, %s" % (sg, s(q + 5), C2X),
         "v_mul_f32 %s, %s, %s" % (ng, s(q + 4), C2Y),
         "v_sub_f32 %s, %s, %s" % (sg, sg, ng),
         "v_mul_f32 v37, v37, %s" % bx,
         "v_mul_f32 v38, v38, %s" % by,
         "v_add_f32 v37, v37, v38",
         "v_mul_f32 v38, %s, %s" % (sg, bz),
         "v_add_f32 %s, v37, v38" % ng,
         "v_med3_f32 v38, %s, 0, %s" % (ng, det),
         "v_cmp_eq_f32_e64 %s, v38, %s" % (T64, ng),
         "s_and_b64 %s, %s, %s" % (PASS, PASS, T64),
         "s_cbranch_scc0 LT_e%s_%%=" % tag,
         "@COUNT_S3",
         "v_and_b32 %s, 0x80000000, %s" % (sg, det),
         "v_mul_f32 v37, %s, %s" % (s(q + 7), bx),
         "v_mul_f32 v38, %s, %s" % (s(q + 14), by),
         "v_add_f32 v37, v37, v38",
         "v_mul_f32 v38, %s, %s" % (s(q + 15), bz),
         "v_add_f32 %s, v37, v38" % nt,
         "v_xor_b32 v38, %s, %s" % (sg, nt),
         "v_mul_f32_e64 v37, %s, |%s|" % (T, det),
         "v_mul_f32 v37, 0x3f800008, v37",
         "v_cmp_lt_f32_e64 %s, 0, v38" % TMPP,
         "v_cmp_ge_f32_e64 %s, v37, v38" % T64,
         "s_and_b64 %s, %s, %s" % (T64, T64, TMPP),
         "s_and_b64 %s, %s, %s" % (PASS, PASS, T64),
         "s_cbranch_scc0 LT_e%s_%%=" % tag,
         "s_bitset1_b32 %s, 30" % RF]
    L += call_finish(q)
    L += ["LT_e%s_%%=:" % tag]
    return L


def leaf_big(with_cull):
    """The loop with four record sets.  Entered at LT_big (SURVB: the triangles to test, no marker yet) with nothing parked."""
    def long_wave(count, back):
        return ["s_add_u32 %s, %s, %s" % (WORK, WORK, count),
                "s_cmp_lt_u32 %s, %d" % (WORK, WORK0),
                "s_cbranch_scc1 %s" % back]
    L = []
    if with_cull:
        L += ["LT_big_masked_%=:"] + long_wave(JT, "LT_small_masked_%=") + [
              "s_mov_b64 %s, %s" % (SURVB, SURV),
              "s_branch LT_big_%="]
    L += ["LT_big_all_%=:"] + long_wave(TEND, "LT_small_all_%=") + [
         "s_bfm_b64 %s, %s, 0" % (SURVB, TEND),
         "LT_big_%=:",
         "s_bitset1_b64 %s, 63" % SURVB,
         "s_mov_b32 %s, %s" % (RF, RFIRST)]
    L += ["v_writelane_b32 %s, s%d, %d" % (ST_CUR, r, 32 + i) for i, r in enumerate(PARKED)]
    L += ["LT_big_start_%=:"]
    L += batch_issue(0, TMP0) + batch_issue(1, TMP1)
    L += ["LT_lp_%=:",
          "s_waitcnt lgkmcnt(0)",
          "s_lshl_b64 %s, %s, 1" % (TMPP, SURVB),                       # nothing but the end marker left: no further requests
          "s_cbranch_scc0 LT_noq_%="]
    L += batch_issue(2, TMP0) + batch_issue(3, TMP1)
    L += ["LT_p0_%=:"]
    L += batch_test(0) + batch_test(1)
    L += ["s_waitcnt lgkmcnt(0)",
          "s_lshl_b64 %s, %s, 1" % (TMPP, SURVB),
          "s_cbranch_scc0 LT_nop_%="]
    L += batch_issue(0, TMP0) + batch_issue(1, TMP1)
    L += ["LT_q0_%=:"]
    L += batch_test(2) + batch_test(3)
    L += ["s_branch LT_lp_%=",
          "LT_noq_%=:", "s_mov_b32 %s, 63" % J[2], "s_mov_b32 %s, 63" % J[3], "s_branch LT_p0_%=",
          "LT_nop_%=:", "s_mov_b32 %s, 63" % J[0], "s_mov_b32 %s, 63" % J[1], "s_branch LT_q0_%=",
          "LT_unpark_%=:",
          "s_waitcnt lgkmcnt(0)"]                                        # (a record still on its way would land in the state that comes back now)
    L += ["v_readlane_b32 s%d, %s, %d" % (r, ST_CUR, 32 + i) for i, r in enumerate(PARKED)]
    L += ["s_nop 1",
          "s_mov_b32 %s, %d" % (WORK, WORK0),                            # (J[3]'s register: the walk stays one with a long life)
          "s_lshl_b32 %s, %s, 5" % (OFF, IDX),
          "s_load_dwordx8 s[84:91], %s, %s" % (BLK, OFF),               # the sibling behind the leaf, whose record gave way
          "s_bitcmp1_b32 %s, 30" % RF,
          "s_cbranch_scc0 LT_next_%=",
          "s_branch LT_leaf_changed_%="]
    return L


def block(pattern, with_cull, bary):
    # The block leaves for one thing only and is entered again afterwards: the exact node test of lanes that the
    # classification could not decide (RES = 1).  RES = 0: the walk is over.
    L = ["s_waitcnt lgkmcnt(0)",
         "s_cmp_eq_u32 %s, 1" % RES,
         "s_cbranch_scc1 LT_resume_exact_%="]
    L += scan_sorted() if pattern == 0 else scan_zero(pattern)
    L += leaf(with_cull)
    if BATCH:
        L += leaf_big(with_cull)
    L += finish(bary)
    L += [# ---- re-entry (the record of the next sibling came along: s[84:91] are operands that the caller carries)
          "LT_resume_exact_%=:",
          "s_branch LT_after_exact_%=",
          "LT_exit_exact_%=:",
          "s_mov_b32 %s, 1" % RES,
          "s_branch LT_out_%=",
          "LT_done_%=:",
          "s_mov_b32 %s, 0" % RES,
          "LT_out_%=:",
          "s_waitcnt lgkmcnt(0)"]
    return L


def statement(with_cull, bary):
    # ONE assembly statement with the seven copies behind a dispatch on the pattern (s18): several statements in a
    # switch would have the compiler merge their scalar outputs through values it no longer knows to be wave-uniform
    L = ["s_cmp_eq_u32 s18, 0", "s_cbranch_scc1 LT0_begin_%="]       # (nearly every walk)
    for pattern in range(1, 6):
        L += ["s_cmp_eq_u32 s18, %d" % pattern, "s_cbranch_scc1 LT%d_begin_%%=" % pattern]
    L += ["s_branch LT6_begin_%="]
    for pattern in range(7):
        B = ["LT%d_begin_%%=:" % pattern] + block(pattern, with_cull, bary) + ["s_branch LT_end_all_%="]
        L += [line.replace("LT_", "LT%d_" % pattern) if "LT_end_all" not in line else line for line in B]
        print("cull" if with_cull else "plain", "pattern", pattern, len(B), "lines")
    L += ["LT_end_all_%=:"]
    return L


COUNTERS = {"@COUNT_TRI": "s_add_u32 s14, s14, 1", "@COUNT_S2": "s_add_u32 s17, s17, 1", "@COUNT_S3": "s_add_u32 s17, s17, 0x10000", "@COUNT_POP": "s_add_u32 s16, s16, 0x10000", "@COUNT_TAIL": "s_add_u32 s16, s16, 1", "@COUNT_LEAF": "s_add_u32 s15, s15, 0x10000"}


def emit(line):
    return '"%s\\n"' % line


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracer_amd", "csrc", "rt_trace_asm.inc")
    if len(sys.argv) > 1:
        out = sys.argv[1]
    with open(out, "w") as f:
        f.write("// Generated by tools/gen_trace_asm.py - do not edit.  Register map and operands: rt_kernel_wave.hip (walk_asm, trace_wave).\n")
        f.write("// RT_TRACE_ASM_CULL_*: the walk with the bundle test in front of a leaf's triangles (operands %[b0] .. %[b9], s19);\n")
        f.write("// RT_TRACE_ASM_PLAIN_*: the walk that tests every triangle of every leaf it enters (kernels whose rays never form a bundle).\n")
        f.write("// *_REG: the barycentrics of an accepted triangle go to the operands %[bet] / %[gam]; *_LDS: to the wave's two columns in LDS (%[bary], %[boff]).\n")
        f.write("// In all, one copy per kind of node test: rays that agree on their direction signs (s18 = 0), rays that share a zero pattern (s18 = 1 .. 6, bit k: component k is zero).\n")
        for name, with_cull, bary in (("RT_TRACE_ASM_CULL_REG", True, False), ("RT_TRACE_ASM_CULL_LDS", True, True),
                                      ("RT_TRACE_ASM_PLAIN_REG", False, False), ("RT_TRACE_ASM_PLAIN_LDS", False, True)):
            f.write("#define %s \\\n" % name)
            body = statement(with_cull, bary)
            if os.environ.get("RT_GEN_NOLOAD"):
                # (measurement: what a triangle iteration costs without the wait for its record.  Only the triangle loops: without
                #  the siblings' records the scan would descend into the same child again and again)
                body = ["s_nop 0" if ("s_load_dwordx16" in l and i and ("s_add_u32" in body[i - 1] or "s_lshl_b32 s2" in body[i - 1])) else l for i, l in enumerate(body)]
            lines = [(count_instr(l) if l in COUNTERS else l) if counting() else l for l in body if l != "@PRIO" and (counting() or l not in COUNTERS)]
            f.write(" \\\n".join(emit(line) for line in lines))
            f.write("\n")
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
