// wave_prims.hpp — 64-lane wavefront scans/reductions on gfx950 with DPP
// (data-parallel primitives: no LDS traffic, ~1 VALU op per step).
// DPP controls (GFX9): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ldsp {

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                               CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_i(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_d(double old, double v) {
  unsigned long long o = __builtin_bit_cast(unsigned long long, old), s = __builtin_bit_cast(unsigned long long, v);
  int lo = __builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)s, CTRL, ROW_MASK, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(s >> 32), CTRL, ROW_MASK, 0xf, false);
  unsigned long long r = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
  return __builtin_bit_cast(double, r);
}

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  unsigned long long s = __builtin_bit_cast(unsigned long long, v);
  unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)s, l);
  unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(s >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// max / min without the canonicalising `v_max_f32 x, x, x` hipcc puts in front of fmaxf/fminf when it
// cannot prove an operand is not a signalling NaN (a NaN operand is ignored, as with fmaxf)
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// ---- fused DPP steps -------------------------------------------------------------
// One VALU instruction per scan step: `v = op(dpp(v), v)` in place.  Without bound_ctrl a
// lane whose DPP source is out of range (or whose row is masked off) is simply not
// written, i.e. keeps v — the identity of every step below, so no `old` operand and no
// separate v_mov_b32_dpp.  hipcc only fuses integer adds by itself (0.0 is not an fadd
// identity for -0.0, fmax gets a canonicalisation), hence the asm.  s_nop 1 = the two
// wait states a DPP read needs after a VALU write of the same VGPR; the first step of a
// sequence uses s_nop 4 (covers a preceding v_cmpx write of EXEC as well).
#define LDSP_DPP2(NOPS, OP, CTL, v) asm("s_nop " #NOPS "\n\t" OP " %0, %0, %0 " CTL : "+v"(v))
#define LDSP_DPP3(NOPS, OP, CTL, v, p) asm("s_nop " #NOPS "\n\t" OP " %0, %0, %1 " CTL : "+v"(v) : "v"(p))
#define LDSP_ROWS "row_mask:0xf bank_mask:0xf"

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ float wave_incl_scan_sum(float v) {
  LDSP_DPP2(4, "v_add_f32_dpp", "row_shr:1 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_add_f32_dpp", "row_shr:2 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_add_f32_dpp", "row_shr:4 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_add_f32_dpp", "row_shr:8 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_add_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf", v);
  LDSP_DPP2(1, "v_add_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf", v);
  return v;
}
__device__ __forceinline__ double wave_incl_scan_sum_f64(double v) {
  v += dpp_d<0x111>(0.0, v);
  v += dpp_d<0x112>(0.0, v);
  v += dpp_d<0x114>(0.0, v);
  v += dpp_d<0x118>(0.0, v);
  v += dpp_d<0x142, 0xa>(0.0, v);
  v += dpp_d<0x143, 0xc>(0.0, v);
  return v;
}
__device__ __forceinline__ int wave_incl_scan_sum_i(int v) {
  v += dpp_i<0x111>(0, v);
  v += dpp_i<0x112>(0, v);
  v += dpp_i<0x114>(0, v);
  v += dpp_i<0x118>(0, v);
  v += dpp_i<0x142, 0xa>(0, v);
  v += dpp_i<0x143, 0xc>(0, v);
  return v;
}
// reductions: result in every lane (scan, then broadcast lane 63 through an SGPR)
__device__ __forceinline__ float wave_sum_all(float v) { return readlane_f(wave_incl_scan_sum(v), 63); }
__device__ __forceinline__ double wave_sum_all_f64(double v) { return readlane_d(wave_incl_scan_sum_f64(v), 63); }
__device__ __forceinline__ int wave_sum_all_i(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_sum_i(v), 63); }
__device__ __forceinline__ float wave_max_all(float v) {
  LDSP_DPP2(4, "v_max_f32_dpp", "row_shr:1 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_max_f32_dpp", "row_shr:2 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_max_f32_dpp", "row_shr:4 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_max_f32_dpp", "row_shr:8 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_max_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf", v);
  LDSP_DPP2(1, "v_max_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf", v);
  return readlane_f(v, 63);
}
__device__ __forceinline__ float wave_min_all(float v) {
  LDSP_DPP2(4, "v_min_f32_dpp", "row_shr:1 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_min_f32_dpp", "row_shr:2 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_min_f32_dpp", "row_shr:4 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_min_f32_dpp", "row_shr:8 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_min_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf", v);
  LDSP_DPP2(1, "v_min_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf", v);
  return readlane_f(v, 63);
}

// unsigned minimum over the wave, result in every lane
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  LDSP_DPP2(4, "v_min_u32_dpp", "row_shr:1 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_min_u32_dpp", "row_shr:2 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_min_u32_dpp", "row_shr:4 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_min_u32_dpp", "row_shr:8 " LDSP_ROWS, v);
  LDSP_DPP2(1, "v_min_u32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf", v);
  LDSP_DPP2(1, "v_min_u32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf", v);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Inclusive scan of the recurrence s_l = v_l + a*s_{l-1} over the lanes (one-pole
// IIR with constant per-lane decay a): s_l = sum_{j<=l} a^(l-j) v_j.
// The caller supplies exact powers of a (derived in double on the host):
// p[0..3] = a^1, a^2, a^4, a^8 (the decay of each doubling step inside a 16-lane
// row), f15 = a^((lane&15)+1), f31 = a^((lane&31)+1) (row-total broadcasts).
struct AffinePow { float p1, p2, p4, p8; };
__device__ __forceinline__ float wave_incl_scan_affine(float v, const AffinePow& P, float f15, float f31) {
  const float p1 = P.p1, p2 = P.p2, p4 = P.p4, p8 = P.p8;  // VOP2 DPP wants the multiplier in a VGPR
  LDSP_DPP3(4, "v_fmac_f32_dpp", "row_shr:1 " LDSP_ROWS, v, p1);   // v += a^1 * v[l-1]
  LDSP_DPP3(1, "v_fmac_f32_dpp", "row_shr:2 " LDSP_ROWS, v, p2);
  LDSP_DPP3(1, "v_fmac_f32_dpp", "row_shr:4 " LDSP_ROWS, v, p4);
  LDSP_DPP3(1, "v_fmac_f32_dpp", "row_shr:8 " LDSP_ROWS, v, p8);
  LDSP_DPP3(1, "v_fmac_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf", v, f15);
  LDSP_DPP3(1, "v_fmac_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf", v, f31);
  return v;
}
// mirror image: s_l = v_l + a*s_{l+1}  (anti-causal).  In-row steps by row_shl; there is no
// reverse row broadcast, so the three row heads go through SGPRs (readlane) and are folded
// with a16 = a^16; frow = a^(16-(lane&15)) carries a lane to the head of the next row.
__device__ __forceinline__ float wave_incl_scan_affine_rev(float v, const AffinePow& P, float a16, float frow) {
  const float p1 = P.p1, p2 = P.p2, p4 = P.p4, p8 = P.p8;
  LDSP_DPP3(4, "v_fmac_f32_dpp", "row_shl:1 " LDSP_ROWS, v, p1);   // v += a^1 * v[l+1]
  LDSP_DPP3(1, "v_fmac_f32_dpp", "row_shl:2 " LDSP_ROWS, v, p2);
  LDSP_DPP3(1, "v_fmac_f32_dpp", "row_shl:4 " LDSP_ROWS, v, p4);
  LDSP_DPP3(1, "v_fmac_f32_dpp", "row_shl:8 " LDSP_ROWS, v, p8);
  const float s1 = readlane_f(v, 16), s2 = readlane_f(v, 32), s3 = readlane_f(v, 48);
  const float t2 = fmaf(a16, s3, s2), t1 = fmaf(a16, t2, s1);
  const int row = (threadIdx.x & 63) >> 4;
  const float sel = (row == 0) ? t1 : (row == 1) ? t2 : (row == 2) ? s3 : 0.f;
  return fmaf(frow, sel, v);
}
// The same for FOUR independent chains in one asm statement (as LDSP_DPP_GROUP4 below: the other chains fill the two wait states
// between a step's write and the next step's DPP read of the same register, so no s_nop per step).  The four in-row multipliers are
// wave-uniform: they arrive in scalar registers and pass through ONE temporary vector register (a DPP instruction takes no scalar
// source), so the statement holds seven vector registers, not ten.
#define LDSP_AFF4_STEP(CTL, M) "v_fmac_f32_dpp %0, %0, " M " " CTL "\n\tv_fmac_f32_dpp %1, %1, " M " " CTL "\n\tv_fmac_f32_dpp %2, %2, " M " " CTL "\n\tv_fmac_f32_dpp %3, %3, " M " " CTL "\n\t"
__device__ __forceinline__ void wave_incl_scan_affine4(float (&v)[4], const AffinePow& P, float f15, float f31) {
  float tmp;
  asm volatile("v_mov_b32 %4, %5\n\ts_nop 0\n\t" LDSP_AFF4_STEP("row_shr:1 row_mask:0xf bank_mask:0xf", "%4")
               "v_mov_b32 %4, %6\n\t" LDSP_AFF4_STEP("row_shr:2 row_mask:0xf bank_mask:0xf", "%4")
               "v_mov_b32 %4, %7\n\t" LDSP_AFF4_STEP("row_shr:4 row_mask:0xf bank_mask:0xf", "%4")
               "v_mov_b32 %4, %8\n\t" LDSP_AFF4_STEP("row_shr:8 row_mask:0xf bank_mask:0xf", "%4")
               LDSP_AFF4_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf", "%9") LDSP_AFF4_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf", "%10")
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "=&v"(tmp) : "s"(P.p1), "s"(P.p2), "s"(P.p4), "s"(P.p8), "v"(f15), "v"(f31));
}
// anti-causal: the four in-row steps of four chains in one statement; the row heads are folded as in wave_incl_scan_affine_rev
__device__ __forceinline__ void wave_incl_scan_affine_rev4(float (&v)[4], const AffinePow& P, float a16, float frow) {
  float tmp;
  asm volatile("v_mov_b32 %4, %5\n\ts_nop 0\n\t" LDSP_AFF4_STEP("row_shl:1 row_mask:0xf bank_mask:0xf", "%4")
               "v_mov_b32 %4, %6\n\t" LDSP_AFF4_STEP("row_shl:2 row_mask:0xf bank_mask:0xf", "%4")
               "v_mov_b32 %4, %7\n\t" LDSP_AFF4_STEP("row_shl:4 row_mask:0xf bank_mask:0xf", "%4")
               "v_mov_b32 %4, %8\n\t" LDSP_AFF4_STEP("row_shl:8 row_mask:0xf bank_mask:0xf", "%4")
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "=&v"(tmp) : "s"(P.p1), "s"(P.p2), "s"(P.p4), "s"(P.p8));
  const int row = (threadIdx.x & 63) >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float s1 = readlane_f(v[r], 16), s2 = readlane_f(v[r], 32), s3 = readlane_f(v[r], 48);
    const float t2 = fmaf(a16, s3, s2), t1 = fmaf(a16, t2, s1);
    const float sel = (row == 0) ? t1 : (row == 1) ? t2 : (row == 2) ? s3 : 0.f;
    v[r] = fmaf(frow, sel, v[r]);
  }
}
#undef LDSP_AFF4_STEP
// value of lane l+1 (0 for lane 63) / lane l-1 (0 for lane 0)
__device__ __forceinline__ float wave_shl1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// A wave's ballots of 16 rows are collected in ONE register: v_writelane deposits the two halves of ballot `slot` in lanes
// 2*slot and 2*slot+1 (+32 for a second mask); the words then go to LDS with a single ds_write_b32 of 32 (64) lanes
// instead of a compare, a branch, two moves and a store per ballot.
// (ballots a, b -> lanes lane0 .. lane0+3; the s_nop separates the v_cmp that wrote the SGPRs from their first read here —
// hipcc does not see hazards of instructions inside an asm statement)
__device__ __forceinline__ void put_ballots(uint32_t& acc, unsigned long long a, unsigned long long b, int lane0) {
  asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, %5\n\tv_writelane_b32 %0, %2, %6\n\tv_writelane_b32 %0, %3, %7\n\tv_writelane_b32 %0, %4, %8"
               : "+v"(acc)
               : "s"((uint32_t)a), "s"((uint32_t)(a >> 32)), "s"((uint32_t)b), "s"((uint32_t)(b >> 32)), "i"(lane0), "i"(lane0 + 1), "i"(lane0 + 2),
                 "i"(lane0 + 3));
}

}  // namespace ldsp

// ---- interleaved DPP chains ------------------------------------------------------------------------------------
// A wave issues one instruction every ~8-9 cycles whatever its kind (tools/micro/valu_rate3.hip), so the `s_nop` in front of
// every step of the scans above costs as much as the step itself.  LDSP_DPP_GROUPn(op0, v0, op1, v1, ...) runs n INDEPENDENT
// chains (op = "v_add_f32_dpp", "v_max_f32_dpp", "v_min_f32_dpp", "v_min_u32_dpp") interleaved: one asm statement per step,
// holding that step of every chain, with ALL values as operands — the compiler then cannot place a write of one of them
// between the leading s_nop and its first DPP read (with separate statements it did: a scan that started from a stale
// register).  With three or more chains the two wait states a DPP read needs after a VALU write of the same VGPR are filled by
// the other chains; one or two chains keep an s_nop in every step.  After the last step lane 63 holds the wave total /
// maximum / minimum and lane l the inclusive scan (sums).
// (generated: ONE asm statement per group — all six steps — so that neither the scheduler nor the register allocator can put a
// copy or a reload between a step's write and the next step's DPP read; the hazard recogniser does not look inside inline asm)
// ---- butterfly reductions (round 4): several values reduced over the wave TOGETHER.  v_permlane32_swap_b32 a, b exchanges the upper half of a
// with the lower half of b, so OP(a, b) afterwards holds 32 partials of a in lanes 0-31 and 32 partials of b in lanes 32-63; v_permlane16_swap_b32
// does the same for the odd rows of its first operand and the even rows of its second.  Four values: 3 swaps + 3 OPs + 4 DPP steps inside a
// row (ten instructions where four separate reductions take 24); the TOTALS end in the last lane of each row of `a`:
//   lane 15: a,   lane 31: c,   lane 47: b,   lane 63: d.
// Two values: 1 swap + 1 OP + 5 DPP steps; totals in lane 31 (a) and lane 63 (b) of `a`.  Only for order-independent OPs where bits matter
// (max / min): a sum's rounding depends on the order.  OP / OPD: the plain and the DPP mnemonic ("v_max_f32" / "v_max_f32_dpp").
// (s_nop: a VALU result is not read by a lane-crossing instruction in the next two issue slots.)
#define LDSP_BFLY4(OP, OPD, a, b, c, d) \
  asm volatile( \
    "s_nop 1\n\t" "v_permlane32_swap_b32 %0, %1\n\t" "v_permlane32_swap_b32 %2, %3\n\t" "s_nop 1\n\t" \
    OP " %0, %0, %1\n\t" OP " %2, %2, %3\n\t" "s_nop 1\n\t" "v_permlane16_swap_b32 %0, %2\n\t" "s_nop 1\n\t" OP " %0, %0, %2\n\t" "s_nop 1\n\t" \
    OPD " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" "s_nop 1\n\t" OPD " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" "s_nop 1\n\t" \
    OPD " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" "s_nop 1\n\t" OPD " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" \
    : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define LDSP_BFLY2(OP, OPD, a, b) \
  asm volatile( \
    "s_nop 1\n\t" "v_permlane32_swap_b32 %0, %1\n\t" "s_nop 1\n\t" OP " %0, %0, %1\n\t" "s_nop 1\n\t" \
    OPD " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" "s_nop 1\n\t" OPD " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" "s_nop 1\n\t" \
    OPD " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" "s_nop 1\n\t" OPD " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" "s_nop 1\n\t" \
    OPD " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" \
    : "+v"(a), "+v"(b))
#define LDSP_DPP_GROUP1(O0, v0) \
  asm volatile( \
    "s_nop 1\n\t" O0 " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" \
    "s_nop 1\n\t" O0 " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" \
    "s_nop 1\n\t" O0 " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" \
    "s_nop 1\n\t" O0 " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" \
    "s_nop 1\n\t" O0 " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" \
    "s_nop 1\n\t" O0 " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" \
    : "+v"(v0))
#define LDSP_DPP_GROUP2(O0, v0, O1, v1) \
  asm volatile( \
    "s_nop 1\n\t" O0 " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" \
    "s_nop 0\n\t" O0 " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" \
    "s_nop 0\n\t" O0 " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" \
    "s_nop 0\n\t" O0 " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" \
    "s_nop 0\n\t" O0 " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" \
    "s_nop 0\n\t" O0 " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" \
    : "+v"(v0), "+v"(v1))
#define LDSP_DPP_GROUP3(O0, v0, O1, v1, O2, v2) \
  asm volatile( \
    "s_nop 1\n\t" O0 " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf" \
    : "+v"(v0), "+v"(v1), "+v"(v2))
#define LDSP_DPP_GROUP4(O0, v0, O1, v1, O2, v2, O3, v3) \
  asm volatile( \
    "s_nop 1\n\t" O0 " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf" \
    : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3))
#define LDSP_DPP_GROUP5(O0, v0, O1, v1, O2, v2, O3, v3, O4, v4) \
  asm volatile( \
    "s_nop 1\n\t" O0 " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_bcast:31 row_mask:0xc bank_mask:0xf" \
    : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4))
#define LDSP_DPP_GROUP6(O0, v0, O1, v1, O2, v2, O3, v3, O4, v4, O5, v5) \
  asm volatile( \
    "s_nop 1\n\t" O0 " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_bcast:31 row_mask:0xc bank_mask:0xf" \
    : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5))
#define LDSP_DPP_GROUP7(O0, v0, O1, v1, O2, v2, O3, v3, O4, v4, O5, v5, O6, v6) \
  asm volatile( \
    "s_nop 1\n\t" O0 " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O6 " %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O6 " %6, %6, %6 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O6 " %6, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O6 " %6, %6, %6 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O6 " %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O6 " %6, %6, %6 row_bcast:31 row_mask:0xc bank_mask:0xf" \
    : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6))
#define LDSP_DPP_GROUP8(O0, v0, O1, v1, O2, v2, O3, v3, O4, v4, O5, v5, O6, v6, O7, v7) \
  asm volatile( \
    "s_nop 1\n\t" O0 " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O6 " %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" O7 " %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O6 " %6, %6, %6 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" O7 " %7, %7, %7 row_shr:2 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O6 " %6, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" O7 " %7, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O6 " %6, %6, %6 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" O7 " %7, %7, %7 row_shr:8 row_mask:0xf bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O6 " %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" O7 " %7, %7, %7 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n\t" \
    O0 " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O1 " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O2 " %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O3 " %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O4 " %4, %4, %4 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O5 " %5, %5, %5 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O6 " %6, %6, %6 row_bcast:31 row_mask:0xc bank_mask:0xf" "\n\t" O7 " %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf" \
    : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7))
