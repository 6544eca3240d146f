// ldsp_device.hpp — device-side building blocks shared by the gfx950 trace kernels.
//
// One trace per workgroup of NT threads.  Two views of a trace:
//   * S4 ("striped by 4"): thread t holds, for each row r < R, the four
//     consecutive samples 4*(t + NT*r) .. +3 — exactly what a coalesced
//     global_load_dwordx4 delivers, and what a conflict-free ds_write_b128 into a
//     LINEAR LDS array wants.  Prefix sums and one-pole recursions run in this
//     view: 4-sample serial part, DPP scan across the 64 lanes of a wave, then
//     one [R][NW] table of wave-row partials in LDS.
//   * LS ("lane strided"): thread t visits samples t, t+NT, t+2NT, ...  Every
//     shifted read T[k+s] of a trapezoid / FIR window is then a ds_read_b32 at a
//     per-shift base register plus an IMMEDIATE row offset (no address VALU) and
//     consecutive lanes hit consecutive banks whatever the shift.
// Threshold decisions live as bit arrays in LDS, filled by wave ballots in the LS
// view; Intersect-style run scans work on those words.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wave_prims.hpp"
#include "run_scan.hpp"

#ifndef LDSP_AFFINE_GROUPED   // 1: the four chains of an affine scan in one asm statement (no s_nop per step; in icpc_lean3_kernel the statement's
#define LDSP_AFFINE_GROUPED 0  // seven simultaneous registers push six values into scratch: measured before it is made the default)
#endif
#ifdef LDSP_WHATIF_NOBAR_CZ   // (timing experiment, icpc_lean3.hip)
#define LDSP_BAR_SCAN() ((void)0)
#else
#define LDSP_BAR_SCAN() __syncthreads()
#endif
namespace ldsp {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- S4 block scans -------------------------------------------------------------
// Exclusive prefix (in double) of per-chunk totals, chunk order = t + NT*r.
// tot[r]: total of this thread's 4-sample chunk in row r.  off[r]: sum of all
// chunks before it.  part: LDS scratch [R*NW] doubles (caller alternates buffers).
// The R*NW (<= 64) wave-row partials are combined by a second DPP scan that every
// wave performs redundantly on lanes 0..R*NW-1 — no serial loop, no extra barrier.
template <int NT, int R, typename T>
__device__ __forceinline__ void s4_exscan_sum(const T (&tot)[R], double (&off)[R], double* part, double* total) {
  constexpr int NW = NT / 64;
  static_assert(R * NW <= 64, "wave-row partials must fit one wave");
  const int w = wave_id(), l = lane_id();
  double inc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if constexpr (sizeof(T) == 8) inc[r] = wave_incl_scan_sum_f64((double)tot[r]);
    else inc[r] = (double)wave_incl_scan_sum((float)tot[r]);
  }
  if (l == 63) {
#pragma unroll
    for (int r = 0; r < R; ++r) part[r * NW + w] = inc[r];
  }
  __syncthreads();
  const double pv = (l < R * NW) ? part[l] : 0.0;
  const double pinc = wave_incl_scan_sum_f64(pv);
  const double pexc = pinc - pv;
#pragma unroll
  for (int r = 0; r < R; ++r) off[r] = readlane_d(pexc, r * NW + w) + (inc[r] - (double)tot[r]);
  if (total) *total = readlane_d(pinc, R * NW - 1);
}

// One-pole recursion s <- a*s + b over chunks (a = q^4 constant): state ENTERING
// each of the thread's chunks, forward direction (state 0 before chunk 0).
// b[r]: the chunk's own end state from zero input state.
// qp4[j] = q^(4j), j = 0..64;  qpw[j] = q^(256j), j = 0..64 (one wave-row = 256 samples).
// (w: the wave index; a caller that holds it in a scalar register passes it — as a lane-select operand a vector value costs a
// v_readfirstlane and four wait states per use)
template <int NT, int R>
__device__ __forceinline__ void s4_exscan_affine_fwd(const float (&b)[R], float (&s_in)[R], const float* qp4, const float* qpw,
                                                     float* part, int w = wave_id()) {
  constexpr int NW = NT / 64;
  const int l = lane_id();
  const AffinePow P = {qp4[1], qp4[2], qp4[4], qp4[8]};
  const float f15 = qp4[(l & 15) + 1], f31 = qp4[(l & 31) + 1], fl = qp4[l];
  float inc[R];
  if constexpr (R == 4 && LDSP_AFFINE_GROUPED) {
#pragma unroll
    for (int r = 0; r < R; ++r) inc[r] = b[r];
    wave_incl_scan_affine4(inc, P, f15, f31);
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) inc[r] = wave_incl_scan_affine(b[r], P, f15, f31);
  }
  if (l == 63) {
#pragma unroll
    for (int r = 0; r < R; ++r) part[r * NW + w] = inc[r];
  }
  LDSP_BAR_SCAN();
  // scan of the wave-row end states with decay q^256 per step
  const AffinePow PW = {qpw[1], qpw[2], qpw[4], qpw[8]};
  const float pv = (l < R * NW) ? part[l] : 0.f;
  const float pinc = wave_incl_scan_affine(pv, PW, qpw[(l & 15) + 1], qpw[(l & 31) + 1]);
  const float pexc = dpp_f<0x138>(0.f, pinc);  // state at the END of the previous wave-row = state entering this one
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float sw_in = readlane_f(pexc, r * NW + w);
    const float ex = dpp_f<0x138>(0.f, inc[r]);  // wave_shr:1 — inclusive value of lane l-1 (0 for lane 0)
    s_in[r] = fmaf(fl, sw_in, ex);
  }
}
// Anti-causal mirror: state entering each chunk from the RIGHT (state 0 after the last chunk).
template <int NT, int R>
__device__ __forceinline__ void s4_exscan_affine_bwd(const float (&b)[R], float (&s_in)[R], const float* qp4, const float* qpw,
                                                     float* part, int w = wave_id()) {
  constexpr int NW = NT / 64;
  const int l = lane_id();
  const AffinePow P = {qp4[1], qp4[2], qp4[4], qp4[8]};
  const float fl = qp4[63 - l], frow = qp4[16 - (l & 15)], a16 = qp4[16];
  float inc[R];
  if constexpr (R == 4 && LDSP_AFFINE_GROUPED) {
#pragma unroll
    for (int r = 0; r < R; ++r) inc[r] = b[r];
    wave_incl_scan_affine_rev4(inc, P, a16, frow);
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) inc[r] = wave_incl_scan_affine_rev(b[r], P, a16, frow);
  }
  if (l == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) part[r * NW + w] = inc[r];
  }
  LDSP_BAR_SCAN();
  // mirrored order: lane j holds the wave-row (R*NW-1-j), scanned forward with decay q^256
  constexpr int NP = R * NW;
  const AffinePow PW = {qpw[1], qpw[2], qpw[4], qpw[8]};
  const float pv = (l < NP) ? part[NP - 1 - l] : 0.f;
  const float pinc = wave_incl_scan_affine(pv, PW, qpw[(l & 15) + 1], qpw[(l & 31) + 1]);
  const float pexc = dpp_f<0x138>(0.f, pinc);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float sw_in = readlane_f(pexc, NP - 1 - (r * NW + w));
    const float ex = wave_shl1(inc[r]);  // inclusive value of lane l+1 (0 for lane 63)
    s_in[r] = fmaf(fl, sw_in, ex);
  }
}

// The thread's R quads of one trace (S4 view).  FULL (L == 4*NT*R): R back-to-back 16-byte
// loads with nothing between them; otherwise the address is clamped so that the loads are
// still unconditional (all in flight together) and out-of-range quads are zeroed afterwards.
template <int NT, int R, bool FULL>
__device__ __forceinline__ void load_trace_s4(const float* __restrict__ w, int L, int tid, float (&x)[R][4]) {
  if (FULL || (L & 3) == 0) {
    float4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = 4 * (tid + NT * r);
      v[r] = *reinterpret_cast<const float4*>(w + (FULL ? i : max(min(i, L - 4), 0)));
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool ok = FULL || 4 * (tid + NT * r) < L;
      x[r][0] = ok ? v[r].x : 0.f; x[r][1] = ok ? v[r].y : 0.f; x[r][2] = ok ? v[r].z : 0.f; x[r][3] = ok ? v[r].w : 0.f;
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = 4 * (tid + NT * r);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[r][e] = w[min(i + e, L - 1)];
#pragma unroll
      for (int e = 0; e < 4; ++e) x[r][e] = (i + e < L) ? x[r][e] : 0.f;
    }
  }
}

// the same from uint16 ADC counts (production data; ldsp_icpc_opts.in_u16): 8-byte loads, converted in registers — no separate
// cast pass over HBM (2L + 4L bytes per trace that the float path of a uint16 source pays before the first kernel)
template <int NT, int R, bool FULL>
__device__ __forceinline__ void load_trace_s4_u16(const uint16_t* __restrict__ w, int L, int tid, float (&x)[R][4]) {
  if (FULL || (L & 3) == 0) {
    uint2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = 4 * (tid + NT * r);
      v[r] = *reinterpret_cast<const uint2*>(w + (FULL ? i : max(min(i, L - 4), 0)));
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool ok = FULL || 4 * (tid + NT * r) < L;
      x[r][0] = ok ? (float)(v[r].x & 0xffffu) : 0.f; x[r][1] = ok ? (float)(v[r].x >> 16) : 0.f;
      x[r][2] = ok ? (float)(v[r].y & 0xffffu) : 0.f; x[r][3] = ok ? (float)(v[r].y >> 16) : 0.f;
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = 4 * (tid + NT * r);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[r][e] = (i + e < L) ? (float)w[min(i + e, L - 1)] : 0.f;
    }
  }
}


// ---- bit-mask helpers ----------------------------------------------------------
__device__ __forceinline__ void ballot_store(bool pred, uint32_t* bm, int word_base) {
  unsigned long long m = __ballot(pred);
  if (lane_id() == 0) { bm[word_base] = (uint32_t)m; bm[word_base + 1] = (uint32_t)(m >> 32); }
}

// parabola vertex through three points — reference src/interpolation.jl:8-10
__device__ __forceinline__ float extrema3points(float y1, float y2, float y3) {
  float a = y3 - 4.f * y2 + 3.f * y1;
  return y1 - a * a / (8.f * (y3 - 2.f * y2 + y1));
}

// (value, index) packed so that an unsigned 64-bit max picks the largest value and,
// among equal values, the SMALLEST index (findmax: first occurrence).
// order-preserving map float <-> uint32 (a < b  <=>  ford(a) < ford(b)); lets float
// max/min reductions run on the LDS integer atomics (ds_max_u32 / ds_min_u32)
__device__ __forceinline__ uint32_t ford(float v) {
  uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ford_inv(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ unsigned long long pack_vi(float v, int i) {
  return ((unsigned long long)ford(v) << 32) | (uint32_t)(0x7fffffff - i);
}
__device__ __forceinline__ void unpack_vi(unsigned long long k, float* v, int* i) {
  *v = ford_inv((uint32_t)(k >> 32));
  *i = 0x7fffffff - (int)(uint32_t)(k & 0xffffffffu);
}
// 64-bit unsigned max over the wave (DPP scan, result in every lane)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, false);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
  unsigned long long t;
  t = dpp_u64<0x111>(k); k = t > k ? t : k;
  t = dpp_u64<0x112>(k); k = t > k ? t : k;
  t = dpp_u64<0x114>(k); k = t > k ? t : k;
  t = dpp_u64<0x118>(k); k = t > k ? t : k;
  t = dpp_u64<0x142, 0xa>(k); k = t > k ? t : k;
  t = dpp_u64<0x143, 0xc>(k); k = t > k ? t : k;
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

}  // namespace ldsp
