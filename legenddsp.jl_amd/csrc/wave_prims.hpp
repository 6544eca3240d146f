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

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ float wave_incl_scan_sum(float v) {
  v += dpp_f<0x111>(0.f, v);
  v += dpp_f<0x112>(0.f, v);
  v += dpp_f<0x114>(0.f, v);
  v += dpp_f<0x118>(0.f, v);
  v += dpp_f<0x142, 0xa>(0.f, v);
  v += dpp_f<0x143, 0xc>(0.f, v);
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
  const float ninf = -__builtin_inff();
  v = fmaxf(v, dpp_f<0x111>(ninf, v));
  v = fmaxf(v, dpp_f<0x112>(ninf, v));
  v = fmaxf(v, dpp_f<0x114>(ninf, v));
  v = fmaxf(v, dpp_f<0x118>(ninf, v));
  v = fmaxf(v, dpp_f<0x142, 0xa>(ninf, v));
  v = fmaxf(v, dpp_f<0x143, 0xc>(ninf, v));
  return readlane_f(v, 63);
}
__device__ __forceinline__ float wave_min_all(float v) {
  const float pinf = __builtin_inff();
  v = fminf(v, dpp_f<0x111>(pinf, v));
  v = fminf(v, dpp_f<0x112>(pinf, v));
  v = fminf(v, dpp_f<0x114>(pinf, v));
  v = fminf(v, dpp_f<0x118>(pinf, v));
  v = fminf(v, dpp_f<0x142, 0xa>(pinf, v));
  v = fminf(v, dpp_f<0x143, 0xc>(pinf, v));
  return readlane_f(v, 63);
}

// Inclusive scan of the recurrence s_l = v_l + a*s_{l-1} over the lanes (one-pole
// IIR with constant per-lane decay a): s_l = sum_{j<=l} a^(l-j) v_j.
// The caller supplies exact powers of a (derived in double on the host):
// p[0..3] = a^1, a^2, a^4, a^8 (the decay of each doubling step inside a 16-lane
// row), f15 = a^((lane&15)+1), f31 = a^((lane&31)+1) (row-total broadcasts).
struct AffinePow { float p1, p2, p4, p8; };
__device__ __forceinline__ float wave_incl_scan_affine(float v, const AffinePow& P, float f15, float f31) {
  v = fmaf(P.p1, dpp_f<0x111>(0.f, v), v);
  v = fmaf(P.p2, dpp_f<0x112>(0.f, v), v);
  v = fmaf(P.p4, dpp_f<0x114>(0.f, v), v);
  v = fmaf(P.p8, dpp_f<0x118>(0.f, v), v);
  v = fmaf(f15, dpp_f<0x142, 0xa>(0.f, v), v);
  v = fmaf(f31, dpp_f<0x143, 0xc>(0.f, v), v);
  return v;
}
// mirror image: s_l = v_l + a*s_{l+1}  (anti-causal), via ds_bpermute shuffles
// pw[s] = a^(2^s), s = 0..5
__device__ __forceinline__ float wave_incl_scan_affine_rev(float v, const float (&pw)[6]) {
  const int l = threadIdx.x & 63;
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int o = 1 << s;
    float t = __shfl_down(v, o, 64);
    if (l + o < 64) v = fmaf(pw[s], t, v);
  }
  return v;
}

}  // namespace ldsp
