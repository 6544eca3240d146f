// trace_blocks.hpp — per-trace building blocks on an LDS-resident signal of
// arbitrary length n, executed by one workgroup (any multiple of 64 threads).
// They are the generic counterparts of what the fused dsp_icpc kernels specialise:
// the filter-functor kernels (functor_kernels.hip) and the fused dsp_sipm kernel
// (sipm_kernel.hip) are compositions of these.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "ldsp_device.hpp"

namespace ldsp {
namespace tb {

struct Scratch {           // small LDS work area shared by the blocks below
  double d[3 * 16];        // per-wave partials
  unsigned long long u64[2];
  uint32_t hist[256];      // radix-select histogram
  int i[8];
  float f[8];
};

__device__ __forceinline__ int nwaves() { return blockDim.x >> 6; }

// ---- reductions: result in every thread; 2 barriers ---------------------------
__device__ __forceinline__ double blk_sum(double v, Scratch& s) {
  v = wave_incl_scan_sum_f64(v);
  if (lane_id() == 63) s.d[wave_id()] = v;
  __syncthreads();
  double t = 0;
  for (int w = 0; w < nwaves(); ++w) t += s.d[w];
  __syncthreads();
  return t;
}
__device__ __forceinline__ void blk_sum3(double (&v)[3], Scratch& s) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double t = wave_incl_scan_sum_f64(v[k]);
    if (lane_id() == 63) s.d[k * 16 + wave_id()] = t;
  }
  __syncthreads();
  double a = 0, b = 0, c = 0;
  for (int w = 0; w < nwaves(); ++w) { a += s.d[w]; b += s.d[16 + w]; c += s.d[32 + w]; }
  __syncthreads();
  v[0] = a; v[1] = b; v[2] = c;
}
// (value, index) maximum with first-occurrence tie-break; result to every thread
__device__ __forceinline__ void blk_argmax(float v, int idx, Scratch& s, float* vout, int* iout) {
  unsigned long long k = wave_max_u64(pack_vi(v, idx));
  if (threadIdx.x == 0) s.u64[0] = 0ull;
  __syncthreads();
  if (lane_id() == 0) atomicMax(&s.u64[0], k);
  __syncthreads();
  unpack_vi(s.u64[0], vout, iout);
  __syncthreads();
}
__device__ __forceinline__ void blk_argmin(float v, int idx, Scratch& s, float* vout, int* iout) {
  blk_argmax(-v, idx, s, vout, iout);
  *vout = -*vout;
}
__device__ __forceinline__ int blk_sum_i(int v, Scratch& s) {
  v = wave_sum_all_i(v);
  if (threadIdx.x == 0) s.i[0] = 0;
  __syncthreads();
  if (lane_id() == 0 && v) atomicAdd(&s.i[0], v);
  __syncthreads();
  int r = s.i[0];
  __syncthreads();
  return r;
}
__device__ __forceinline__ int blk_min_i(int v, Scratch& s) {
  if (threadIdx.x == 0) s.i[1] = 0x7fffffff;
  __syncthreads();
  atomicMin(&s.i[1], v);
  __syncthreads();
  int r = s.i[1];
  __syncthreads();
  return r;
}
__device__ __forceinline__ int blk_max_i(int v, Scratch& s) {
  if (threadIdx.x == 0) s.i[2] = (int)0x80000000;
  __syncthreads();
  atomicMax(&s.i[2], v);
  __syncthreads();
  int r = s.i[2];
  __syncthreads();
  return r;
}

// ---- staging ---------------------------------------------------------------------
// global [n] -> LDS (coalesced, 16 B/lane when the row is 16-B aligned)
__device__ __forceinline__ void load_trace(const float* __restrict__ g, float* s, int n) {
  const int tid = threadIdx.x, NT = blockDim.x;
  if ((((uintptr_t)g) & 15) == 0) {
    const int n4 = n >> 2;
    for (int i = tid; i < n4; i += NT) reinterpret_cast<float4*>(s)[i] = reinterpret_cast<const float4*>(g)[i];
    for (int i = (n4 << 2) + tid; i < n; i += NT) s[i] = g[i];
  } else {
    for (int i = tid; i < n; i += NT) s[i] = g[i];
  }
}
__device__ __forceinline__ void store_trace(float* __restrict__ g, const float* s, int n) {
  const int tid = threadIdx.x, NT = blockDim.x;
  if ((((uintptr_t)g) & 15) == 0) {
    const int n4 = n >> 2;
    for (int i = tid; i < n4; i += NT) reinterpret_cast<float4*>(g)[i] = reinterpret_cast<const float4*>(s)[i];
    for (int i = (n4 << 2) + tid; i < n; i += NT) g[i] = s[i];
  } else {
    for (int i = tid; i < n; i += NT) g[i] = s[i];
  }
}

// ---- inclusive prefix sum in place: s[i] = sum_{j<=i} s[j]  (S4 rows, DPP scans,
// double offsets).  s must be padded to a multiple of 4 floats (pad = 0).
__device__ __forceinline__ void prefix_sum_inplace(float* s, int n, Scratch& sc) {
  const int tid = threadIdx.x, NT = blockDim.x, l = lane_id(), w = wave_id();
  double carry = 0;
  for (int base = 0; base < n; base += 4 * NT) {
    const int i = base + 4 * tid;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) v = *reinterpret_cast<const float4*>(&s[i]);
    if (i + 1 >= n) v.y = 0.f;
    if (i + 2 >= n) v.z = 0.f;
    if (i + 3 >= n) v.w = 0.f;
    const float t = (v.x + v.y) + (v.z + v.w);
    const float inc = wave_incl_scan_sum(t);
    if (l == 63) sc.d[w] = (double)inc;
    __syncthreads();
    double basev = carry, tot = 0;
    for (int ww = 0; ww < nwaves(); ++ww) {
      if (ww < w) basev += sc.d[ww];
      tot += sc.d[ww];
    }
    carry += tot;
    double run = basev + (double)(inc - t);
    float4 o;
    run += (double)v.x; o.x = (float)run;
    run += (double)v.y; o.y = (float)run;
    run += (double)v.z; o.z = (float)run;
    run += (double)v.w; o.w = (float)run;
    if (i < n) *reinterpret_cast<float4*>(&s[i]) = o;  // pad lanes repeat the last value: harmless
    __syncthreads();
  }
}

// ---- window statistics (signalstats sums; tailstats with the log) ----------------
// returns mean, sigma, slope (per time unit), offset.  LOG: tailstats arithmetic
// (src/tailstats.jl:22-72) incl. the all-zero result when any sample <= 0.
template <bool LOG>
__device__ __forceinline__ bool window_stats(const float* s, int from, int until, float t_first, float dt, Scratch& sc,
                                             float* mean, float* sigma, float* slope, float* offset) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const double ic = 0.5 * ((double)from + (double)until);
  double v[3] = {0, 0, 0};
  int bad = 0;
  for (int i = from + tid; i <= until; i += NT) {
    float y = s[i];
    if (LOG) {
      if (y <= 0.f) { bad = 1; continue; }
      y = logf(y);
    }
    const double d = (double)y;
    v[0] += d;
    v[1] = fma(d, d, v[1]);
    v[2] = fma((double)i - ic, d, v[2]);
  }
  if (LOG) {
    if (blk_sum_i(bad, sc) > 0) { *mean = 0.f; *sigma = 0.f; *slope = 0.f; *offset = 0.f; return false; }
  }
  blk_sum3(v, sc);
  const double n = (double)(until - from + 1), inv_n = 1.0 / n;
  const double m = v[0] * inv_n;
  double var = v[1] * inv_n - m * m;
  if (var < 0) var = 0;
  const double var_i = (n * n - 1.0) / 12.0;
  const double sl_t = (v[2] * inv_n) / var_i / (double)dt;
  const double mean_x = (double)t_first + ic * (double)dt;
  *mean = (float)m; *sigma = (float)sqrt(var); *slope = (float)sl_t; *offset = (float)(m - sl_t * mean_x);
  return true;
}

// ---- extremestats (src/extremestats.jl:25-40): first-occurrence min / max in [from, until]
__device__ __forceinline__ void extreme_stats(const float* s, int from, int until, Scratch& sc, float* vmin, int* imin,
                                              float* vmax, int* imax) {
  const int tid = threadIdx.x, NT = blockDim.x;
  float bmx = -INFINITY, bmn = INFINITY;
  int imx = 0x7fffffff, imn = 0x7fffffff;
  for (int i = from + tid; i <= until; i += NT) {
    const float y = s[i];
    if (y > bmx) { bmx = y; imx = i; }
    if (y < bmn) { bmn = y; imn = i; }
  }
  blk_argmax(bmx, imx, sc, vmax, imax);
  blk_argmin(bmn, imn, sc, vmin, imin);
}

// get_wvf_maximum (src/interpolation.jl:30-46)
__device__ __forceinline__ float window_max_interp(const float* s, int from, int until, Scratch& sc) {
  const int tid = threadIdx.x, NT = blockDim.x;
  float bmx = -INFINITY; int imx = 0x7fffffff;
  for (int i = from + tid; i <= until; i += NT) {
    const float y = s[i];
    if (y > bmx) { bmx = y; imx = i; }
  }
  float v; int i;
  blk_argmax(bmx, imx, sc, &v, &i);
  if (i > from && i < until) v = extrema3points(s[i - 1], s[i], s[i + 1]);
  return v;
}

// ---- order statistics by radix select ----------------------------------------------
// KEY(i) -> (valid, 32-bit ordered key).  Returns the key of rank k (0-based) among
// the valid elements; 4 passes of 8 bits, histogram by LDS atomics.
template <typename KEY>
__device__ __forceinline__ uint32_t radix_select(int n, int k, KEY key, Scratch& sc) {
  const int tid = threadIdx.x, NT = blockDim.x;
  uint32_t prefix = 0, mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int b = tid; b < 256; b += NT) sc.hist[b] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
      bool valid; uint32_t kv;
      key(i, &valid, &kv);
      if (valid && (kv & mask) == prefix) atomicAdd(&sc.hist[(kv >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {  // wave 0: find the bin that contains rank k
      uint32_t c0 = sc.hist[4 * tid], c1 = sc.hist[4 * tid + 1], c2 = sc.hist[4 * tid + 2], c3 = sc.hist[4 * tid + 3];
      int tot = (int)(c0 + c1 + c2 + c3);
      int inc = wave_incl_scan_sum_i(tot);
      int exc = inc - tot;
      if (k >= exc && k < inc) {
        int r = k - exc, b = 4 * tid;
        if (r >= (int)c0) { r -= c0; ++b; if (r >= (int)c1) { r -= c1; ++b; if (r >= (int)c2) { r -= c2; ++b; } } }
        sc.i[4] = b; sc.i[5] = r;
      }
    }
    __syncthreads();
    prefix |= ((uint32_t)sc.i[4]) << shift;
    mask |= 0xffu << shift;
    k = sc.i[5];
    __syncthreads();
  }
  return prefix;
}
// Statistics.median over the valid elements: mean of the two middle order statistics for
// even counts.  m = number of valid elements (> 0).
template <typename KEY>
__device__ __forceinline__ float median_valid(int n, int m, KEY key, Scratch& sc) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int k = (m - 1) >> 1;
  const uint32_t kk = radix_select(n, k, key, sc);
  const float lo = ford_inv(kk);
  if (m & 1) return lo;
  // next order statistic: equal to lo if enough duplicates, else the smallest key above
  int cle = 0;
  uint32_t nxt = 0xffffffffu;
  for (int i = tid; i < n; i += NT) {
    bool valid; uint32_t kv;
    key(i, &valid, &kv);
    if (valid) { cle += (kv <= kk); if (kv > kk) nxt = min(nxt, kv); }
  }
  cle = blk_sum_i(cle, sc);
  const int nx = blk_min_i((int)(nxt ^ 0x80000000u), sc);  // unsigned order via sign flip
  const float hi = (cle >= k + 2) ? lo : ford_inv(((uint32_t)nx) ^ 0x80000000u);
  return 0.5f * (lo + hi);
}
// thresholdstats_mad (src/thresholdstats.jl:61-71): 1.4826 * median(|y - median(y)|) over lo <= y <= hi
__device__ __forceinline__ float mad_threshold(const float* s, int n, float lo, float hi, float scale_sign, Scratch& sc) {
  const int tid = threadIdx.x, NT = blockDim.x;
  int m = 0;
  for (int i = tid; i < n; i += NT) { const float y = scale_sign * s[i]; m += (lo <= y && y <= hi); }
  m = blk_sum_i(m, sc);
  if (m == 0) return 0.f;
  auto k1 = [&](int i, bool* valid, uint32_t* kv) {
    const float y = scale_sign * s[i];
    *valid = (lo <= y && y <= hi);
    *kv = ford(y);
  };
  const float med = median_valid(n, m, k1, sc);
  auto k2 = [&](int i, bool* valid, uint32_t* kv) {
    const float y = scale_sign * s[i];
    *valid = (lo <= y && y <= hi);
    *kv = ford(fabsf(y - med));
  };
  return 1.4826f * median_valid(n, m, k2, sc);
}

// ---- threshold bit-mask + Intersect scans ---------------------------------------------
// bm: ceil(n/32)+2 words.  bit i = (sign*s[i] >= thr)
__device__ __forceinline__ void build_mask(const float* s, int n, float sign, float thr, uint32_t* bm) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int nwords = (n + 31) >> 5;
  const int npad = ((nwords * 32 + 63) >> 6) << 6;
  for (int base = 0; base < npad; base += NT) {
    const int i = base + tid;
    const bool p = (i < n) && (sign * s[i] >= thr);
    const unsigned long long m = __ballot(p);
    if (lane_id() == 0) {
      const int wb = i >> 5;
      if (wb < nwords + 2) bm[wb] = (uint32_t)m;
      if (wb + 1 < nwords + 2) bm[wb + 1] = (uint32_t)(m >> 32);
    }
  }
}
// Intersect(min_n): first confirmed crossing position (or -1) and multiplicity
__device__ __forceinline__ void intersect_scan(const uint32_t* bm, int n, int min_n, Scratch& sc, int* first, int* count) {
  const int nwords = (n + 31) >> 5;
  int c = 0, f = 0x7fffffff;
  for (int w = threadIdx.x; w < nwords; w += blockDim.x) {
    int cc, ff;
    intersect_word(bm, w, nwords, min_n, &cc, &ff);
    c += cc; f = min(f, ff);
  }
  *count = blk_sum_i(c, sc);
  *first = blk_min_i(f, sc);
}

// one-pass y = scale * x (+ in place variants are trivial loops in the callers)

}  // namespace tb
}  // namespace ldsp
