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
  uint32_t hist[2048];     // radix-select histogram (11-bit digits)
  int i[8];
  float f[8];
  int wtot[16];            // per-wave totals of block scans
  uint32_t u32min;
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
// the same from uint16 ADC counts (converted on the way)
__device__ __forceinline__ void load_trace_u16(const uint16_t* __restrict__ g, float* s, int n) {
  const int tid = threadIdx.x, NT = blockDim.x;
  for (int i = tid; i < n; i += NT) s[i] = (float)g[i];
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

// ---- order statistics ----------------------------------------------------------------
// Exact selection (rank k of the valid elements, plus rank k+1 for even counts) in two levels:
//   level 1: ONE histogram pass over the trace that isolates a small candidate set, written to
//            an LDS list — by linear bins around (mean +- 4 sigma) when the caller has those
//            (median_linear: a noise trace then spreads over ~2000 bins and the LDS atomics are
//            nearly conflict-free; measured 4.6 clk per wave-atomic conflict-free vs 30 clk at
//            8 lanes per address, tools/micro/lds_atomic_test.hip), or by the top 11 bits of the
//            ordered key (radix_select: always applicable);
//   level 2: three radix passes (11 + 11 + 10 bits) over the candidate list only.
// Histogram bin location: every thread owns nbins/NT consecutive bins, block-wide scan.

// after a histogram of `nbins` bins sits in sc.hist: bin holding rank k -> sc.i[4], rank inside
// it -> sc.i[5], its population -> sc.i[6].  Barrier before (hist complete) is the caller's.
__device__ __forceinline__ void locate_bin(const uint32_t* hist, int nbins, int k, Scratch& sc) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int per = (nbins + NT - 1) / NT, b0 = per * tid;
  int tot = 0;
  for (int q = 0; q < per; ++q) if (b0 + q < nbins) tot += (int)hist[b0 + q];
  const int inc = wave_incl_scan_sum_i(tot);
  if ((tid & 63) == 63) sc.wtot[tid >> 6] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < (tid >> 6); ++w) base += sc.wtot[w];
  const int exc = base + inc - tot;
  if (tot > 0 && k >= exc && k < exc + tot) {   // exactly one thread
    int r = k - exc, b = b0;
    while (r >= (int)hist[b]) { r -= (int)hist[b]; ++b; }
    sc.i[4] = b; sc.i[5] = r; sc.i[6] = (int)hist[b];
  }
  __syncthreads();
}
// One histogram pass over `cnt` items: GET(j) -> (valid, 32-bit ordered key).  Among the valid
// keys with (key & mask) == prefix, finds the `nbins`-way digit (key >> shift) & (nbins-1) that
// holds rank k; k becomes the rank inside that digit, cnt_in the digit's population.
template <typename GET>
__device__ __forceinline__ void radix_pass(int cnt, GET get, uint32_t mask, uint32_t prefix, int shift, int nbins, int* k,
                                           uint32_t* digit, int* cnt_in, Scratch& sc) {
  const int tid = threadIdx.x, NT = blockDim.x;
  for (int b = tid; b < nbins; b += NT) sc.hist[b] = 0;
  __syncthreads();
  for (int j = tid; j < cnt; j += NT) {
    bool valid; uint32_t kv;
    get(j, &valid, &kv);
    if (valid && (kv & mask) == prefix) atomicAdd(&sc.hist[(kv >> shift) & (uint32_t)(nbins - 1)], 1u);
  }
  __syncthreads();
  locate_bin(sc.hist, nbins, *k, sc);
  *digit = (uint32_t)sc.i[4]; *k = sc.i[5]; *cnt_in = sc.i[6];
  __syncthreads();
}
// append the keys selected by PRED(i) -> (take, key) for i < n to list[] (order irrelevant):
// wave-aggregated, one LDS atomic per wave and row.  The count is known to the caller.
template <typename PRED>
__device__ __forceinline__ void compact_keys(int n, PRED pred, uint32_t* list, Scratch& sc) {
  const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63;
  if (tid == 0) sc.i[7] = 0;
  __syncthreads();
  for (int base = 0; base < n; base += NT) {
    const int i = base + tid;
    bool take = false; uint32_t kv = 0;
    if (i < n) pred(i, &take, &kv);
    const unsigned long long m = __ballot(take);
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      int pos = 0;
      if (lane == leader) pos = atomicAdd(&sc.i[7], __popcll(m));
      pos = __builtin_amdgcn_readlane(pos, leader);
      if (take) list[pos + __popcll(m & ((1ull << lane) - 1ull))] = kv;
    }
  }
  __syncthreads();
}
// rank k of list[0..cnt) (all 32 key bits); *has_next / *next: the key of rank k+1 if it is in the list
__device__ __forceinline__ uint32_t select_in_list(const uint32_t* list, int cnt, int k, Scratch& sc, bool want_next, bool* has_next,
                                                   uint32_t* next) {
  const int tid = threadIdx.x, NT = blockDim.x;
  auto kc = [&](int j, bool* valid, uint32_t* kv) { *valid = true; *kv = list[j]; };
  uint32_t d1, d2, d3;
  int c1, c2, c3;
  const bool last_of_list = (k == cnt - 1);
  radix_pass(cnt, kc, 0u, 0u, 21, 2048, &k, &d1, &c1, sc);
  radix_pass(cnt, kc, 0x7ffu << 21, d1 << 21, 10, 2048, &k, &d2, &c2, sc);
  radix_pass(cnt, kc, 0xfffffc00u, (d1 << 21) | (d2 << 10), 0, 1024, &k, &d3, &c3, sc);
  const uint32_t kk = (d1 << 21) | (d2 << 10) | d3;
  if (want_next) {
    if (k + 1 < c3) {            // c3 keys equal kk, k = rank among them
      *has_next = true; *next = kk;
    } else if (last_of_list) {
      *has_next = false;
    } else {                     // the smallest key of the list above kk
      if (tid == 0) sc.u32min = 0xffffffffu;
      __syncthreads();
      uint32_t mn = 0xffffffffu;
      for (int j = tid; j < cnt; j += NT) { const uint32_t kv = list[j]; if (kv > kk) mn = min(mn, kv); }
      if (mn != 0xffffffffu) atomicMin(&sc.u32min, mn);
      __syncthreads();
      *has_next = true; *next = sc.u32min;
      __syncthreads();
    }
  }
  return kk;
}
// the smallest valid key above kk over the whole trace (rank k+1 when it falls outside the candidate list)
template <typename KEY>
__device__ __forceinline__ uint32_t next_key_above(int n, KEY key, uint32_t kk, Scratch& sc) {
  const int tid = threadIdx.x, NT = blockDim.x;
  if (tid == 0) sc.u32min = 0xffffffffu;
  __syncthreads();
  uint32_t mn = 0xffffffffu;
  for (int i = tid; i < n; i += NT) {
    bool valid; uint32_t kv;
    key(i, &valid, &kv);
    if (valid && kv > kk) mn = min(mn, kv);
  }
  if (mn != 0xffffffffu) atomicMin(&sc.u32min, mn);
  __syncthreads();
  const uint32_t r = sc.u32min;
  __syncthreads();
  return r;
}
// KEY(i) -> (valid, 32-bit ordered key) for i < n.  Returns the key of rank k (0-based) among the
// valid elements; `next` (optional) receives the key of rank k+1 (the caller guarantees it exists).
// `list`: n words of LDS that are free at the call, or nullptr (then three passes over the trace).
template <typename KEY>
__device__ __forceinline__ uint32_t radix_select(int n, int k, KEY key, Scratch& sc, uint32_t* list, uint32_t* next = nullptr) {
  uint32_t d1, d2, d3;
  int c1, c2, c3;
  radix_pass(n, key, 0u, 0u, 21, 2048, &k, &d1, &c1, sc);
  uint32_t kk;
  bool has_next = false;
  if (list) {
    compact_keys(n, [&](int i, bool* take, uint32_t* kv) { bool v; key(i, &v, kv); *take = v && (*kv >> 21) == d1; }, list, sc);
    kk = select_in_list(list, c1, k, sc, next != nullptr, &has_next, next);
  } else {
    radix_pass(n, key, 0x7ffu << 21, d1 << 21, 10, 2048, &k, &d2, &c2, sc);
    radix_pass(n, key, 0xfffffc00u, (d1 << 21) | (d2 << 10), 0, 1024, &k, &d3, &c3, sc);
    kk = (d1 << 21) | (d2 << 10) | d3;
    if (next && k + 1 < c3) { *next = kk; has_next = true; }
  }
  if (next && !has_next) *next = next_key_above(n, key, kk, sc);
  return kk;
}
// Statistics.median over the valid elements: mean of the two middle order statistics for
// even counts.  m = number of valid elements (> 0).
template <typename KEY>
__device__ __forceinline__ float median_valid(int n, int m, KEY key, Scratch& sc, uint32_t* list) {
  const int k = (m - 1) >> 1;
  uint32_t nxt = 0;
  const uint32_t kk = radix_select(n, k, key, sc, list, (m & 1) ? nullptr : &nxt);
  const float lo = ford_inv(kk);
  if (m & 1) return lo;
  return 0.5f * (lo + ford_inv(nxt));
}
// rank r (and r+1) among list[0..cnt), cnt <= 64, by every wave on its own: lane j holds key j and
// counts the keys that sort before it (ties by position); no LDS writes, no barrier.
__device__ __forceinline__ uint32_t select_small(const uint32_t* list, int cnt, int r, bool* has_next, uint32_t* next) {
  const int lane = threadIdx.x & 63;
  const uint32_t mine = (lane < cnt) ? list[lane] : 0xffffffffu;
  int rank = 0;
  for (int i = 0; i < cnt; ++i) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)mine, i);
    rank += (o < mine) || (o == mine && i < lane);
  }
  const unsigned long long mr = __ballot(lane < cnt && rank == r), mn = __ballot(lane < cnt && rank == r + 1);
  const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)mine, __ffsll((long long)mr) - 1);
  *has_next = mn != 0ull;
  if (mn) *next = (uint32_t)__builtin_amdgcn_readlane((int)mine, __ffsll((long long)mn) - 1);
  return kk;
}
// The same median with a linear first level: VAL(i) -> (valid, value); NB bins of width 1/scale from `a`
// (NB = 2048 .. 8192, as many as the work area allows: a trace of noise then leaves a handful of
// candidates in the median's bin and the second level runs inside a wave).  work: nwork words of
// free LDS, split into histogram and candidate list.  Returns false (nothing computed) when a needed
// rank falls outside the binned range — the caller then uses median_valid.  Binning is monotone in
// the value, so the selection stays exact.
template <typename VAL>
__device__ __forceinline__ bool median_linear(int n, int m, VAL val, float a, float sd_span, Scratch& sc, uint32_t* work, int nwork,
                                              float* med) {
  const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63;
  const int NB = (nwork >= 16384) ? 8192 : (nwork >= 8192 ? 4096 : 2048);
  uint32_t* hist = (nwork >= 4096) ? work : sc.hist;
  uint32_t* list = (nwork >= 4096) ? work + NB : work;
  const float scale = (float)NB / sd_span;
  for (int b = tid; b < NB; b += NT) hist[b] = 0;
  if (tid == 0) { sc.i[2] = 0; sc.i[3] = 0; }
  __syncthreads();
  auto bin_of = [&](float v) -> int {   // -1 below, NB above (also NaN)
    const float t = (v - a) * scale;
    return (t < 0.f) ? -1 : (t < (float)NB ? (int)t : NB);
  };
  for (int base = 0; base < n; base += NT) {
    const int i = base + tid;
    bool valid = false; float v = 0.f;
    if (i < n) val(i, &valid, &v);
    const int b = valid ? bin_of(v) : 0;
    if (valid && b >= 0 && b < NB) atomicAdd(&hist[b], 1u);
    const unsigned long long mu = __ballot(valid && b < 0), mo = __ballot(valid && b >= NB);
    if (lane == 0) {
      if (mu) atomicAdd(&sc.i[2], __popcll(mu));
      if (mo) atomicAdd(&sc.i[3], __popcll(mo));
    }
  }
  __syncthreads();
  const int k = (m - 1) >> 1, need = (m & 1) ? 0 : 1;
  const int cu = sc.i[2], co = sc.i[3];
  if (k < cu || k + need >= m - co) return false;   // block-uniform
  locate_bin(hist, NB, k - cu, sc);
  const int bsel = sc.i[4], r = sc.i[5], cb = sc.i[6];
  __syncthreads();
  if (cb > ((nwork >= 4096) ? nwork - NB : nwork)) return false;   // (block-uniform) heavily clustered values: the list would not fit
  compact_keys(n, [&](int i, bool* take, uint32_t* kv) { bool v; float x; val(i, &v, &x); *take = v && bin_of(x) == bsel; *kv = ford(x); }, list, sc);
  bool has_next = false;
  uint32_t nxt = 0, kk;
  if (cb <= 64) kk = select_small(list, cb, r, &has_next, &nxt);
  else kk = select_in_list(list, cb, r, sc, need != 0, &has_next, &nxt);
  const float lo = ford_inv(kk);
  if (!need) { *med = lo; return true; }
  if (!has_next) nxt = next_key_above(n, [&](int i, bool* v, uint32_t* kv) { float x; val(i, v, &x); *kv = ford(x); }, kk, sc);
  *med = 0.5f * (lo + ford_inv(nxt));
  return true;
}
// thresholdstats_mad (src/thresholdstats.jl:61-71): 1.4826 * median(|y - median(y)|) over lo <= y <= hi.
// work: nwork >= n words of free LDS (or nullptr: three radix passes over the trace per median)
__device__ __forceinline__ float mad_threshold(const float* s, int n, float lo, float hi, float scale_sign, Scratch& sc,
                                               uint32_t* work = nullptr, int nwork = 0) {
  uint32_t* list = work;
  const int tid = threadIdx.x, NT = blockDim.x;
  // count, mean and spread of the valid samples (the spread only steers the binning)
  double v[3] = {0, 0, 0};
  for (int i = tid; i < n; i += NT) {
    const float y = scale_sign * s[i];
    if (lo <= y && y <= hi) { v[0] += 1.0; v[1] += (double)y; v[2] = fma((double)y, (double)y, v[2]); }
  }
  blk_sum3(v, sc);
  const int m = (int)v[0];
  if (m == 0) return 0.f;
  const float mean = (float)(v[1] / v[0]);
  const double var = v[2] / v[0] - (v[1] / v[0]) * (v[1] / v[0]);
  const float sd = var > 0 ? (float)sqrt(var) : 0.f;
  auto v1 = [&](int i, bool* valid, float* x) { const float y = scale_sign * s[i]; *valid = (lo <= y && y <= hi); *x = y; };
  auto k1 = [&](int i, bool* valid, uint32_t* kv) { float x; v1(i, valid, &x); *kv = ford(x); };
  float med;
  const bool lin = list != nullptr && sd > 0.f && isfinite(sd) && isfinite(mean);
  if (!lin || !median_linear(n, m, v1, mean - 4.f * sd, 8.f * sd, sc, work, nwork, &med)) med = median_valid(n, m, k1, sc, list);
  auto v2 = [&](int i, bool* valid, float* x) { const float y = scale_sign * s[i]; *valid = (lo <= y && y <= hi); *x = fabsf(y - med); };
  auto k2 = [&](int i, bool* valid, uint32_t* kv) { float x; v2(i, valid, &x); *kv = ford(x); };
  float mad;
  if (!lin || !median_linear(n, m, v2, 0.f, 4.f * sd, sc, work, nwork, &mad)) mad = median_valid(n, m, k2, sc, list);
  return 1.4826f * mad;
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
