// functor_kernels.hip — the per-functor entry points of include/ldsp.h:
// rdfilt!-style batched filters and the feature extractors, one workgroup per
// trace, the trace staged in LDS (trace_blocks.hpp).  These are the operator
// boundary of the reference (filter functors + callable extractors); the
// production hot path is the fused kernels (icpc_kernel.hip, sipm_kernel.hip).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <vector>

#include "host_math.hpp"
#include "ldsp_ctx.hpp"
#include "trace_blocks.hpp"

namespace ldsp {
namespace fk {

using tb::Scratch;

__device__ __forceinline__ int pad4(int n) { return ((n + 3) & ~3) + 64; }

struct Lds {
  float* s;      // [pad4(n)]
  float* t;      // second array (only where allocated)
  uint32_t* bm;  // [(n+31)/32 + 2]
  Scratch* sc;
};
// layout: s | t (optional) | bm | scratch
__device__ __forceinline__ Lds carve(unsigned char* raw, int n, bool two) {
  Lds l;
  l.s = reinterpret_cast<float*>(raw);
  l.t = l.s + pad4(n);
  float* after = two ? l.t + pad4(n) : l.t;
  l.bm = reinterpret_cast<uint32_t*>(after);
  const int nw = ((n + 31) >> 5) + 2;
  // offsets as integers, pointers derived from `raw`: a round trip through uintptr_t would turn every access
  // to the scratch area into a FLAT instruction (the compiler loses the LDS address space)
  const size_t off = ((size_t)(reinterpret_cast<unsigned char*>(l.bm + nw) - raw) + 15) & ~(size_t)15;
  l.sc = reinterpret_cast<Scratch*>(raw + off);
  return l;
}
static size_t lds_bytes(int n, bool two) {
  size_t p4 = (size_t)(((n + 3) & ~3) + 64);
  return p4 * 4 * (two ? 2 : 1) + (size_t)(((n + 31) >> 5) + 2) * 4 + 16 + sizeof(Scratch);
}

// ---------------------------------------------------------------- scan filters
// MODE 0 InvCRFilter: y = x + c*cumsum(x)            
// MODE 1 IntegratorFilter: y = g*cumsum(x)           
// MODE 2 MovingWindowFilter(len)                     (src/moving_window_multi.jl:99-116)
// MODE 3 MovingWindowMultiFilter(len)                (src/moving_window_multi.jl:118-129)
__device__ __forceinline__ float mw_value(const float* S, int i, int l, float x0, float invl) {
  // mean of x~[i-l+1 .. i], x~_j = x_0 for j < 0  (closed form of the reference's recursion)
  const float lo = (i - l >= 0) ? S[i - l] : 0.f;
  const float padn = (float)max(l - 1 - i, 0);
  return (S[i] - lo + padn * x0) * invl;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_scan_filter(const float* __restrict__ x, int L, float p, int len, float* __restrict__ y) {
  extern __shared__ __align__(16) unsigned char raw[];
  Lds l = carve(raw, L, MODE == 3);
  const float* xr = x + (size_t)blockIdx.x * L;
  float* yr = y + (size_t)blockIdx.x * L;
  const int tid = threadIdx.x, NT = blockDim.x;
  tb::load_trace(xr, l.s, L);
  for (int i = L + tid; i < pad4(L); i += NT) l.s[i] = 0.f;
  __syncthreads();
  if (MODE == 0 || MODE == 1) {
    tb::prefix_sum_inplace(l.s, L, *l.sc);
    for (int i = tid; i < L; i += NT) yr[i] = (MODE == 0) ? xr[i] + p * l.s[i] : p * l.s[i];
  } else {
    const float invl = 1.f / (float)len;
    const int passes = (MODE == 2) ? 1 : 3;
    for (int ps = 0; ps < passes; ++ps) {
      const float x0 = l.s[0];
      __syncthreads();
      tb::prefix_sum_inplace(l.s, L, *l.sc);
      if (ps == passes - 1) {
        if (MODE == 2) {
          for (int i = tid; i < L; i += NT) yr[i] = mw_value(l.s, i, len, x0, invl);
        } else {
          for (int i = tid; i < L; i += NT) yr[i] = mw_value(l.s, i, len, x0, invl);
        }
      } else {
        for (int i = tid; i < L; i += NT) l.t[i] = mw_value(l.s, i, len, x0, invl);
        __syncthreads();
        // pass 0 -> 1 reverses; pass 1 -> 2 reverses back
        for (int i = tid; i < L; i += NT) l.s[i] = l.t[L - 1 - i];
        for (int i = L + tid; i < pad4(L); i += NT) l.s[i] = 0.f;
        __syncthreads();
      }
    }
  }
}

// TrapezoidalChargeFilter: out[k] = mean(x[k+n1+g .. +n2-1]) - mean(x[k .. k+n1-1])
__global__ void __launch_bounds__(256) k_trap(const float* __restrict__ x, int L, ldsp_trap t, float* __restrict__ y) {
  extern __shared__ __align__(16) unsigned char raw[];
  Lds l = carve(raw, L, false);
  const int flen = t.navg + t.ngap + t.navg2, nout = L - flen + 1;
  const float* xr = x + (size_t)blockIdx.x * L;
  float* yr = y + (size_t)blockIdx.x * nout;
  const int tid = threadIdx.x, NT = blockDim.x;
  tb::load_trace(xr, l.s, L);
  for (int i = L + tid; i < pad4(L); i += NT) l.s[i] = 0.f;
  __syncthreads();
  tb::prefix_sum_inplace(l.s, L, *l.sc);
  const float i1 = 1.f / (float)t.navg, i2 = 1.f / (float)t.navg2;
  // A window sum taken as a difference of float prefix sums carries ulp(S) of absolute error;
  // divided by a SHORT window that is too much (the 2-sample leg of the t0 filter), so short
  // windows are summed directly from the samples.
  constexpr int SHORT = 16;
  for (int k = tid; k < nout; k += NT) {
    float a, b;
    if (t.navg2 <= SHORT) {
      a = 0.f;
      for (int j = 0; j < t.navg2; ++j) a += xr[k + t.navg + t.ngap + j];
    } else {
      a = l.s[k + flen - 1] - l.s[k + t.navg + t.ngap - 1];
    }
    if (t.navg <= SHORT) {
      b = 0.f;
      for (int j = 0; j < t.navg; ++j) b += xr[k + j];
    } else {
      b = l.s[k + t.navg - 1] - ((k > 0) ? l.s[k - 1] : 0.f);
    }
    yr[k] = a * i2 - b * i1;
  }
}

// valid-mode FIR; hr = taps in correlation order (out[k] = sum_j hr[j] x[k+j])
__global__ void __launch_bounds__(256) k_fir(const float* __restrict__ x, int L, const float* __restrict__ hr, int m, float* __restrict__ y) {
  extern __shared__ __align__(16) unsigned char raw[];
  Lds l = carve(raw, L, false);
  const int nout = L - m + 1;
  const float* xr = x + (size_t)blockIdx.x * L;
  float* yr = y + (size_t)blockIdx.x * nout;
  const int tid = threadIdx.x, NT = blockDim.x;
  tb::load_trace(xr, l.s, L);
  for (int i = L + tid; i < pad4(L); i += NT) l.s[i] = 0.f;
  __syncthreads();
  constexpr int U = 8;
  for (int k0 = 0; k0 < nout; k0 += U * NT) {
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0.f;
    for (int j = 0; j < m; ++j) {
      const float hj = hr[j];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + tid + NT * u;
        if (k < nout) acc[u] = fmaf(hj, l.s[k + j], acc[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + tid + NT * u;
      if (k < nout) yr[k] = acc[u];
    }
  }
}

// element-wise filters straight from global memory
// MODE 0 DerivativeFilter(gain)  src/derivative.jl:47-55
// MODE 1 HaarAveragingFilter(ds) src/haar_filter.jl:26-39
// MODE 2 affine / truncate / reverse (shift_waveform, multiply_waveform, reverse_waveform, TruncateFilter)
template <int MODE>
__global__ void __launch_bounds__(256) k_elem(const float* __restrict__ x, int L, int Lout, float a, float b, int from, int until,
                                              int rev, const float* __restrict__ per_trace, float* __restrict__ y) {
  const float* xr = x + (size_t)blockIdx.x * L;
  float* yr = y + (size_t)blockIdx.x * Lout;
  const float extra = (MODE == 2 && per_trace) ? per_trace[blockIdx.x] : 0.f;
  for (int i = threadIdx.x; i < Lout; i += blockDim.x) {
    float v;
    if (MODE == 0) {
      const int hi = min(max(i, 1), L - 1), lo = max(i - 1, 0);
      v = a * (xr[hi] - xr[lo]);
    } else if (MODE == 1) {
      const int s = i * from;  // from = down-sampling rate
      v = (xr[s] + xr[min(s + 1, L - 1)]) * 0.70710678118654752f;
    } else {
      const int src = rev ? until - i : from + i;
      v = a * xr[src] + (b + extra);
    }
    yr[i] = v;
  }
}

// ---------------------------------------------------------------- QC classifier front end
// get_qc_classifier / get_qc_classifier_compressed (src/dsp_ml_routines.jl:9-24, 45-60): [signalstats(bl).mean and
// shift_waveform(-mean) when a baseline window is given (:26-34, :62-70)] -> HaarAveragingFilter(2) `levels` times
// (5 / 2) -> divide by max(|min|, |max|) of the result (0 -> 1) -> one row of the feature matrix handed to the SVM.
// One workgroup per trace; level 1 reads the trace from global memory, the further levels ping-pong in LDS.
__global__ void __launch_bounds__(256) k_qc_features(const float* __restrict__ x, int L, int levels, int bl_from, int bl_until,
                                                     float* __restrict__ feat, float* __restrict__ norm_out) {
  extern __shared__ __align__(16) unsigned char raw[];
  __shared__ double red[8];
  __shared__ float fred[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (size_t)blockIdx.x * L;
  const int n1 = (L + 1) >> 1;
  float* A = reinterpret_cast<float*>(raw);   // [ceil(L/2)]
  float* B = A + ((n1 + 3) & ~3);             // [ceil(L/4)]
  float mean = 0.f;
  if (bl_from >= 0) {   // mean about the window's first sample, wave partials combined in double (as the fused kernels)
    const float pv = xr[bl_from];
    float s = 0.f;
    for (int i = bl_from + tid; i <= bl_until; i += 256) s += xr[i] - pv;
    s = wave_sum_all(s);
    if (lane == 0) red[wave] = (double)s;
    __syncthreads();
    mean = (float)((double)pv + (red[0] + red[1] + red[2] + red[3]) / (double)(bl_until - bl_from + 1));
  }
  const float c = 0.70710678118654752f;   // inv(sqrt(T(2)))  src/haar_filter.jl:27
  if ((L & 3) == 0) {   // rows 16-byte aligned: one dwordx4 load feeds two outputs, the loads of a thread are independent
    for (int j = tid; j < (L >> 2); j += 256) {
      const float4 v = *reinterpret_cast<const float4*>(xr + 4 * j);
      *reinterpret_cast<float2*>(&A[2 * j]) = make_float2(((v.x - mean) + (v.y - mean)) * c, ((v.z - mean) + (v.w - mean)) * c);
    }
  } else {
    for (int i = tid; i < n1; i += 256) {
      const int s = 2 * i;
      A[i] = ((xr[s] - mean) + (xr[min(s + 1, L - 1)] - mean)) * c;
    }
  }
  __syncthreads();
  float *src = A, *dst = B;
  int n = n1;
  for (int lv = 1; lv < levels; ++lv) {
    const int no = (n + 1) >> 1;
    for (int i = tid; i < no; i += 256) dst[i] = (src[2 * i] + src[min(2 * i + 1, n - 1)]) * c;
    __syncthreads();
    float* t = src; src = dst; dst = t;
    n = no;
  }
  float mx = -INFINITY, mn = INFINITY;
  for (int i = tid; i < n; i += 256) { mx = fmaxf(mx, src[i]); mn = fminf(mn, src[i]); }
  mx = wave_max_all(mx); mn = wave_min_all(mn);
  if (lane == 0) { fred[wave] = mx; fred[4 + wave] = mn; }
  __syncthreads();
  mx = fmaxf(fmaxf(fred[0], fred[1]), fmaxf(fred[2], fred[3]));
  mn = fminf(fminf(fred[4], fred[5]), fminf(fred[6], fred[7]));
  float nf = fmaxf(fabsf(mn), fabsf(mx));   // max(abs(first(extrema)), abs(last(extrema)))  :17
  if (nf == 0.f) nf = 1.f;                  // replace!(norm_fact, 0.0 => 1)  :18
  const float inv = 1.f / nf;               // multiply_waveform.(w, 1 ./ norm_fact)  :20
  float* fr = feat + (size_t)blockIdx.x * n;
  for (int i = tid; i < n; i += 256) fr[i] = src[i] * inv;
  if (norm_out && tid == 0) norm_out[blockIdx.x] = nf;
}

// ---------------------------------------------------------------- extractors
// MODE 0 signalstats, 1 tailstats, 2 extremestats, 3 get_wvf_maximum, 4 thresholdstats,
//      5 thresholdstats_mad, 6 saturation
struct StatOut {
  float* f[4];
  int32_t* i[4];
};
template <int MODE>
__global__ void __launch_bounds__(256) k_stats(const float* __restrict__ x, int L, int from, int until, float t_first, float dt,
                                               float lo, float hi, StatOut o) {
  extern __shared__ __align__(16) unsigned char raw[];
  Lds l = carve(raw, L, false);
  const float* xr = x + (size_t)blockIdx.x * L;
  const int tid = threadIdx.x, NT = blockDim.x;
  const size_t b = blockIdx.x;
  tb::load_trace(xr, l.s, L);
  __syncthreads();
  if (MODE == 0 || MODE == 1) {
    float m, sg, sl, of;
    const bool ok = tb::window_stats<MODE == 1>(l.s, from, until, t_first, dt, *l.sc, &m, &sg, &sl, &of);
    if (tid == 0) {
      if (MODE == 0) { o.f[0][b] = m; o.f[1][b] = sg; o.f[2][b] = sl; o.f[3][b] = of; }
      else { o.f[0][b] = m; o.f[1][b] = sg; o.f[2][b] = ok ? -1.f / sl : 0.f; }
    }
  } else if (MODE == 2) {
    float vmin, vmax; int imin, imax;
    tb::extreme_stats(l.s, from, until, *l.sc, &vmin, &imin, &vmax, &imax);
    if (tid == 0) { o.f[0][b] = vmin; o.f[1][b] = vmax; o.f[2][b] = t_first + dt * (float)imin; o.f[3][b] = t_first + dt * (float)imax; }
  } else if (MODE == 3) {
    const float v = tb::window_max_interp(l.s, from, until, *l.sc);
    if (tid == 0) o.f[0][b] = v;
  } else if (MODE == 4) {
    // thresholdstats (src/thresholdstats.jl:19-41): excluded samples enter as 0, n counts the included
    double v[3] = {0, 0, 0};
    for (int i = tid; i < L; i += NT) {
      const float yv = l.s[i];
      if (lo <= yv && yv <= hi) { v[0] += (double)yv; v[1] = fma((double)yv, (double)yv, v[1]); v[2] += 1.0; }
    }
    tb::blk_sum3(v, *l.sc);
    if (tid == 0) {
      const double inv_n = 1.0 / v[2], m = v[0] * inv_n;
      double var = v[1] * inv_n - m * m;
      if (var < 0) var = 0;
      o.f[0][b] = (float)sqrt(var);  // n == 0: inf*0 = NaN, as inv(0) does in the reference
    }
  } else if (MODE == 5) {
    const float r = tb::mad_threshold(l.s, L, lo, hi, 1.f, *l.sc);
    if (tid == 0) o.f[0][b] = r;
  } else {
    // saturation (src/saturation.jl:28-65): lo / hi carry the two levels
    int nl = 0, nh = 0;
    for (int i = from + tid; i <= until; i += NT) { nl += (l.s[i] == lo); nh += (l.s[i] == hi); }
    nl = tb::blk_sum_i(nl, *l.sc);
    nh = tb::blk_sum_i(nh, *l.sc);
    int cl = 0, ch = 0;
    if (nl > 0 || nh > 0) {
      if (tid < 2) {  // rare path: one thread per level walks the window
        const float lev = tid ? hi : lo;
        int best = 0, run = 0;
        for (int i = from; i <= until; ++i) {
          if (l.s[i] == lev) ++run;
          else { best = max(best, run); run = 0; }
        }
        l.sc->i[6 + tid] = max(best, run);
      }
      __syncthreads();
      cl = l.sc->i[6]; ch = l.sc->i[7];
    }
    if (tid == 0) { o.i[0][b] = nl; o.i[1][b] = nh; o.i[2][b] = cl; o.i[3][b] = ch; }
  }
}

// Intersect(min_n)(wf, thr[trace]) -> (x, multiplicity)   
__global__ void __launch_bounds__(256) k_intersect(const float* __restrict__ x, int L, float t_first, float dt,
                                                   const float* __restrict__ thr, int min_n, float* __restrict__ xout,
                                                   int32_t* __restrict__ mult) {
  extern __shared__ __align__(16) unsigned char raw[];
  Lds l = carve(raw, L, false);
  const float* xr = x + (size_t)blockIdx.x * L;
  tb::load_trace(xr, l.s, L);
  __syncthreads();
  const float th = thr[blockIdx.x];
  tb::build_mask(l.s, L, 1.f, th, l.bm);
  __syncthreads();
  int first, count;
  tb::intersect_scan(l.bm, L, min_n, *l.sc, &first, &count);
  if (threadIdx.x == 0) {
    float xo = NAN;
    if (count > 0) {
      const float yl = l.s[first - 1], yh = l.s[first];
      xo = t_first + dt * ((float)(first - 1) + (th - yl) / (yh - yl));
    }
    xout[blockIdx.x] = xo;
    if (mult) mult[blockIdx.x] = count;
  }
}

#include "intersect_maximum_block.inc"

__global__ void __launch_bounds__(256) k_intersect_maximum(const float* __restrict__ x, int L, double t_first, double dt,
                                                           const float* __restrict__ thr, int min_n, int max_n,
                                                           ldsp_trig_out o) {
  extern __shared__ __align__(16) unsigned char raw[];
  Lds l = carve(raw, L, false);
  const float* xr = x + (size_t)blockIdx.x * L;
  tb::load_trace(xr, l.s, L);
  __syncthreads();
  const float th = thr[blockIdx.x];
  tb::build_mask(l.s, L, 1.f, th, l.bm);
  __syncthreads();
  const size_t off = (size_t)blockIdx.x * (size_t)o.cap;
  const int total = intersect_maximum_block(l.s, L, 1.f, th, min_n, max_n, t_first, dt, l.bm, *l.sc, o.cap,
                                            o.x ? o.x + off : nullptr, o.x_high ? o.x_high + off : nullptr,
                                            o.x_tot ? o.x_tot + off : nullptr, o.max ? o.max + off : nullptr, nullptr);
  if (threadIdx.x == 0 && o.count) o.count[blockIdx.x] = total;
}

// MultiIntersect (src/multi_intersect.jl:36-104).  tab: [K ratios | W (m x w) upsampling weights]
//
// The reference walks the trace once with a threshold index that advances at every confirmed crossing and a rewind to
// the crossing's first sample (:53-72).  For ascending thresholds that walk has a closed form:
//   pos_0 = Intersect(min_n): the first run of min_n samples >= thr_0 that does not start the trace (:51, the counter
//           armed at min_n + 1), and
//   pos_k = the first s >= pos_0 with y[s .. s+min_n-1] >= thr_k              (k >= 1)
// because the counter restarts at 0 on the rewind (a run may begin AT the restart sample), {y >= thr_k} shrinks with k,
// and therefore the first window of threshold k from pos_0 is also the first one from pos_{k-1} (pos_{k-1} <= it).  A
// threshold that is never confirmed ends the walk: it and every later one keep the initial position 1 (:47).
// So: pos_0 by the block's bit-mask Intersect, then one WAVE per threshold, 64 samples per step by ballot, starting
// from the wave's previous crossing.  Thresholds that do not ascend (negative maximum, unsorted ratios) or
// min_n > 32 take the reference's serial walk on one lane.
__global__ void __launch_bounds__(512) k_multi_intersect(const float* __restrict__ x, int L, float t_first, float dt,
                                                         const float* __restrict__ tab, int K, int min_n, int half_n, int m,
                                                         int force_serial, float* __restrict__ xout, int32_t* __restrict__ status) {
  extern __shared__ __align__(16) unsigned char raw[];
  Lds l = carve(raw, L, false);
  const float* xr = x + (size_t)blockIdx.x * L;
  const int tid = threadIdx.x, NT = blockDim.x;
  tb::load_trace(xr, l.s, L);
  __syncthreads();
  float bmx = -INFINITY; int bi = 0;
  for (int i = tid; i < L; i += NT) if (l.s[i] > bmx) { bmx = l.s[i]; bi = i; }
  float ymax; int imax;
  tb::blk_argmax(bmx, bi, *l.sc, &ymax, &imax);
  __shared__ int s_ipos[LDSP_MAX_MULTI];
  __shared__ int s_rc, s_desc;
  if (tid == 0) s_desc = 0;
  for (int k = tid; k < K; k += NT) s_ipos[k] = 1;
  __syncthreads();
  for (int k = tid; k + 1 < K; k += NT) if (!(tab[k + 1] * ymax >= tab[k] * ymax)) s_desc = 1;
  const float th0 = tab[0] * ymax;
  tb::build_mask(l.s, L, 1.f, th0, l.bm);
  __syncthreads();
  const bool serial = force_serial || s_desc || min_n > 32;
  if (!serial) {
    int first, count;
    tb::intersect_scan(l.bm, L, min_n, *l.sc, &first, &count);
    if (count > 0) {
      const int wave = tid >> 6, lane = tid & 63, nw = NT >> 6;
      if (tid == 0) s_ipos[0] = first;
      int base = first;                      // a later threshold's window is never left of an earlier one's
      const int stride = 64 - (min_n - 1);   // every window of min_n samples lies inside one step
      for (int k = 1 + wave; k < K; k += nw) {
        const float th = tab[k] * ymax;
        int pos = -1;
        while (base < L) {
          const int i = base + lane;
          const float v = i < L ? l.s[i] : -INFINITY;
          unsigned long long r = __ballot(v >= th);
          for (int j = 1; j < min_n; ++j) r &= r >> 1;     // bit s: samples s .. s+min_n-1 of this step all high
          if (r) { pos = base + (int)__builtin_ctzll(r); break; }
          base += stride;
        }
        if (pos < 0) break;                  // not confirmed: neither is any later threshold
        base = pos;
        if (lane == 0) s_ipos[k] = pos;
      }
    }
  } else if (tid == 0) {
    // the reference's scan with threshold index advance and rewind (:53-72)
    const float* ratios = tab;
    int cand = 1, ic = 0, i = 0;
    int cnt = (l.s[0] >= ratios[0] * ymax) ? min_n + 1 : 0;
    while (i < L && ic < K) {
      const float th = ratios[ic] * ymax;
      const bool high = l.s[i] >= th;
      if (high && cnt == 0) cand = i;
      cnt = high ? cnt + 1 : 0;
      const bool found = (cnt == min_n);
      const int pos = found ? cand : s_ipos[ic];
      s_ipos[ic] = pos;
      i = found ? pos : i + 1;
      if (found) { ++ic; cnt = 0; }
    }
  }
  if (tid == 0) s_rc = 0;
  __syncthreads();
  // :75-78 asserts the first and the last window; the fit loop (:88-103) runs @inbounds, so a window of a threshold in
  // between that leaves the trace (an unconfirmed threshold keeps position 1, n >= 2) would read out of bounds there.
  // Every window is checked here and the trace flagged (as the oracle does).
  for (int k = tid; k < K; k += NT)
    if (s_ipos[k] - half_n < 0 || s_ipos[k] + half_n - 1 > L - 1) s_rc = LDSP_ERR_WINDOW;
  __syncthreads();
  const int w = 2 * half_n;
  const float* W = tab + K;
  for (int k = tid; k < K; k += NT) {
    float xo = 0.f;
    const int from = s_ipos[k] - half_n, to = s_ipos[k] + half_n - 1;
    if (s_rc == 0 && from >= 0 && to <= L - 1) {
      const float th = tab[k] * ymax;
      const float x0 = t_first + dt * (float)from, x1 = t_first + dt * (float)to;
      const float dtu = (m > 1) ? (x1 - x0) / (float)(m - 1) : 0.f;
      // Intersect with min_n = 1 on the upsampled window (:97): first up-crossing that is
      // not the initial run (an already-high first sample arms the counter at min_n + 1)
      float prev = 0.f, yl = 0.f, yh = 0.f;
      int cnt = 0, first = -1;
      for (int q = 0; q < m; ++q) {
        float v = 0.f;
        for (int a = 0; a < w; ++a) v = fmaf(W[q * w + a], l.s[from + a], v);
        const bool high = v >= th;
        if (q == 0) cnt = high ? 2 : 0;
        if (high && cnt == 0 && first < 0) { first = q; yl = prev; yh = v; }
        cnt = high ? cnt + 1 : 0;
        prev = v;
      }
      xo = (first > 0) ? x0 + dtu * ((float)(first - 1) + (th - yl) / (yh - yl)) : NAN;
    }
    xout[(size_t)blockIdx.x * K + k] = xo;
  }
  if (tid == 0 && status) status[blockIdx.x] = s_rc;
}

// SignalEstimator(PolynomialDNI(degree, npts))(wf, t[trace])   (assumption A3)
// B: [npts][deg+1] LSQ weights in the centred/scaled basis, then c, s_inv.
__global__ void __launch_bounds__(64) k_signal_estimator(const float* __restrict__ x, int L, float t_first, float dt,
                                                         const float* __restrict__ t, const float* __restrict__ B, int npts,
                                                         int deg, float* __restrict__ out) {
  const float* xr = x + (size_t)blockIdx.x * L;
  const int l = threadIdx.x;
  float res = NAN;
  if (L >= npts) {
    float p = (t[blockIdx.x] - t_first) / dt;
    if (p == p) {
      p = fminf(fmaxf(p, 0.f), (float)(L - 1));
      int i0 = (int)ceilf(p - 0.5f * (float)npts);
      i0 = max(0, min(i0, L - npts));
      const float c = B[npts * (deg + 1)], s_inv = B[npts * (deg + 1) + 1];
      const float u = (p - (float)i0 - c) * s_inv;
      float v = 0.f;
      if (l < npts) {
        const float* b = B + l * (deg + 1);
        float wgt = b[deg];
        for (int j = deg - 1; j >= 0; --j) wgt = fmaf(wgt, u, b[j]);
        v = wgt * xr[i0 + l];
      }
      res = wave_sum_all(v);
    }
  }
  if (l == 0) out[blockIdx.x] = res;
}

}  // namespace fk
}  // namespace ldsp

// ===========================================================================
// C ABI
using namespace ldsp;
using namespace ldsp::fk;

static int upload_coef(ldsp_ctx* c, const std::vector<float>& v) {
  if (v.size() > LDSP_MAX_FIR_TAPS) return ldsp_fail(LDSP_ERR_UNSUPPORTED, "coefficient table of %zu entries exceeds %d", v.size(), LDSP_MAX_FIR_TAPS);
  HIP_TRY(hipMemcpyAsync(c->d_coef, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // staging vector dies at return; also keeps d_coef single-use
  return LDSP_OK;
}
template <typename K>
static int set_lds(K kern, size_t bytes) {
  if (bytes > 160 * 1024) return ldsp_fail(LDSP_ERR_UNSUPPORTED, "trace does not fit the 160 KiB LDS of a CU (%zu B needed)", bytes);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return LDSP_OK;
}
#define LAUNCH_CHECK()                                                       \
  do {                                                                        \
    hipError_t _e = hipGetLastError();                                        \
    if (_e != hipSuccess) return ldsp_fail(LDSP_ERR_HIP, "launch: %s", hipGetErrorString(_e)); \
  } while (0)

static bool win_ok(int from, int until, int L) { return 0 <= from && from <= until && until <= L - 1; }

extern "C" {

int ldsp_rdfilt_invcr(ldsp_ctx* c, const float* x, int64_t n, int32_t L, double cc, float* y) {
  int rc = ldsp_check_batch(c, x, n, L, "ldsp_rdfilt_invcr");
  if (rc || n == 0) return rc;
  if (!y) return ldsp_fail(LDSP_ERR_INVALID_ARG, "output pointer is NULL");
  size_t b = lds_bytes(L, false);
  if ((rc = set_lds(k_scan_filter<0>, b))) return rc;
  hipLaunchKernelGGL(k_scan_filter<0>, dim3((unsigned)n), dim3(256), b, c->stream, x, L, (float)cc, 0, y);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_rdfilt_integrator(ldsp_ctx* c, const float* x, int64_t n, int32_t L, double gain, float* y) {
  int rc = ldsp_check_batch(c, x, n, L, "ldsp_rdfilt_integrator");
  if (rc || n == 0) return rc;
  if (!y) return ldsp_fail(LDSP_ERR_INVALID_ARG, "output pointer is NULL");
  size_t b = lds_bytes(L, false);
  if ((rc = set_lds(k_scan_filter<1>, b))) return rc;
  hipLaunchKernelGGL(k_scan_filter<1>, dim3((unsigned)n), dim3(256), b, c->stream, x, L, (float)gain, 0, y);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_rdfilt_moving_window(ldsp_ctx* c, const float* x, int64_t n, int32_t L, int32_t len, float* y) {
  int rc = ldsp_check_batch(c, x, n, L, "ldsp_rdfilt_moving_window");
  if (rc || n == 0) return rc;
  if (!y || len < 1) return ldsp_fail(LDSP_ERR_INVALID_ARG, "bad output pointer / window length %d", len);
  size_t b = lds_bytes(L, false);
  if ((rc = set_lds(k_scan_filter<2>, b))) return rc;
  hipLaunchKernelGGL(k_scan_filter<2>, dim3((unsigned)n), dim3(256), b, c->stream, x, L, 0.f, len, y);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_rdfilt_moving_window_multi(ldsp_ctx* c, const float* x, int64_t n, int32_t L, int32_t len, float* y) {
  int rc = ldsp_check_batch(c, x, n, L, "ldsp_rdfilt_moving_window_multi");
  if (rc || n == 0) return rc;
  if (!y || len < 1) return ldsp_fail(LDSP_ERR_INVALID_ARG, "bad output pointer / window length %d", len);
  size_t b = lds_bytes(L, true);
  if ((rc = set_lds(k_scan_filter<3>, b))) return rc;
  hipLaunchKernelGGL(k_scan_filter<3>, dim3((unsigned)n), dim3(256), b, c->stream, x, L, 0.f, len, y);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_rdfilt_trap(ldsp_ctx* c, const float* x, int64_t n, int32_t L, ldsp_trap t, float* y) {
  int rc = ldsp_check_batch(c, x, n, L, "ldsp_rdfilt_trap");
  if (rc || n == 0) return rc;
  if (!y) return ldsp_fail(LDSP_ERR_INVALID_ARG, "output pointer is NULL");
  if (t.navg < 1 || t.navg2 < 1 || t.ngap < 0) return ldsp_fail(LDSP_ERR_INVALID_ARG, "bad trapezoid (%d,%d,%d)", t.navg, t.ngap, t.navg2);
  if (t.navg + t.ngap + t.navg2 > L) return ldsp_fail(LDSP_ERR_WINDOW, "trapezoid longer than the trace");
  size_t b = lds_bytes(L, false);
  if ((rc = set_lds(k_trap, b))) return rc;
  hipLaunchKernelGGL(k_trap, dim3((unsigned)n), dim3(256), b, c->stream, x, L, t, y);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_rdfilt_fir(ldsp_ctx* c, const float* x, int64_t n, int32_t L, const double* h, int32_t ntaps, float* y) {
  int rc = ldsp_check_batch(c, x, n, L, "ldsp_rdfilt_fir");
  if (rc || n == 0) return rc;
  if (!y || !h || ntaps < 1) return ldsp_fail(LDSP_ERR_INVALID_ARG, "bad FIR arguments");
  if (ntaps > L) return ldsp_fail(LDSP_ERR_WINDOW, "FIR longer than the trace");
  std::vector<float> hr(ntaps);
  for (int j = 0; j < ntaps; ++j) hr[j] = (float)h[ntaps - 1 - j];
  if ((rc = upload_coef(c, hr))) return rc;
  size_t b = lds_bytes(L, false);
  if ((rc = set_lds(k_fir, b))) return rc;
  hipLaunchKernelGGL(k_fir, dim3((unsigned)n), dim3(256), b, c->stream, x, L, (const float*)c->d_coef, ntaps, y);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_rdfilt_derivative(ldsp_ctx* c, const float* x, int64_t n, int32_t L, double gain, float* y) {
  int rc = ldsp_check_batch(c, x, n, L, "ldsp_rdfilt_derivative");
  if (rc || n == 0) return rc;
  if (!y) return ldsp_fail(LDSP_ERR_INVALID_ARG, "output pointer is NULL");
  hipLaunchKernelGGL(k_elem<0>, dim3((unsigned)n), dim3(256), 0, c->stream, x, L, L, (float)gain, 0.f, 0, 0, 0, (const float*)nullptr, y);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_rdfilt_haar(ldsp_ctx* c, const float* x, int64_t n, int32_t L, int32_t ds, float* y) {
  int rc = ldsp_check_batch(c, x, n, L, "ldsp_rdfilt_haar");
  if (rc || n == 0) return rc;
  if (!y || ds < 1) return ldsp_fail(LDSP_ERR_INVALID_ARG, "bad output pointer / down-sampling rate %d", ds);
  hipLaunchKernelGGL(k_elem<1>, dim3((unsigned)n), dim3(256), 0, c->stream, x, L, (L + ds - 1) / ds, 0.f, 0.f, ds, 0, 0, (const float*)nullptr, y);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_rdfilt_affine(ldsp_ctx* c, const float* x, int64_t n, int32_t L, int32_t from, int32_t until, double scale, double shift,
                       const float* shift_per_trace, int32_t reverse, float* y) {
  int rc = ldsp_check_batch(c, x, n, L, "ldsp_rdfilt_affine");
  if (rc || n == 0) return rc;
  if (!y) return ldsp_fail(LDSP_ERR_INVALID_ARG, "output pointer is NULL");
  if (!win_ok(from, until, L)) return ldsp_fail(LDSP_ERR_WINDOW, "range [%d,%d] outside the trace", from, until);
  hipLaunchKernelGGL(k_elem<2>, dim3((unsigned)n), dim3(256), 0, c->stream, x, L, until - from + 1, (float)scale, (float)shift, from,
                     until, reverse, shift_per_trace, y);
  LAUNCH_CHECK();
  return LDSP_OK;
}

#define STATS_COMMON(NAME)                                              \
  int rc = ldsp_check_batch(c, x, n, L, NAME);                          \
  if (rc || n == 0) return rc;                                          \
  size_t b = lds_bytes(L, false);

int ldsp_signalstats(ldsp_ctx* c, const float* x, int64_t n, int32_t L, int32_t from, int32_t until, double t_first, double dt,
                     float* mean, float* sigma, float* slope, float* offset) {
  STATS_COMMON("ldsp_signalstats");
  if (!mean || !sigma || !slope || !offset) return ldsp_fail(LDSP_ERR_INVALID_ARG, "output pointer is NULL");
  if (!win_ok(from, until, L)) return ldsp_fail(LDSP_ERR_WINDOW, "window [%d,%d] outside the trace", from, until);
  if ((rc = set_lds(k_stats<0>, b))) return rc;
  StatOut o{{mean, sigma, slope, offset}, {}};
  hipLaunchKernelGGL(k_stats<0>, dim3((unsigned)n), dim3(256), b, c->stream, x, L, from, until, (float)t_first, (float)dt, 0.f, 0.f, o);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_tailstats(ldsp_ctx* c, const float* x, int64_t n, int32_t L, int32_t from, int32_t until, double t_first, double dt,
                   float* mean, float* sigma, float* tau) {
  STATS_COMMON("ldsp_tailstats");
  if (!mean || !sigma || !tau) return ldsp_fail(LDSP_ERR_INVALID_ARG, "output pointer is NULL");
  if (!win_ok(from, until, L)) return ldsp_fail(LDSP_ERR_WINDOW, "window [%d,%d] outside the trace", from, until);
  if ((rc = set_lds(k_stats<1>, b))) return rc;
  StatOut o{{mean, sigma, tau, nullptr}, {}};
  hipLaunchKernelGGL(k_stats<1>, dim3((unsigned)n), dim3(256), b, c->stream, x, L, from, until, (float)t_first, (float)dt, 0.f, 0.f, o);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_extremestats(ldsp_ctx* c, const float* x, int64_t n, int32_t L, int32_t from, int32_t until, double t_first, double dt,
                      float* vmin, float* vmax, float* tmin, float* tmax) {
  STATS_COMMON("ldsp_extremestats");
  if (!vmin || !vmax || !tmin || !tmax) return ldsp_fail(LDSP_ERR_INVALID_ARG, "output pointer is NULL");
  if (!win_ok(from, until, L)) return ldsp_fail(LDSP_ERR_WINDOW, "window [%d,%d] outside the trace", from, until);
  if ((rc = set_lds(k_stats<2>, b))) return rc;
  StatOut o{{vmin, vmax, tmin, tmax}, {}};
  hipLaunchKernelGGL(k_stats<2>, dim3((unsigned)n), dim3(256), b, c->stream, x, L, from, until, (float)t_first, (float)dt, 0.f, 0.f, o);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_get_wvf_maximum(ldsp_ctx* c, const float* x, int64_t n, int32_t L, int32_t from, int32_t until, float* vmax) {
  STATS_COMMON("ldsp_get_wvf_maximum");
  if (!vmax) return ldsp_fail(LDSP_ERR_INVALID_ARG, "output pointer is NULL");
  if (!win_ok(from, until, L)) return ldsp_fail(LDSP_ERR_WINDOW, "window [%d,%d] outside the trace", from, until);
  if ((rc = set_lds(k_stats<3>, b))) return rc;
  StatOut o{{vmax, nullptr, nullptr, nullptr}, {}};
  hipLaunchKernelGGL(k_stats<3>, dim3((unsigned)n), dim3(256), b, c->stream, x, L, from, until, 0.f, 1.f, 0.f, 0.f, o);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_thresholdstats(ldsp_ctx* c, const float* x, int64_t n, int32_t L, double lo, double hi, float* sigma) {
  STATS_COMMON("ldsp_thresholdstats");
  if (!sigma) return ldsp_fail(LDSP_ERR_INVALID_ARG, "output pointer is NULL");
  if ((rc = set_lds(k_stats<4>, b))) return rc;
  StatOut o{{sigma, nullptr, nullptr, nullptr}, {}};
  hipLaunchKernelGGL(k_stats<4>, dim3((unsigned)n), dim3(256), b, c->stream, x, L, 0, L - 1, 0.f, 1.f, (float)lo, (float)hi, o);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_thresholdstats_mad(ldsp_ctx* c, const float* x, int64_t n, int32_t L, double lo, double hi, float* mad) {
  STATS_COMMON("ldsp_thresholdstats_mad");
  if (!mad) return ldsp_fail(LDSP_ERR_INVALID_ARG, "output pointer is NULL");
  if ((rc = set_lds(k_stats<5>, b))) return rc;
  StatOut o{{mad, nullptr, nullptr, nullptr}, {}};
  hipLaunchKernelGGL(k_stats<5>, dim3((unsigned)n), dim3(256), b, c->stream, x, L, 0, L - 1, 0.f, 1.f, (float)lo, (float)hi, o);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_saturation(ldsp_ctx* c, const float* x, int64_t n, int32_t L, int32_t from, int32_t until, double low, double high,
                    int32_t* n_low, int32_t* n_high, int32_t* cons_low, int32_t* cons_high) {
  STATS_COMMON("ldsp_saturation");
  if (!n_low || !n_high || !cons_low || !cons_high) return ldsp_fail(LDSP_ERR_INVALID_ARG, "output pointer is NULL");
  if (!win_ok(from, until, L)) return ldsp_fail(LDSP_ERR_WINDOW, "window [%d,%d] outside the trace", from, until);
  if ((rc = set_lds(k_stats<6>, b))) return rc;
  StatOut o{{}, {n_low, n_high, cons_low, cons_high}};
  hipLaunchKernelGGL(k_stats<6>, dim3((unsigned)n), dim3(256), b, c->stream, x, L, from, until, 0.f, 1.f, (float)low, (float)high, o);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_intersect(ldsp_ctx* c, const float* x, int64_t n, int32_t L, double t_first, double dt, const float* thr, int32_t min_n,
                   float* xout, int32_t* mult) {
  STATS_COMMON("ldsp_intersect");
  if (!thr || !xout || min_n < 1) return ldsp_fail(LDSP_ERR_INVALID_ARG, "bad Intersect arguments");
  if ((rc = set_lds(k_intersect, b))) return rc;
  hipLaunchKernelGGL(k_intersect, dim3((unsigned)n), dim3(256), b, c->stream, x, L, (float)t_first, (float)dt, thr, min_n, xout, mult);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_intersect_maximum(ldsp_ctx* c, const float* x, int64_t n, int32_t L, double t_first, double dt, const float* thr,
                           int32_t min_n, int32_t max_n, const ldsp_trig_out* out) {
  STATS_COMMON("ldsp_intersect_maximum");
  if (!thr || !out || min_n < 1 || max_n < 1 || out->cap < 0) return ldsp_fail(LDSP_ERR_INVALID_ARG, "bad IntersectMaximum arguments");
  if ((rc = set_lds(k_intersect_maximum, b))) return rc;
  ldsp_trig_out o = *out;
  if (o.cap == 0) o.cap = LDSP_MAX_TRIG;
  hipLaunchKernelGGL(k_intersect_maximum, dim3((unsigned)n), dim3(256), b, c->stream, x, L, t_first, dt, thr, min_n, max_n, o);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_multi_intersect(ldsp_ctx* c, const float* x, int64_t n, int32_t L, double t_first, double dt, const double* ratios,
                         int32_t K, int32_t min_n, int32_t half_n, int32_t degree, int32_t rate, float* xout, int32_t* status) {
  STATS_COMMON("ldsp_multi_intersect");
  if (!ratios || !xout || K < 1 || K > LDSP_MAX_MULTI || min_n < 1 || half_n < 1 || rate < 1 || degree < 0)
    return ldsp_fail(LDSP_ERR_INVALID_ARG, "bad MultiIntersect arguments");
  const int w = 2 * half_n, m = w * rate;
  if (degree >= w || (size_t)K + (size_t)m * w > LDSP_MAX_FIR_TAPS) return ldsp_fail(LDSP_ERR_UNSUPPORTED, "MultiIntersect window/degree/rate too large");
  std::vector<double> B;
  double cc, ss;
  if (!hm::lsq_basis(w, degree, B, cc, ss)) return ldsp_fail(LDSP_ERR_INVALID_ARG, "MultiIntersect fit matrix");
  std::vector<float> tab(K + (size_t)m * w);
  for (int k = 0; k < K; ++k) tab[k] = (float)ratios[k];
  for (int q = 0; q < m; ++q) {  // x_up = range(0, 2n-1, m)   (:81)
    const double xu = (m > 1) ? (double)q * (double)(w - 1) / (double)(m - 1) : 0.0;
    const double u = (xu - cc) / ss;
    for (int a = 0; a < w; ++a) {
      double wgt = 0, pu = 1;
      for (int j = 0; j <= degree; ++j) { wgt += B[(size_t)a * (degree + 1) + j] * pu; pu *= u; }
      tab[K + (size_t)q * w + a] = (float)wgt;
    }
  }
  if ((rc = upload_coef(c, tab))) return rc;
  if ((rc = set_lds(k_multi_intersect, b))) return rc;
  hipLaunchKernelGGL(k_multi_intersect, dim3((unsigned)n), dim3(512), b, c->stream, x, L, (float)t_first, (float)dt,
                     (const float*)c->d_coef, K, min_n, half_n, m, c->multi_serial ? 1 : 0, xout, status);
  LAUNCH_CHECK();
  return LDSP_OK;
}
int ldsp_signal_estimator(ldsp_ctx* c, const float* x, int64_t n, int32_t L, double t_first, double dt, const float* t, ldsp_dni est,
                          float* out) {
  int rc = ldsp_check_batch(c, x, n, L, "ldsp_signal_estimator");
  if (rc || n == 0) return rc;
  if (!t || !out) return ldsp_fail(LDSP_ERR_INVALID_ARG, "NULL pointer");
  if (est.npts < 1 || est.npts > LDSP_MAX_EST_PTS || est.degree < 0 || est.degree >= est.npts || est.degree > 12)
    return ldsp_fail(LDSP_ERR_UNSUPPORTED, "PolynomialDNI(%d, %d pts) unsupported", est.degree, est.npts);
  std::vector<double> B;
  double cc, ss;
  if (!hm::lsq_basis(est.npts, est.degree, B, cc, ss)) return ldsp_fail(LDSP_ERR_INVALID_ARG, "estimator fit matrix");
  std::vector<float> tab(B.begin(), B.end());
  tab.push_back((float)cc);
  tab.push_back((float)(1.0 / ss));
  if ((rc = upload_coef(c, tab))) return rc;
  hipLaunchKernelGGL(k_signal_estimator, dim3((unsigned)n), dim3(64), 0, c->stream, x, L, (float)t_first, (float)dt, t,
                     (const float*)c->d_coef, est.npts, est.degree, out);
  LAUNCH_CHECK();
  return LDSP_OK;
}

int32_t ldsp_qc_features_len(int32_t L, int32_t levels) {
  int32_t n = L;
  for (int i = 0; i < levels; ++i) n = (n + 1) / 2;
  return n;
}
int ldsp_qc_features(ldsp_ctx* c, const float* x, int64_t n, int32_t L, int32_t levels, int32_t bl_from, int32_t bl_until,
                     float* features, float* norm) {
  int rc = ldsp_check_batch(c, x, n, L, "ldsp_qc_features");
  if (rc || n == 0) return rc;
  if (!features) return ldsp_fail(LDSP_ERR_INVALID_ARG, "output pointer is NULL");
  if (levels < 1 || levels > 16) return ldsp_fail(LDSP_ERR_INVALID_ARG, "Haar levels %d outside 1..16", levels);
  if (bl_from >= 0 && !win_ok(bl_from, bl_until, L)) return ldsp_fail(LDSP_ERR_WINDOW, "baseline window [%d,%d] outside the trace", bl_from, bl_until);
  const int n1 = (L + 1) / 2, n2 = (n1 + 1) / 2;
  const size_t b = ((size_t)((n1 + 3) & ~3) + (size_t)((n2 + 3) & ~3)) * 4;
  if (b > 150 * 1024) return ldsp_fail(LDSP_ERR_UNSUPPORTED, "trace of %d samples does not fit the Haar stage's LDS", L);
  if ((rc = set_lds(k_qc_features, b))) return rc;
  hipLaunchKernelGGL(k_qc_features, dim3((unsigned)n), dim3(256), b, c->stream, x, L, levels, bl_from, bl_until, features, norm);
  LAUNCH_CHECK();
  return LDSP_OK;
}

}  // extern "C"
