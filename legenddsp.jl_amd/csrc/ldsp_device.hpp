// ldsp_device.hpp — device-side building blocks shared by the gfx950 kernels.
//
// Data model inside a workgroup: ONE trace per workgroup, NT = ceil(L/32)
// threads (rounded to whole 64-lane waves).  Two register/LDS views of a trace:
//   * thread-blocked: thread t owns samples 32t..32t+31 (serial recursions,
//     prefix scans, bit-packing);
//   * lane-strided:   thread t visits samples t, t+NT, t+2NT, ... (shifted
//     reads for trapezoid/FIR windows: consecutive lanes -> consecutive banks).
// LDS arrays use a 16-byte-chunk XOR swizzle so that thread-blocked
// ds_read/write_b128 (row stride 128 B) are conflict-free while lane-strided
// b32 accesses stay at most 2-way conflicted when a wave straddles two rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ldsp {

constexpr int SPT = 32;        // samples per thread in the thread-blocked view
constexpr int MAX_WAVES = 16;  // 1024 threads

__device__ __forceinline__ int sw(int i) { return i ^ (((i >> 5) & 7) << 2); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- wave-level reductions (64 lanes) -------------------------------------
template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce(T v, Op op) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o, 64));
  return v;
}
struct OpSum { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; } };
struct OpMax { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; } };
struct OpMin { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a < b ? a : b; } };

// (value, index) pair; max by value, ties -> smaller index (findmax: first occurrence)
struct ValIdx { float v; int i; };
__device__ __forceinline__ ValIdx vi_max(ValIdx a, ValIdx b) {
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ValIdx wave_reduce_vimax(ValIdx x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ValIdx y; y.v = __shfl_xor(x.v, o, 64); y.i = __shfl_xor(x.i, o, 64);
    x = vi_max(x, y);
  }
  return x;
}

// ---- block-level helpers -----------------------------------------------------
// All take a scratch area in LDS of at least MAX_WAVES*K elements of T and end
// with every thread holding the result.  Two barriers each (publish, release).
template <int K, typename T, typename Op>
__device__ __forceinline__ void block_reduce(T (&v)[K], T* scratch, Op op) {
  const int nw = blockDim.x >> 6, w = wave_id(), l = lane_id();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_reduce(v[k], op);
  if (l == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) scratch[w * K + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    T a = scratch[k];
    for (int i = 1; i < nw; ++i) a = op(a, scratch[i * K + k]);
    v[k] = a;
  }
  __syncthreads();
}

__device__ __forceinline__ ValIdx block_reduce_vimax(ValIdx x, void* scratch_) {
  ValIdx* scratch = reinterpret_cast<ValIdx*>(scratch_);
  const int nw = blockDim.x >> 6, w = wave_id(), l = lane_id();
  x = wave_reduce_vimax(x);
  if (l == 0) scratch[w] = x;
  __syncthreads();
  ValIdx a = scratch[0];
  for (int i = 1; i < nw; ++i) a = vi_max(a, scratch[i]);
  __syncthreads();
  return a;
}

// Exclusive prefix sum over the block of one double per thread (thread order).
// Returns the exclusive prefix; *total (optional) receives the block total.
__device__ __forceinline__ double block_exscan_f64(double v, double* scratch, double* total) {
  const int nw = blockDim.x >> 6, w = wave_id(), l = lane_id();
  double inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double t = __shfl_up(inc, o, 64);
    if (l >= o) inc += t;
  }
  if (l == 63) scratch[w] = inc;
  __syncthreads();
  double base = 0, tot = 0;
  for (int i = 0; i < nw; ++i) {
    double s = scratch[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  if (total) *total = tot;
  return base + inc - v;
}

// Scan of first-order recurrences s <- a*s + b composed in thread order:
// each thread contributes the affine map (a, b) of its 32-sample chunk; returns
// the state ENTERING the thread's chunk given state 0 before thread 0.
// (forward direction; for the anti-causal filter the caller mirrors thread ids)
__device__ __forceinline__ float block_exscan_affine(float a, float b, float* scratch /*[2*MAX_WAVES]*/) {
  const int nw = blockDim.x >> 6, w = wave_id(), l = lane_id();
  float A = a, Bv = b;  // inclusive composite: x -> A*x + Bv
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float pa = __shfl_up(A, o, 64), pb = __shfl_up(Bv, o, 64);
    if (l >= o) { Bv = fmaf(A, pb, Bv); A = A * pa; }
  }
  if (l == 63) { scratch[2 * w] = A; scratch[2 * w + 1] = Bv; }
  __syncthreads();
  float s = 0.f;  // state entering this wave
  for (int i = 0; i < w && i < nw; ++i) s = fmaf(scratch[2 * i], s, scratch[2 * i + 1]);
  __syncthreads();
  // exclusive within the wave: composite of lanes < l applied to s
  float ea = __shfl_up(A, 1, 64), eb = __shfl_up(Bv, 1, 64);
  if (l == 0) { ea = 1.f; eb = 0.f; }
  return fmaf(ea, s, eb);
}

// Mirror image: maps composed from the LAST thread towards the first (anti-causal
// recurrences); returns the state entering the thread's chunk from the right.
__device__ __forceinline__ float block_exscan_affine_rev(float a, float b, float* scratch /*[2*MAX_WAVES]*/) {
  const int nw = blockDim.x >> 6, w = wave_id(), l = lane_id();
  float A = a, Bv = b;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float pa = __shfl_down(A, o, 64), pb = __shfl_down(Bv, o, 64);
    if (l + o < 64) { Bv = fmaf(A, pb, Bv); A = A * pa; }
  }
  if (l == 0) { scratch[2 * w] = A; scratch[2 * w + 1] = Bv; }
  __syncthreads();
  float s = 0.f;
  for (int i = nw - 1; i > w; --i) s = fmaf(scratch[2 * i], s, scratch[2 * i + 1]);
  __syncthreads();
  float ea = __shfl_down(A, 1, 64), eb = __shfl_down(Bv, 1, 64);
  if (l == 63) { ea = 1.f; eb = 0.f; }
  return fmaf(ea, s, eb);
}

// ---- bit-mask helpers ----------------------------------------------------------
// A threshold decision per sample is kept as a bit array in LDS (bit b of word w
// = sample 32w+b), written by wave ballots in the lane-strided view.
__device__ __forceinline__ void ballot_store(bool pred, uint32_t* bm, int word_base) {
  unsigned long long m = __ballot(pred);
  if (lane_id() == 0) { bm[word_base] = (uint32_t)m; bm[word_base + 1] = (uint32_t)(m >> 32); }
}

// all bits [s, s+len) set?  (bits beyond the array are stored as zero)
__device__ __forceinline__ bool bits_all_set(const uint32_t* bm, int s, int len, int nwords) {
  int pos = s, rem = len;
  while (rem > 0) {
    int w = pos >> 5, b = pos & 31;
    if (w >= nwords) return false;
    int take = min(32 - b, rem);
    uint32_t mask = (take == 32) ? 0xffffffffu : ((1u << take) - 1u);
    if (((bm[w] >> b) & mask) != mask) return false;
    pos += take; rem -= take;
  }
  return true;
}

// Intersect(min_n) on a bit array: counts runs of set bits that do not start at
// sample 0 and are at least min_n long; *first = start of the first such run.
// Word `w` is handled by thread w (call with w < nwords, others pass cnt=0).
__device__ __forceinline__ void intersect_word(const uint32_t* bm, int w, int nwords, int min_n,
                                               int* cnt, int* first) {
  uint32_t h = bm[w];
  uint32_t prev = (w == 0) ? 1u : (bm[w - 1] >> 31);  // sample -1 counts as "high": initial run excluded
  uint32_t starts = h & ~((h << 1) | prev);
  int c = 0, f = 0x7fffffff;
  while (starts) {
    int b = __ffs(starts) - 1;
    starts &= starts - 1;
    int s = 32 * w + b;
    if (min_n <= 1 || bits_all_set(bm, s + 1, min_n - 1, nwords)) {
      ++c;
      f = min(f, s);
    }
  }
  *cnt = c; *first = f;
}

// The same scan on the REVERSED trace (get_intracePileUp, src/dsp_routines.jl:79):
// runs that do not touch the last sample (n-1), at least min_n long, counted;
// *last_end = largest end index of such a run (or -1).
__device__ __forceinline__ void intersect_word_rev(const uint32_t* bm, int w, int nwords, int n, int min_n,
                                                   int* cnt, int* last_end) {
  uint32_t h = bm[w];
  uint32_t nextbit;
  // bit of sample 32w+32; the sample just past the end (index n) counts as "high"
  if (32 * w + 32 == n) nextbit = 1u;
  else nextbit = (w + 1 < nwords) ? (bm[w + 1] & 1u) : 0u;
  uint32_t hn = (h >> 1) | (nextbit << 31);
  if ((n >> 5) == w && (n & 31) != 0) hn |= 1u << ((n & 31) - 1);  // sample n inside this word
  uint32_t ends = h & ~hn;
  int c = 0, e_best = -1;
  while (ends) {
    int b = __ffs(ends) - 1;
    ends &= ends - 1;
    int e = 32 * w + b;
    if (e >= n) continue;
    int s = e - min_n + 1;
    if (s >= 0 && (min_n <= 1 || bits_all_set(bm, s, min_n - 1, nwords))) {
      ++c;
      e_best = max(e_best, e);
    }
  }
  *cnt = c; *last_end = e_best;
}

// parabola vertex through three points — reference src/interpolation.jl:8-10
__device__ __forceinline__ float extrema3points(float y1, float y2, float y3) {
  float a = y3 - 4.f * y2 + 3.f * y1;
  return y1 - a * a / (8.f * (y3 - 2.f * y2 + y1));
}

}  // namespace ldsp
