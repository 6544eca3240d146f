// run_scan.hpp — Intersect-style run scans on bit masks (one bit per sample, 32 samples per word): plain integer code shared
// by the gfx950 kernels and, compiled by g++, by tests/sanitize/run_scan_driver.cpp (which checks every form against a
// bit-by-bit scan).  RadiationDetectorDSP `_find_intersect_impl` = the scan of reference src/intersect_maximum.jl:41-56.
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define LDSP_RS __device__ __forceinline__
#else
#include <algorithm>
#define LDSP_RS static inline
namespace ldsp {
using std::max;
using std::min;
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
static inline int __ffs(uint32_t v) { return __builtin_ffs((int)v); }
static inline int __clz(uint32_t v) { return v ? __builtin_clz(v) : 32; }
}  // namespace ldsp
#endif

namespace ldsp {

// all bits [s, s+len) set?  (bits beyond the trace are stored as zero)
LDSP_RS bool bits_all_set(const uint32_t* bm, int s, int len, int nwords) {
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

// Intersect(min_n) on a bit array (RadiationDetectorDSP `_find_intersect_impl`, the
// scan of reference src/intersect_maximum.jl:41-56): counts runs of set bits that do
// not start at sample 0 and are at least min_n long; *first = start of the first one.
LDSP_RS void intersect_word(const uint32_t* bm, int w, int nwords, int min_n, int* cnt, int* first) {
  uint32_t h = bm[w];
  if (h == 0u) { *cnt = 0; *first = 0x7fffffff; return; }  // no run can start in an empty word (most words of a sparse mask)
  // a run of min_n >= 64 samples that starts in this word covers the whole next word: a flickering mask (threshold inside the
  // noise: the t0 trapezoid on a baseline) has several short runs per word and none of them needs a closer look
  if (min_n >= 64 && (w + 1 >= nwords || bm[w + 1] != 0xffffffffu)) { *cnt = 0; *first = 0x7fffffff; return; }
  uint32_t prev = (w == 0) ? 1u : (bm[w - 1] >> 31);  // sample -1 counts as "high": initial run excluded
  uint32_t starts = h & ~((h << 1) | prev);
  if (min_n >= 2) {  // cheap prune: the sample after a run start must be high too
    const uint32_t next = (w + 1 < nwords) ? (bm[w + 1] & 1u) : 0u;
    starts &= (h >> 1) | (next << 31);
  }
  int c = 0, f = 0x7fffffff;
  while (starts) {
    int b = __ffs(starts) - 1;
    starts &= starts - 1;
    int s = 32 * w + b;
    if (min_n <= 2 || bits_all_set(bm, s + 2, min_n - 2, nwords)) { ++c; f = min(f, s); }
  }
  *cnt = c; *first = f;
}
// The same scan on the REVERSED trace (get_intracePileUp, reference src/dsp_routines.jl:79):
// runs that do not touch the last sample n-1, at least min_n long; *last_end = largest end index.
LDSP_RS void intersect_word_rev(const uint32_t* bm, int w, int nwords, int n, int min_n, int* cnt, int* last_end) {
  uint32_t h = bm[w];
  if (h == 0u) { *cnt = 0; *last_end = -1; return; }
  uint32_t nextbit;
  if (32 * w + 32 == n) nextbit = 1u;  // the sample just past the end counts as "high"
  else nextbit = (w + 1 < nwords) ? (bm[w + 1] & 1u) : 0u;
  uint32_t hn = (h >> 1) | (nextbit << 31);
  if ((n >> 5) == w && (n & 31) != 0) hn |= 1u << ((n & 31) - 1);
  uint32_t ends = h & ~hn;
  int c = 0, e_best = -1;
  while (ends) {
    int b = __ffs(ends) - 1;
    ends &= ends - 1;
    int e = 32 * w + b;
    if (e >= n) continue;
    int s = e - min_n + 1;
    if (s >= 0 && (min_n <= 1 || bits_all_set(bm, s, min_n - 1, nwords))) { ++c; e_best = max(e_best, e); }
  }
  *cnt = c; *last_end = e_best;
}

// ---- loop-free forms of intersect_word / intersect_word_rev on words the caller has read: hm1, h, h1.. are the
// words w-1, w, w+1.. of the mask (zero outside the mask).
// bit p of the result: bits p .. p+n-1 of x are all set (1 <= n <= 64)
LDSP_RS unsigned long long runs_from(unsigned long long x, int n) {
  for (int have = 1; have < n;) { const int st = min(have, n - have); x &= x >> st; have += st; }
  return x;
}
// Intersect(min_n), min_n <= 97: the starts (bits of word w) of the runs that begin in word w — not at sample 0 — and last min_n samples
LDSP_RS uint32_t intersect_pre_mask(uint32_t hm1, uint32_t h, uint32_t h1, uint32_t h2, uint32_t h3, int w, int min_n) {
  const uint32_t prev = (w == 0) ? 1u : (hm1 >> 31);   // sample -1 counts as "high": a run that starts the trace is no crossing
  const uint32_t starts = h & ~((h << 1) | prev);
  if (starts == 0u) return 0u;
  const unsigned long long lo = (unsigned long long)h | ((unsigned long long)h1 << 32);
  if (min_n <= 32) return starts & (uint32_t)runs_from(lo, min_n);
  // a run of more than 32 samples leaves the word: only the last start of the word can be one
  const int b = 31 - __clz(starts);
  const unsigned long long hi = (unsigned long long)h2 | ((unsigned long long)h3 << 32);
  const unsigned long long l2 = b ? (lo >> b) | (hi << (64 - b)) : lo, u2 = hi >> b;   // the 128-bit window from the start on
  bool all;
  if (min_n <= 64) {
    const unsigned long long m = (min_n == 64) ? ~0ull : ((1ull << min_n) - 1ull);
    all = (l2 & m) == m;
  } else {
    const unsigned long long m = (1ull << (min_n - 64)) - 1ull;
    all = l2 == ~0ull && (u2 & m) == m;
  }
  return all ? (1u << b) : 0u;
}
// ... as count and first start
LDSP_RS void intersect_pre(uint32_t hm1, uint32_t h, uint32_t h1, uint32_t h2, uint32_t h3, int w, int min_n, int* cnt, int* first) {
  const uint32_t ok = intersect_pre_mask(hm1, h, h1, h2, h3, w, min_n);
  *cnt = __popc(ok);
  *first = ok ? 32 * w + __ffs(ok) - 1 : 0x7fffffff;
}
// the scan on the reversed trace (get_intracePileUp), min_n <= 32: runs of min_n samples that END in word w and do not touch
// sample n-1; count and the largest end index
LDSP_RS void intersect_rev_pre(uint32_t hm1, uint32_t h, uint32_t h1, int w, int n, int min_n, int* cnt, int* last_end) {
  *cnt = 0; *last_end = -1;
  const uint32_t nextbit = (32 * w + 32 == n) ? 1u : (h1 & 1u);   // the sample just past the end counts as "high"
  uint32_t hn = (h >> 1) | (nextbit << 31);
  if ((n >> 5) == w && (n & 31) != 0) hn |= 1u << ((n & 31) - 1);
  const uint32_t ends = h & ~hn;
  if (ends == 0u) return;
  unsigned long long x = (unsigned long long)hm1 | ((unsigned long long)h << 32);   // bit p of x: bits p-min_n+1 .. p all set
  for (int have = 1; have < min_n;) { const int st = min(have, min_n - have); x &= x << st; have += st; }
  const uint32_t ok = ends & (uint32_t)(x >> 32);
  if (ok) { *cnt = __popc(ok); *last_end = 32 * w + 31 - __clz(ok); }
}

}  // namespace ldsp
