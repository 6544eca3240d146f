// ASan/UBSan + correctness driver for csrc/run_scan.hpp: every form of the Intersect run scan (word-by-word: intersect_word /
// intersect_word_rev; loop-free on preloaded words: intersect_pre / intersect_rev_pre) against a bit-by-bit scan of the same
// mask, on random masks of every density with long runs, ragged ends and every run length the kernels send.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "run_scan.hpp"

using namespace ldsp;

static bool bit(const std::vector<uint32_t>& bm, int i) { return i >= 0 && i < 32 * (int)bm.size() && ((bm[i >> 5] >> (i & 31)) & 1u); }

// runs of >= min_n set bits that do not start at sample 0: how many start in word w, and the first such start
static void ref_fwd(const std::vector<uint32_t>& bm, int w, int min_n, int* c, int* f) {
  *c = 0; *f = 0x7fffffff;
  for (int s = 32 * w; s < 32 * w + 32; ++s) {
    if (s == 0 || !bit(bm, s) || bit(bm, s - 1)) continue;
    int len = 0;
    while (bit(bm, s + len)) ++len;
    if (len >= min_n) { if (!*c) *f = s; ++*c; }
  }
}
// runs of >= min_n set bits that do not touch sample n-1: how many END in word w, and the largest such end
static void ref_rev(const std::vector<uint32_t>& bm, int w, int n, int min_n, int* c, int* e_best) {
  *c = 0; *e_best = -1;
  for (int e = 32 * w; e < 32 * w + 32 && e < n - 1; ++e) {
    if (!bit(bm, e) || bit(bm, e + 1)) continue;
    int len = 0;
    while (bit(bm, e - len)) ++len;
    if (len >= min_n) { ++*c; *e_best = e; }
  }
}

int main() {
  srand(12345);
  long bad = 0, tot = 0;
  for (int it = 0; it < 40000; ++it) {
    const int nw = 1 + rand() % 9;
    std::vector<uint32_t> bm(nw);
    const int dens[5] = {3, 50, 90, 97, 100};
    const int p = dens[rand() % 5];
    for (int w = 0; w < nw; ++w) { uint32_t v = 0; for (int b = 0; b < 32; ++b) if (rand() % 100 < p) v |= 1u << b; bm[w] = v; }
    if (rand() % 3 == 0) { const int a = rand() % (32 * nw), l = rand() % 140; for (int i = a; i < a + l && i < 32 * nw; ++i) bm[i >> 5] |= 1u << (i & 31); }
    const int n = 32 * nw - (rand() % 3 == 0 ? rand() % 32 : 0);                 // samples n .. are stored as zero
    for (int i = n; i < 32 * nw; ++i) bm[i >> 5] &= ~(1u << (i & 31));
    const int mn = 1 + rand() % 120, mr = 1 + rand() % 40;
    auto g = [&](int i) { return (i >= 0 && i < nw) ? bm[i] : 0u; };
    for (int w = 0; w < nw; ++w) {
      int c0, f0, c1, f1, c2, f2;
      ref_fwd(bm, w, mn, &c0, &f0);
      intersect_word(bm.data(), w, nw, mn, &c1, &f1);
      ++tot;
      if (c0 != c1 || (c0 && f0 != f1)) { if (bad++ < 5) printf("intersect_word min_n=%d w=%d: %d/%d vs %d/%d\n", mn, w, c1, f1, c0, f0); }
      if (mn <= 97) {
        intersect_pre(g(w - 1), g(w), g(w + 1), g(w + 2), g(w + 3), w, mn, &c2, &f2);
        ++tot;
        if (c0 != c2 || (c0 && f0 != f2)) { if (bad++ < 5) printf("intersect_pre min_n=%d w=%d: %d/%d vs %d/%d\n", mn, w, c2, f2, c0, f0); }
      }
      if (32 * w >= n) continue;
      ref_rev(bm, w, n, mr, &c0, &f0);
      intersect_word_rev(bm.data(), w, nw, n, mr, &c1, &f1);
      ++tot;
      if (c0 != c1 || f0 != f1) { if (bad++ < 5) printf("intersect_word_rev min_n=%d n=%d w=%d: %d/%d vs %d/%d\n", mr, n, w, c1, f1, c0, f0); }
      if (mr <= 32) {
        intersect_rev_pre(g(w - 1), g(w), g(w + 1), w, n, mr, &c2, &f2);
        ++tot;
        if (c0 != c2 || f0 != f2) { if (bad++ < 5) printf("intersect_rev_pre min_n=%d n=%d w=%d: %d/%d vs %d/%d\n", mr, n, w, c2, f2, c0, f0); }
      }
    }
  }
  printf("run_scan: %ld mismatches in %ld checks\n", bad, tot);
  return bad != 0;
}
