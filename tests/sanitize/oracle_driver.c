/* Sanitizer driver of the CPU oracle (tests/test_sanitizers_cpu.py builds it with gcc -fsanitize=address,undefined together with
 * oracle/ldsp_oracle.c): runs dsp_icpc / the config-2 sub-chain on every ldsp_icpc_params block of file argv[1] and dsp_sipm
 * on every ldsp_sipm_params block of file argv[2], over synthetic traces that include flat, saturated, negative and
 * spiky ones, then the extractors and filters at their boundary arguments (windows at the trace ends, one-sample traces,
 * thresholds nothing reaches, more triggers than the capacity).  A sanitizer report makes the process fail. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ldsp.h"

int orc_icpc_ncols(void);
int orc_sipm_ncols(void);
int orc_dsp_icpc(const float* wf, long n, const ldsp_icpc_params* p, double* out, int* status, int nthreads);
int orc_icpc_pz_trap(const float* wf, long n, const ldsp_icpc_params* p, double* out);
int orc_dsp_sipm(const float* wf, long n, const ldsp_sipm_params* p, double* out_scalars, double* out_trig, int* counts, int cap,
                 int* status, int nthreads);
int orc_signalstats(const double* y, int n, int from, int until, double t_first, double dt, double* mean, double* sigma, double* slope, double* offset);
int orc_extremestats(const double* y, int n, int from, int until, double t_first, double dt, double* vmin, double* vmax, double* tmin, double* tmax);
double orc_thresholdstats_mad(const double* y, int n, double lo, double hi);
int orc_saturation(const double* y, int n, int from, int until, double low, double high, int out[4]);
int orc_get_wvf_maximum(const double* y, int n, int from, int until, double* out);
void orc_intersect(const double* y, int n, double t_first, double dt, double thr, int min_n, double* xout, int* mult);
int orc_intersect_maximum(const double* y, int n, double t_first, double dt, double thr, int min_n, int max_n, int cap, double* x,
                          double* x_high, double* x_tot, double* vmax);
int orc_multi_intersect(const double* y, int n, double t_first, double dt, const double* ratios, int K, int min_n, int half_n, int degree,
                        int rate, double* xout);
int orc_signal_estimator(const double* y, int n, double t_first, double dt, double t, int npts, int degree, double* out);
int orc_trap(const double* x, int n, int navg, int ngap, int navg2, double* y);
int orc_fir(const double* x, int n, const double* h, int m, double* y);
int orc_haar(const double* x, int n, int ds, double* y);
int orc_moving_window(const double* x, int n, int l, double* y);
int orc_moving_window_multi(const double* x, int n, int l, double* y);
int orc_derivative(const double* x, int n, double gain, double* y);
int orc_invcr(const double* x, int n, double c, double* y);

static unsigned long long rng = 88172645463325252ull;
static double urand(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0; }
static double nrand(void) { double s = 0; for (int i = 0; i < 12; ++i) s += urand(); return s - 6.0; }

static void traces(float* wf, int n, int L, int sipm) {
  for (int i = 0; i < n; ++i) {
    float* w = wf + (size_t)i * L;
    const int kind = i % 6;
    const double base = sipm ? 0.0 : 1000.0, amp = sipm ? 20.0 : 500.0 + 20000.0 * urand(), t0 = L * (0.35 + 0.05 * urand());
    for (int k = 0; k < L; ++k) {
      double v = base + (sipm ? 0.3 : 3.0) * nrand();
      if (!sipm && k > t0) v += amp * (1.0 - exp(-(k - t0) / 10.0)) * exp(-(k - t0) * 16.0 / 500000.0);
      w[k] = (float)v;
    }
    if (sipm) for (int q = 0; q < 6; ++q) { int p0 = (int)(urand() * (L - 200)); for (int j = 0; j < 150; ++j) w[p0 + j] += (float)(amp * exp(-j / 30.0)); }
    if (kind == 1) for (int k = 0; k < L; ++k) w[k] = 1234.0f;                               /* flat */
    if (kind == 2) for (int k = 0; k < L; ++k) w[k] = w[k] > 3000.f ? 3000.f : w[k];         /* flat-topped */
    if (kind == 3) for (int k = 0; k < L; ++k) w[k] = -w[k];                                 /* negative */
    if (kind == 4) { w[0] += 5000.f; w[L - 1] += 5000.f; w[L / 2] -= 5000.f; }               /* spikes at the ends */
  }
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int n = 12;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  ldsp_icpc_params p;
  int nblk = 0;
  while (fread(&p, 1, sizeof p, f) == sizeof p) {
    if (p.L < 8 || p.L > LDSP_MAX_L) { printf("icpc %d skipped (L)\n", nblk++); continue; }
    float* wf = (float*)malloc(sizeof(float) * (size_t)n * p.L);
    traces(wf, n, p.L, 0);
    double* out = (double*)malloc(sizeof(double) * (size_t)n * orc_icpc_ncols());
    int status[12];
    const int rc = orc_dsp_icpc(wf, n, &p, out, status, 2);
    double o2[24];
    const int rc2 = orc_icpc_pz_trap(wf, n, &p, o2);
    printf("icpc %d rc %d %d\n", nblk++, rc, rc2);
    free(out); free(wf);
  }
  fclose(f);
  f = fopen(argv[2], "rb");
  if (!f) return 2;
  ldsp_sipm_params s;
  nblk = 0;
  while (fread(&s, 1, sizeof s, f) == sizeof s) {
    if (s.L < 8 || s.L > LDSP_MAX_L) { printf("sipm %d skipped (L)\n", nblk++); continue; }
    float* wf = (float*)malloc(sizeof(float) * (size_t)n * s.L);
    traces(wf, n, s.L, 1);
    const int cap = 5;   /* fewer slots than triggers: the capacity path */
    double* sc = (double*)malloc(sizeof(double) * (size_t)n * orc_sipm_ncols());
    double* tr = (double*)malloc(sizeof(double) * (size_t)n * 16 * cap);
    int counts[48], status[12];
    const int rc = orc_dsp_sipm(wf, n, &s, sc, tr, counts, cap, status, 2);
    printf("sipm %d rc %d\n", nblk++, rc);
    free(tr); free(sc); free(wf);
  }
  fclose(f);
  /* extractors and filters at their boundary arguments */
  enum { N = 257 };
  double y[N], z[N + 8], o[4], a, b, c, d;
  int it[4], mult;
  for (int k = 0; k < N; ++k) y[k] = sin(k * 0.1) * 100.0 + nrand();
  const int wins[][2] = {{0, 0}, {0, N - 1}, {N - 1, N - 1}, {5, 4}, {-1, 3}, {3, N}, {100, 200}};
  for (unsigned q = 0; q < sizeof wins / sizeof wins[0]; ++q) {
    (void)orc_signalstats(y, N, wins[q][0], wins[q][1], 0.0, 16.0, &a, &b, &c, &d);
    (void)orc_extremestats(y, N, wins[q][0], wins[q][1], 0.0, 16.0, &a, &b, &c, &d);
    (void)orc_saturation(y, N, wins[q][0], wins[q][1], -50.0, 50.0, it);
    (void)orc_get_wvf_maximum(y, N, wins[q][0], wins[q][1], &a);
  }
  for (int nn = 0; nn <= 3; ++nn) {
    (void)orc_thresholdstats_mad(y, nn, -1e9, 1e9);
    orc_intersect(y, nn, 0.0, 1.0, 10.0, 1, &a, &mult);
    (void)orc_intersect_maximum(y, nn, 0.0, 1.0, 10.0, 1, 3, 2, o, o, o, o);
    (void)orc_trap(y, nn, 1, 1, 1, z); (void)orc_haar(y, nn, 2, z); (void)orc_moving_window(y, nn, 2, z); (void)orc_moving_window_multi(y, nn, 2, z);
    (void)orc_derivative(y, nn, 1.0, z); (void)orc_invcr(y, nn, 1e-3, z);
  }
  (void)orc_thresholdstats_mad(y, N, 1e9, 2e9);      /* nothing valid */
  double x4[8], xh[8], xt[8], vm[8];
  (void)orc_intersect_maximum(y, N, 0.0, 1.0, 0.0, 1, 1000, 2, x4, xh, xt, vm);   /* more triggers than slots */
  orc_intersect(y, N, 0.0, 1.0, 1e9, 4, &a, &mult);
  const double ratios[5] = {0.1, 0.5, 0.9, 0.95, 0.99};
  double xo[5];
  for (int half = 1; half <= 3; ++half) (void)orc_multi_intersect(y, N, 0.0, 1.0, ratios, 5, 2, half, 1, 3, xo);
  (void)orc_multi_intersect(y + 200, 40, 0.0, 1.0, ratios, 5, 6, 3, 2, 2, xo);
  const double ts[] = {-100.0, 0.0, 3.3, 255.9, 256.0, 1e6};
  for (unsigned q = 0; q < 6; ++q) (void)orc_signal_estimator(y, N, 0.0, 1.0, ts[q], 5, 2, &a);
  (void)orc_signal_estimator(y, 3, 0.0, 1.0, 1.0, 5, 2, &a);
  double h[9] = {1, 2, 3, 4, 5, 4, 3, 2, 1};
  (void)orc_fir(y, N, h, 9, z); (void)orc_fir(y, 5, h, 9, z); (void)orc_trap(y, N, 100, 100, 100, z); (void)orc_trap(y, N, 1, 0, 1, z);
  (void)orc_haar(y, N, 4, z); (void)orc_moving_window(y, N, 300, z); (void)orc_moving_window_multi(y, N, 64, z);
  printf("done\n");
  return 0;
}
