/*
 * ldsp_oracle.c — CPU restatement (plain C; float64, or the reference's Float32-input typing with -DORC_F32) of the LegendDSP.jl
 * dsp_icpc / dsp_sipm hot path.  TEST INFRASTRUCTURE ONLY: it is the checker
 * for the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
 * leg).  Nothing in the product (legenddsp.jl_amd/) may call, link or import it.
 *
 * Pinning status
 *  - rows a1-a17 of SURVEY.md §8(a) (code that lives in /root/reference/src)
 *    are restated line by line and checked against every known-answer value in
 *    the reference's own tests (tests/test_oracle_golden.py): PINNED.
 *  - rows a18-a28 live in the un-vendored dependency RadiationDetectorDSP.jl
 *    (compat 0.2.17, Project.toml:35; no Manifest => no exact pin; source not
 *    in /root/reference).  Their published algorithm is restated here under
 *    the assumptions A1-A7 listed in DESIGN.md: PARITY UNPINNED for the numeric
 *    output of InvCRFilter, TrapezoidalChargeFilter, CUSP/ZACChargeFilter,
 *    SavitzkyGolayFilter, SignalEstimator, signalstats, and therefore for the
 *    full dsp_icpc / dsp_sipm tables (the reference's own tests for those are
 *    smoke tests: test/test_dsp_icpc.jl:164-200, test/test_dsp_sipm.jl:70-109).
 *
 * Every function cites the reference file:line it follows.  All indices are
 * 0-based here; the reference is 1-based.
 */
#include "../include/ldsp.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_ERR_WINDOW (-2)
#define ORC_ERR_ARG (-1)

/* `real`: the element type of the traces and of everything the reference computes in float(eltype(input)).  The default build
 * is Float64 throughout — the checker of tests/.  -DORC_F32 (libldsp_oracle_f32.so) restates what the reference does with
 * Float32 input (SURVEY §8 "int/fp": samples, filter states and the sums of Y in Float32; the time axis, X sums, X*Y sums,
 * filter taps' derivation and estimator weights in Float64, src/tailstats.jl:39-43): tests/parity.py measures with it how far a
 * correct Float32 chain lies from the Float64 one, column by column, instead of arguing that envelope. */
#ifdef ORC_F32
typedef float real;
#define RLOG(v) logf(v)
#else
typedef double real;
#define RLOG(v) log(v)
#endif

/* Julia round(Int, x): round half to even. Default FP rounding mode. */
static inline long rnd(double x) { return (long)nearbyint(x); }

/* ------------------------------------------------------------------------ */
/* signalstats — RadiationDetectorDSP (not in /root/reference); restated as
 * src/tailstats.jl:22-72 without the log and with the offset line un-commented
 * (tailstats.jl:62), per assumption A6. */
int orc_signalstats(const real* y, int n, int from, int until, double t_first, double dt,
                    double* mean, double* sigma, double* slope, double* offset) {
  if (!(0 <= from && from <= until && until <= n - 1)) return ORC_ERR_WINDOW;
  double sx = 0, sxx = 0, sxy = 0; real sy = 0, syy = 0;   /* src/tailstats.jl:39-43: sums of Y in float(eltype(Y)), of X and X*Y in Float64 */
  for (int i = from; i <= until; ++i) {
    double x = t_first + i * dt; real v = y[i];
    sx += x; sxx = fma(x, x, sxx);
    sy += v; syy = fma(v, v, syy);
    sxy = fma(x, v, sxy);
  }
  double inv_n = 1.0 / (double)(until - from + 1);
  double mx = sx * inv_n, my = sy * inv_n;
  double vx = sxx * inv_n - mx * mx, vy = syy * inv_n - my * my;
  double cxy = sxy * inv_n - mx * my;
  double sl = cxy / vx;
  if (vy < 0) vy = 0;
  *mean = my; *sigma = sqrt(vy); *slope = sl; *offset = my - sl * mx;
  return ORC_OK;
}

/* tailstats — src/tailstats.jl:13-72 */
int orc_tailstats(const real* y, int n, int from, int until, double t_first, double dt,
                  double* mean, double* sigma, double* tau) {
  if (!(0 <= from && from <= until && until <= n - 1)) return ORC_ERR_WINDOW;
  for (int i = from; i <= until; ++i)
    if (y[i] <= 0) { *mean = 0; *sigma = 0; *tau = 0; return ORC_OK; } /* :27-33 */
  double sx = 0, sxx = 0, sxy = 0; real sy = 0, syy = 0;
  for (int i = from; i <= until; ++i) {
    double x = t_first + i * dt; real v = RLOG(y[i]);
    sx += x; sxx = fma(x, x, sxx);
    sy += v; syy = fma(v, v, syy);
    sxy = fma(x, v, sxy);
  }
  double inv_n = 1.0 / (double)(until - from + 1);
  double mx = sx * inv_n, my = sy * inv_n;
  double vx = sxx * inv_n - mx * mx, vy = syy * inv_n - my * my;
  double cxy = sxy * inv_n - mx * my;
  double sl = cxy / vx;
  if (vy < 0) vy = 0;
  *mean = my; *sigma = sqrt(vy); *tau = -1.0 / sl;
  return ORC_OK;
}

/* extremestats — src/extremestats.jl:25-40 (findmin/findmax: first occurrence) */
int orc_extremestats(const real* y, int n, int from, int until, double t_first, double dt,
                     double* vmin, double* vmax, double* tmin, double* tmax) {
  if (!(0 <= from && from <= until && until <= n - 1)) return ORC_ERR_WINDOW;
  int imin = from, imax = from;
  for (int i = from + 1; i <= until; ++i) {
    if (y[i] < y[imin]) imin = i;
    if (y[i] > y[imax]) imax = i;
  }
  *vmin = y[imin]; *vmax = y[imax];
  *tmin = t_first + imin * dt; *tmax = t_first + imax * dt;
  return ORC_OK;
}

/* thresholdstats — src/thresholdstats.jl:19-41 (n == 0 -> NaN via inv(0)) */
double orc_thresholdstats(const real* y, int n, double lo, double hi) {
  real sy = 0, syy = 0; long cnt = 0;
  for (int i = 0; i < n; ++i) {
    int inc = (lo <= y[i] && y[i] <= hi);
    real v = inc ? y[i] : (real)0;
    sy += v; syy = fma(v, v, syy); cnt += inc;
  }
  double inv_n = 1.0 / (double)cnt;
  double m = sy * inv_n;
  double var = syy * inv_n - m * m;
  if (!(var > 0)) var = (var != var) ? var : 0.0; /* max(var, 0), NaN stays NaN */
  return sqrt(var);
}

static int cmp_dbl(const void* a, const void* b) {
  real x = *(const real*)a, y = *(const real*)b;
  return (x > y) - (x < y);
}
/* Statistics.median: mean of the two middle order statistics for even counts */
static real median_inplace(real* v, int m) {
  qsort(v, (size_t)m, sizeof(real), cmp_dbl);
  return (m & 1) ? v[m / 2] : 0.5 * (v[m / 2 - 1] + v[m / 2]);
}
/* thresholdstats_mad — src/thresholdstats.jl:61-71 */
double orc_thresholdstats_mad(const real* y, int n, double lo, double hi) {
  real* f = (real*)malloc(sizeof(real) * (size_t)(n > 0 ? n : 1));
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (lo <= y[i] && y[i] <= hi) f[m++] = y[i];
  if (m == 0) { free(f); return 0.0; }
  real med = median_inplace(f, m);
  for (int i = 0; i < m; ++i) f[i] = fabs(f[i] - med);
  double r = 1.4826 * (double)median_inplace(f, m);
  free(f);
  return r;
}

/* saturation — src/saturation.jl:28-65 */
int orc_saturation(const real* y, int n, int from, int until, double low, double high, int out[4]) {
  if (!(0 <= from && from <= until && until <= n - 1)) return ORC_ERR_WINDOW;
  int n_low = 0, n_high = 0, cons_low = 0, cons_high = 0, c_low = 0, c_high = 0;
  for (int i = from; i <= until; ++i) {
    if (y[i] == low) {
      n_low++; c_low++;
      if (c_high > cons_high) cons_high = c_high;
      c_high = 0;
    } else if (y[i] == high) {
      n_high++; c_high++;
      if (c_low > cons_low) cons_low = c_low;
      c_low = 0;
    } else {
      if (c_low > cons_low) cons_low = c_low;
      c_low = 0;
      if (c_high > cons_high) cons_high = c_high;
      c_high = 0;
    }
  }
  if (c_low > cons_low) cons_low = c_low;
  if (c_high > cons_high) cons_high = c_high;
  out[0] = n_low; out[1] = n_high; out[2] = cons_low; out[3] = cons_high;
  return ORC_OK;
}

/* extrema3points — src/interpolation.jl:8-10 */
static inline real extrema3points(real y1, real y2, real y3) {
  real a = (y3 - 4 * y2 + 3 * y1);
  return y1 - a * a / (8 * (y3 - 2 * y2 + y1));
}
/* argmax (first occurrence) + parabola if strictly interior — src/interpolation.jl:30-46 */
static real window_max_interp(const real* y, int from, int until) {
  int im = from;
  for (int i = from + 1; i <= until; ++i)
    if (y[i] > y[im]) im = i;
  if (from < im && im < until) return extrema3points(y[im - 1], y[im], y[im + 1]);
  return y[im];
}
int orc_get_wvf_maximum(const real* y, int n, int from, int until, double* out) {
  if (!(0 <= from && from <= until && until <= n - 1)) return ORC_ERR_WINDOW;
  *out = window_max_interp(y, from, until);
  return ORC_OK;
}

/* Intersect(mintot)(wf, thr) — RadiationDetectorDSP `_find_intersect_impl`
 * (call sites src/dsp_routines.jl:18,35,74; src/multi_intersect.jl:97).
 * Restated as the scan of src/intersect_maximum.jl:41-56 / src/multi_intersect.jl:53-72
 * keeping only the first confirmed crossing and counting all.
 * Time axis may be non-uniform-free: x(i) = t_first + i*dt. */
void orc_intersect(const real* y, int n, double t_first, double dt, double thr, int min_n,
                   double* xout, int* mult) {
  if (n <= 0) { *xout = NAN; *mult = 0; return; }
  int cand = 1, pos = 1, cnt = (y[0] >= thr) ? min_n + 1 : 0, nint = 0;
  for (int i = 0; i < n; ++i) {
    int high = y[i] >= thr;
    if (high && cnt == 0) cand = i;
    cnt = high ? cnt + 1 : 0;
    if (cnt == min_n) {
      nint++;
      if (nint == 1) pos = cand;
    }
  }
  if (nint > 0 && pos > 0) {
    double xl = t_first + (pos - 1) * dt, xr = t_first + pos * dt;
    real yl = y[pos - 1], yr = y[pos];
    *xout = (thr - yl) * (xr - xl) / (yr - yl) + xl;
  } else {
    *xout = NAN;
  }
  *mult = nint;
}

/* IntersectMaximum — src/intersect_maximum.jl:24-119.  Returns multiplicity;
 * fills at most cap entries of each output. */
int orc_intersect_maximum(const real* y, int n, double t_first, double dt, double thr,
                          int min_n, int max_n, int cap, double* x, double* x_high,
                          double* x_tot, double* vmax) {
  if (n <= 0) return 0; /* :30-38 */
  int cand = 1, cnt = (y[0] > thr) ? min_n + 1 : 0; /* strict '>' :43 */
  int nup = 0;
  int* ups = (int*)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    int high = y[i] >= thr; /* '>=' :47 */
    if (high && cnt == 0) cand = i;
    cnt = high ? cnt + 1 : 0;
    if (cnt == min_n && cand > 0) ups[nup++] = cand; /* :53 */
  }
  for (int k = 0; k < nup && k < cap; ++k) {
    int up = ups[k];
    double xl = t_first + (up - 1) * dt, xr = t_first + up * dt;
    real yl = y[up - 1], yr = y[up];
    double xi = (thr - yl) * (xr - xl) / (yr - yl) + xl; /* :71 */
    int from = up - 2 > 0 ? up - 2 : 0;                  /* :75 */
    int until = up + max_n < n - 1 ? up + max_n : n - 1; /* :76 */
    real mx = window_max_interp(y, from, until);       /* :79-85 */
    int down = -1;
    for (int j = up + min_n; j < n; ++j) /* :90-95 */
      if (y[j] < thr) { down = j; break; }
    double xh;
    if (down > 0) { /* :99 */
      double dxl = t_first + (down - 1) * dt, dxr = t_first + down * dt;
      real dyl = y[down - 1], dyr = y[down];
      xh = (thr - dyl) * (dxr - dxl) / (dyr - dyl) + dxl;
    } else {
      xh = t_first + (n - 1) * dt; /* :106 */
    }
    if (x) x[k] = xi;
    if (x_high) x_high[k] = xh;
    if (x_tot) x_tot[k] = xh - xi;
    if (vmax) vmax[k] = mx;
  }
  free(ups);
  return nup;
}

/* ------------------------------------------------------------------------ */
/* Least-squares polynomial machinery (RadiationDetectorDSP `_lsq_fit_matrix`;
 * convention visible at src/multi_intersect.jl:80-84,115-123: c_j = sum_i
 * A[i,j] y[i], yhat = sum_j c_j x^j  =>  A = V (V'V)^-1).  The fitted
 * polynomial is basis independent, so it is evaluated here in a centred,
 * scaled basis u = (i - c)/s for conditioning.  B[i*(d+1)+j]: yhat(u) = sum_i
 * y_i sum_j B[i][j] u^j. */
static int lsq_basis(int npts, int degree, double* B, double* c_out, double* s_out) {
  if (npts < 1 || degree < 0 || degree > 12) return ORC_ERR_ARG;
  int d1 = degree + 1;
  double c = 0.5 * (npts - 1), s = c > 1 ? c : 1.0;
  long double M[13][26];
  for (int a = 0; a < d1; ++a)
    for (int b = 0; b < 2 * d1; ++b) M[a][b] = 0;
  for (int i = 0; i < npts; ++i) {
    long double u = ((long double)i - c) / s, pa = 1;
    for (int a = 0; a < d1; ++a) {
      long double pb = pa;
      for (int b = 0; b < d1; ++b) { M[a][b] += pb; pb *= u; }
      pa *= u;
    }
  }
  for (int a = 0; a < d1; ++a) M[a][d1 + a] = 1;
  /* Gauss-Jordan with partial pivoting; singular (npts <= degree) -> pseudo
   * inverse is not needed by any call site: report error. */
  for (int col = 0; col < d1; ++col) {
    int piv = col;
    for (int r = col + 1; r < d1; ++r)
      if (fabsl(M[r][col]) > fabsl(M[piv][col])) piv = r;
    if (fabsl(M[piv][col]) < 1e-30L) return ORC_ERR_ARG;
    if (piv != col)
      for (int b = 0; b < 2 * d1; ++b) { long double t = M[col][b]; M[col][b] = M[piv][b]; M[piv][b] = t; }
    long double inv = 1 / M[col][col];
    for (int b = 0; b < 2 * d1; ++b) M[col][b] *= inv;
    for (int r = 0; r < d1; ++r)
      if (r != col) {
        long double f = M[r][col];
        if (f != 0)
          for (int b = 0; b < 2 * d1; ++b) M[r][b] -= f * M[col][b];
      }
  }
  for (int i = 0; i < npts; ++i) {
    long double u = ((long double)i - c) / s;
    for (int j = 0; j < d1; ++j) {
      long double acc = 0, pa = 1;
      for (int a = 0; a < d1; ++a) { acc += pa * M[a][d1 + j]; pa *= u; }
      B[i * d1 + j] = (double)acc;
    }
  }
  *c_out = c; *s_out = s;
  return ORC_OK;
}

/* SavitzkyGolayFilter(length, degree, derivative) coefficients
 * (RadiationDetectorDSP; call sites src/dsp_icpc.jl:181-185, src/dsp_sipm.jl:99).
 * Assumption A2: npts odd, LSQ polynomial over the window, derivative per
 * sample at the window centre.  Returned as a true-convolution kernel h
 * (h[j] multiplies x[k+npts-1-j]). */
int orc_sg_coeffs(int npts, int degree, int deriv, double* h) {
  if (npts < 1 || !(npts & 1) || deriv < 0 || deriv > degree || npts > 4096) return ORC_ERR_ARG;
  if (npts <= degree) return ORC_ERR_ARG;
  double* B = (double*)malloc(sizeof(double) * (size_t)npts * (size_t)(degree + 1));
  double c, s;
  int rc = lsq_basis(npts, degree, B, &c, &s);
  if (rc) { free(B); return rc; }
  double fact = 1;
  for (int k = 2; k <= deriv; ++k) fact *= k;
  double sc = fact / pow(s, deriv);
  for (int i = 0; i < npts; ++i) h[npts - 1 - i] = B[i * (degree + 1) + deriv] * sc;
  free(B);
  return ORC_OK;
}

/* SignalEstimator(PolynomialDNI(degree, length))(wf, t) — RadiationDetectorDSP
 * (call sites src/dsp_icpc.jl:157-177, src/dsp_routines.jl:56-60).
 * Assumption A3: LSQ polynomial of `degree` over the npts samples nearest-centred
 * on t, window clamped inside the trace, evaluated at the fractional sample
 * position of t (itself clamped to the trace). */
int orc_signal_estimator(const real* y, int n, double t_first, double dt, double t,
                         int npts, int degree, double* out) {
  if (npts > n || npts <= degree || npts < 1) { *out = NAN; return ORC_ERR_ARG; }
  double p = (t - t_first) / dt;
  if (!(p == p)) { *out = NAN; return ORC_OK; }
  if (p < 0) p = 0;
  if (p > n - 1) p = n - 1;
  long i0 = (long)ceil(p - 0.5 * npts);
  if (i0 < 0) i0 = 0;
  if (i0 > n - npts) i0 = n - npts;
  /* small per-thread cache of fit bases keyed by (npts, degree) */
  enum { NC = 4, MAXP = 1024 };
  static __thread int c_npts[NC] = {-1, -1, -1, -1}, c_deg[NC], c_next = 0;
  static __thread double c_B[NC][MAXP * 13], c_c[NC], c_s[NC];
  if (npts > MAXP || degree > 12) return ORC_ERR_ARG;
  int e = -1;
  for (int k = 0; k < NC; ++k)
    if (c_npts[k] == npts && c_deg[k] == degree) e = k;
  if (e < 0) {
    e = c_next; c_next = (c_next + 1) % NC;
    c_npts[e] = -1;
    int rc = lsq_basis(npts, degree, c_B[e], &c_c[e], &c_s[e]);
    if (rc) { *out = NAN; return rc; }
    c_npts[e] = npts; c_deg[e] = degree;
  }
  double u = (p - (double)i0 - c_c[e]) / c_s[e], acc = 0;
  int d1 = degree + 1;
  for (int i = 0; i < npts; ++i) {
    double w = 0, pu = 1;
    for (int j = 0; j < d1; ++j) { w += c_B[e][i * d1 + j] * pu; pu *= u; }
    acc += w * y[i0 + i];
  }
  *out = acc;
  return ORC_OK;
}

/* MultiIntersect — src/multi_intersect.jl:36-104.  Returns 0, or ORC_ERR_WINDOW
 * where the reference's boundary @assert (:75-78) fires. */
int orc_multi_intersect(const real* y, int n, double t_first, double dt, const double* ratios,
                        int K, int min_n, int half_n, int degree, int rate, double* xout) {
  for (int k = 0; k < K; ++k) xout[k] = 0;
  if (n <= 0) return ORC_OK; /* :50 */
  real ymax = y[0];
  for (int i = 1; i < n; ++i)
    if (y[i] > ymax) ymax = y[i];
  double* thr = (double*)malloc(sizeof(double) * (size_t)K);
  int* ipos = (int*)malloc(sizeof(int) * (size_t)K);
  for (int k = 0; k < K; ++k) { thr[k] = ratios[k] * ymax; ipos[k] = 1; }
  int cand = 1, cnt = (y[0] >= thr[0]) ? min_n + 1 : 0, ic = 0, i = 0;
  while (i < n && ic < K) { /* :58-72 */
    int high = y[i] >= thr[ic];
    if (high && cnt == 0) cand = i;
    cnt = high ? cnt + 1 : 0;
    int found = (cnt == min_n);
    int pos = found ? cand : ipos[ic];
    ipos[ic] = pos;
    i = found ? pos : i + 1; /* rewind :69 */
    if (found) { ic++; cnt = 0; }
  }
  int rc = ORC_OK;
  if (!(ipos[0] - half_n >= 0) || !(ipos[K - 1] + half_n - 1 <= n - 1)) rc = ORC_ERR_WINDOW;
  if (rc == ORC_OK) {
    int w = 2 * half_n, m = 2 * half_n * rate, d1 = degree + 1;
    double* B = (double*)malloc(sizeof(double) * (size_t)w * (size_t)d1);
    real* yup = (real*)malloc(sizeof(real) * (size_t)m);
    double c, s;
    if (lsq_basis(w, degree, B, &c, &s)) rc = ORC_ERR_ARG;
    for (int k = 0; k < K && rc == ORC_OK; ++k) {
      int from = ipos[k] - half_n, to = ipos[k] + half_n - 1;
      if (from < 0 || to > n - 1) { rc = ORC_ERR_WINDOW; break; }
      for (int q = 0; q < m; ++q) { /* x_up = range(0, 2n-1, m) :81 */
        double xu = (m > 1) ? (double)q * (double)(w - 1) / (double)(m - 1) : 0.0;
        double u = (xu - c) / s, acc = 0;
        for (int a = 0; a < w; ++a) {
          double wgt = 0, pu = 1;
          for (int j = 0; j < d1; ++j) { wgt += B[a * d1 + j] * pu; pu *= u; }
          acc += wgt * y[from + a];
        }
        yup[q] = acc;
      }
      /* _x_axis = range(X[from], X[to], m) :91 ; Intersect with min_n = 1 :97 */
      double x0 = t_first + from * dt, x1 = t_first + to * dt;
      double dtu = (m > 1) ? (x1 - x0) / (double)(m - 1) : 0.0;
      int mult;
      orc_intersect(yup, m, x0, dtu, thr[k], 1, &xout[k], &mult);
    }
    free(B); free(yup);
  }
  free(thr); free(ipos);
  return rc;
}

/* ------------------------------------------------------------------------ */
/* Filters.  All return the output length (>=0) or a negative error. */

/* InvCRFilter(tau) — RadiationDetectorDSP biquad b=(k,-1,0), a=(1,-1,0),
 * k = 1 + dt/tau, zero initial state (assumption A5):
 * y[n] = y[n-1] + k x[n] - x[n-1]. */
int orc_invcr(const real* x, int n, double c, real* y) {
  real k = (real)(1.0 + c), yp = 0, xp = 0;
  for (int i = 0; i < n; ++i) { yp = yp + k * x[i] - xp; xp = x[i]; y[i] = yp; }
  return n;
}
/* IntegratorFilter(gain) — biquad b=(g,0,0), a=(1,-1,0) */
int orc_integrator(const real* x, int n, double gain, real* y) {
  real acc = 0;
  for (int i = 0; i < n; ++i) { acc += gain * x[i]; y[i] = acc; }
  return n;
}
/* TrapezoidalChargeFilter — RadiationDetectorDSP (assumption A1):
 * out[k] = mean(x[k+navg+ngap .. +navg2-1]) - mean(x[k .. k+navg-1]), valid mode. */
int orc_trap(const real* x, int n, int navg, int ngap, int navg2, real* y) {
  int flen = navg + ngap + navg2;
  if (navg < 1 || navg2 < 1 || ngap < 0) return ORC_ERR_ARG;
  int nout = n - flen + 1;
  if (nout < 1) return ORC_ERR_WINDOW;
  real s1 = 0, s2 = 0;
  for (int i = 0; i < navg; ++i) s1 += x[i];
  for (int i = 0; i < navg2; ++i) s2 += x[navg + ngap + i];
  real i1 = (real)(1.0 / navg), i2 = (real)(1.0 / navg2);
  y[0] = s2 * i2 - s1 * i1;
  for (int k = 1; k < nout; ++k) {
    s1 += x[k + navg - 1] - x[k - 1];
    s2 += x[k + flen - 1] - x[k + navg + ngap - 1];
    y[k] = s2 * i2 - s1 * i1;
  }
  return nout;
}
/* ---- FFT form of the long FIR filters (CPU-baseline leg only).  RadiationDetectorDSP's ConvolutionFilter offers direct and FFT
 * convolution ("upstream uses FFT or direct convolution"); which one dsp_icpc's 2375-tap CUSP / ZAC filters take
 * upstream is not visible in /root/reference.  orc_set_fir_mode(1) makes orc_fir evaluate filters of more than 64 taps as ONE
 * zero-padded complex radix-2 FFT of the trace, a product with the taps' transform and an inverse FFT (same valid-mode output
 * to ~1e-12 relative in double): bench.py times both forms and labels them.  The checker (tests/) uses the direct form. */
static int g_fir_mode = 0;
void orc_set_fir_mode(int m) { g_fir_mode = m; }
static void fft_c2(double* re, double* im, int n, int inverse) {   /* in place, n a power of two */
  for (int i = 1, j = 0; i < n; ++i) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
  }
  for (int len = 2; len <= n; len <<= 1) {
    const double ang = (inverse ? 2.0 : -2.0) * M_PI / len;
    const double wr = cos(ang), wi = sin(ang);
    for (int i = 0; i < n; i += len) {
      double cr = 1.0, ci = 0.0;
      for (int k = 0; k < len / 2; ++k) {
        const int a = i + k, b2 = i + k + len / 2;
        const double xr = re[b2] * cr - im[b2] * ci, xi = re[b2] * ci + im[b2] * cr;
        re[b2] = re[a] - xr; im[b2] = im[a] - xi;
        re[a] += xr; im[a] += xi;
        const double t = cr * wr - ci * wi; ci = cr * wi + ci * wr; cr = t;
      }
    }
  }
  if (inverse) { const double s = 1.0 / n; for (int i = 0; i < n; ++i) { re[i] *= s; im[i] *= s; } }
}
/* The taps' transform is computed once per parameter set and thread, not once per trace: a small per-thread cache keyed by the tap
 * array (pointer, length, transform size and a checksum of the values) — what any FFT implementation of a fixed filter does. */
#define FIR_FFT_SLOTS 4
typedef struct { const double* h; int m, N; double sum; double* H; } fir_fft_slot;
static __thread fir_fft_slot g_fft_cache[FIR_FFT_SLOTS];
static __thread int g_fft_next = 0;
static const double* fir_fft_taps(const double* h, int m, int N) {
  double sum = 0.0;
  for (int j = 0; j < m; ++j) sum += h[j] * (double)(1 + (j & 7));
  for (int s = 0; s < FIR_FFT_SLOTS; ++s)
    if (g_fft_cache[s].H && g_fft_cache[s].h == h && g_fft_cache[s].m == m && g_fft_cache[s].N == N && g_fft_cache[s].sum == sum) return g_fft_cache[s].H;
  fir_fft_slot* c = &g_fft_cache[g_fft_next];
  g_fft_next = (g_fft_next + 1) % FIR_FFT_SLOTS;
  free(c->H);
  c->H = (double*)malloc(sizeof(double) * 2 * (size_t)N);
  if (!c->H) return NULL;
  for (int i = 0; i < N; ++i) { c->H[i] = i < m ? h[i] : 0.0; c->H[N + i] = 0.0; }
  fft_c2(c->H, c->H + N, N, 0);
  c->h = h; c->m = m; c->N = N; c->sum = sum;
  return c->H;
}
static int fir_fft(const real* x, int n, const double* h, int m, real* y) {
  const int nout = n - m + 1;
  int N = 1;
  while (N < n + m - 1) N <<= 1;
  const double* H = fir_fft_taps(h, m, N);
  double* buf = (double*)malloc(sizeof(double) * 2 * (size_t)N);
  if (!buf || !H) { free(buf); return ORC_ERR_ARG; }
  double *xr = buf, *xi = buf + N;
  const double *hr = H, *hi = H + N;
  for (int i = 0; i < N; ++i) { xr[i] = i < n ? (double)x[i] : 0.0; xi[i] = 0.0; }
  fft_c2(xr, xi, N, 0);
  for (int i = 0; i < N; ++i) { const double a = xr[i] * hr[i] - xi[i] * hi[i], b2 = xr[i] * hi[i] + xi[i] * hr[i]; xr[i] = a; xi[i] = b2; }
  fft_c2(xr, xi, N, 1);
  for (int k = 0; k < nout; ++k) y[k] = (real)xr[k + m - 1];   /* full convolution index k + m - 1 = valid-mode output k */
  free(buf);
  return nout;
}

/* Valid-mode true convolution y[k] = sum_j h[j] x[k+m-1-j] (ConvolutionFilter) */
int orc_fir(const real* x, int n, const double* h, int m, real* y) {
  int nout = n - m + 1;
  if (m < 1) return ORC_ERR_ARG;
  if (nout < 1) return ORC_ERR_WINDOW;
  if (g_fir_mode == 1 && m > 64) return fir_fft(x, n, h, m, y);
  real* hr = (real*)malloc(sizeof(real) * (size_t)m);
  for (int j = 0; j < m; ++j) hr[j] = h[m - 1 - j];
  /* tap-outer / output-inner: each y[k] still accumulates its taps in ascending
   * order, but the inner loop vectorises across outputs */
  for (int k = 0; k < nout; ++k) y[k] = 0;
  for (int j = 0; j < m; ++j) {
    const real hj = hr[j];
    const real* restrict xp = x + j;
    real* restrict yp = y;
    for (int k = 0; k < nout; ++k) yp[k] += hj * xp[k];
  }
  free(hr);
  return nout;
}
/* CUSP kernel — RadiationDetectorDSP CUSPChargeFilter; restated from the
 * published pygama `cusp_filter` algorithm it follows (assumption
 * A4): sinh flanks, flat top of toplen+1, convolved with [1, -exp(-1/tau)]
 * ("same" length), scaled by beta/length so that a unit step gives unit
 * amplitude when beta = length (what src/dsp_icpc.jl:90,167 passes). */
static void cusp_shape(const ldsp_cuspzac* p, double* cusp) {
  int L = p->length, flat = p->flat, lt = (L - flat) / 2;
  double den = sinh(lt / p->sigma);
  for (int i = 0; i < L; ++i) cusp[i] = 0;
  for (int i = 0; i < lt && i < L; ++i) cusp[i] = sinh(i / p->sigma) / den;
  for (int i = lt; i <= lt + flat && i < L; ++i) cusp[i] = 1.0;
  for (int i = lt + flat + 1; i < L; ++i) cusp[i] = sinh((L - i) / p->sigma) / den;
}
static void deconv_scale(const ldsp_cuspzac* p, const double* shape, double* h) {
  double a = exp(-1.0 / p->tau), sc = p->beta / (double)p->length;
  for (int i = 0; i < p->length; ++i) h[i] = sc * (shape[i] - (i > 0 ? a * shape[i - 1] : 0.0));
}
int orc_cusp_coeffs(const ldsp_cuspzac* p, double* h) {
  if (p->length < 3 || p->flat < 0 || p->flat >= p->length || !(p->sigma > 0) || !(p->tau > 0)) return ORC_ERR_ARG;
  double* c = (double*)malloc(sizeof(double) * (size_t)p->length);
  cusp_shape(p, c);
  deconv_scale(p, c, h);
  free(c);
  return ORC_OK;
}
/* ZAC kernel — cusp minus area-matched parabolas on both flanks (pygama
 * `zac_filter`, SURVEY a23, assumption A4). */
int orc_zac_coeffs(const ldsp_cuspzac* p, double* h) {
  if (p->length < 3 || p->flat < 0 || p->flat >= p->length || !(p->sigma > 0) || !(p->tau > 0)) return ORC_ERR_ARG;
  int L = p->length, flat = p->flat, lt = (L - flat) / 2;
  double* c = (double*)malloc(sizeof(double) * (size_t)L);
  double* par = (double*)malloc(sizeof(double) * (size_t)L);
  cusp_shape(p, c);
  double half = 0.5 * lt;
  for (int i = 0; i < L; ++i) par[i] = 0;
  for (int i = 0; i < lt && i < L; ++i) par[i] = (i - half) * (i - half) - half * half;
  for (int i = lt + flat + 1; i < L; ++i) par[i] = (L - i - half) * (L - i - half) - half * half;
  double apar = 0, acusp = 0;
  for (int i = 0; i < L; ++i) { apar += par[i]; acusp += c[i]; }
  for (int i = 0; i < L; ++i) c[i] = c[i] - par[i] / apar * acusp;
  deconv_scale(p, c, h);
  free(c); free(par);
  return ORC_OK;
}
/* DerivativeFilter — src/derivative.jl:47-55 */
int orc_derivative(const real* x, int n, double gain, real* y) {
  for (int i = 0; i < n; ++i) {
    int a = i > 1 ? i : 1, b = i - 1 > 0 ? i - 1 : 0;
    if (a > n - 1) a = n - 1; /* n == 1: the reference would index out of bounds */
    y[i] = gain * (x[a] - x[b]);
  }
  return n;
}
/* HaarAveragingFilter — src/haar_filter.jl:26-39 */
int orc_haar(const real* x, int n, int ds, real* y) {
  if (ds < 1) return ORC_ERR_ARG;
  int nout = (n + ds - 1) / ds;
  real inv = 1.0 / sqrt(2.0);
  for (int i = 0; i < nout; ++i) {
    int s = i * ds, e = s + 1 < n ? s + 1 : n - 1;
    y[i] = (x[s] + x[e]) * inv;
  }
  return nout;
}
/* MovingWindowFilter — src/moving_window_multi.jl:99-116 */
int orc_moving_window(const real* x, int n, int l, real* y) {
  if (l < 1 || n < 1) return ORC_ERR_ARG;
  real x1 = x[0], invl = (real)(1.0 / l);
  y[0] = x1;
  for (int i = 1; i < l && i < n; ++i) y[i] = fma(invl, x[i] - x1, y[i - 1]);
  for (int i = l; i < n; ++i) y[i] = fma(invl, x[i] - x[i - l], y[i - 1]);
  return n;
}
/* MovingWindowMultiFilter — src/moving_window_multi.jl:118-129:
 * fwd(x) -> _y; fwd on reversed _y, written reversed into y; fwd(y) -> y */
int orc_moving_window_multi(const real* x, int n, int l, real* y) {
  if (l < 1 || n < 1) return ORC_ERR_ARG;
  real* a = (real*)malloc(sizeof(real) * (size_t)n);
  real* b = (real*)malloc(sizeof(real) * (size_t)n);
  orc_moving_window(x, n, l, a);
  for (int i = 0; i < n; ++i) b[i] = a[n - 1 - i];
  orc_moving_window(b, n, l, a);
  for (int i = 0; i < n; ++i) b[i] = a[n - 1 - i];
  orc_moving_window(b, n, l, y);
  free(a); free(b);
  return n;
}

/* ------------------------------------------------------------------------ */
/* L3 helpers — src/dsp_routines.jl */

/* get_t0 — src/dsp_routines.jl:9-25; returns us, NaN -> 0 */
static int get_t0(const real* y, int n, double t_first, double dt, double upus, ldsp_trap tr,
                  double thr, int mintot, real* scratch, double* t0) {
  int no = orc_trap(y, n, tr.navg, tr.ngap, tr.navg2, scratch);
  if (no < 0) return no;
  double tf = t_first + (tr.navg + tr.ngap + tr.navg2 - 1) * dt; /* A1: trailing alignment */
  double x; int mult;
  orc_intersect(scratch, no, tf, dt, thr, mintot, &x, &mult);
  x /= upus;
  *t0 = (x != x) ? 0.0 : x;
  return ORC_OK;
}
/* get_threshold — src/dsp_routines.jl:33-42 */
static double get_threshold(const real* y, int n, double t_first, double dt, double upus, double thr, int mintot) {
  double x; int mult;
  orc_intersect(y, n, t_first, dt, thr, mintot, &x, &mult);
  x /= upus;
  return (x != x) ? 0.0 : x;
}
/* get_qdrift — src/dsp_routines.jl:51-64 (integ = IntegratorFilter(1)(wvfs), precomputed) */
static double get_qdrift(const real* integ, int n, double t_first, double dt, double t_start,
                         double d1, double d2, ldsp_dni est) {
  double e0, e1, e2;
  orc_signal_estimator(integ, n, t_first, dt, t_start, est.npts, est.degree, &e0);
  orc_signal_estimator(integ, n, t_first, dt, t_start + d1, est.npts, est.degree, &e1);
  orc_signal_estimator(integ, n, t_first, dt, t_start + d2, est.npts, est.degree, &e2);
  double area1 = e1 - e0, area2 = e2 - e1;
  return area2 - area1;
}

static inline int win_idx(double t, double t_first, double dt) { return (int)rnd((t - t_first) / dt); }

/* ------------------------------------------------------------------------ */
/* dsp_icpc — src/dsp_icpc.jl:62-230, one trace.  out: LDSP_ICPC_NCOLS doubles
 * in the column order of ldsp_icpc_out (ints stored as doubles). */
enum {
  C_blmean, C_blsigma, C_blslope, C_bloffset, C_tailmean, C_tailsigma, C_tailslope, C_tailoffset,
  C_t0, C_t10, C_t50, C_t80, C_t90, C_t99, C_t50_current, C_drift_time,
  C_tail_tau, C_tail_mean, C_tail_sigma, C_e_max, C_e_min,
  C_e_10410, C_e_535, C_e_313, C_e_10410_inv, C_e_313_inv, C_t0_inv,
  C_e_trap, C_e_cusp, C_e_zac, C_e_trap_max, C_e_cusp_max, C_e_zac_max,
  C_t_trap_max, C_t_cusp_max, C_t_zac_max, C_qdrift, C_lq,
  C_a_sg, C_a_60, C_a_100, C_a_raw, C_inTrace_intersect, C_inTrace_n,
  C_n_sat_low, C_n_sat_high, C_n_sat_low_cons, C_n_sat_high_cons, C_NCOLS
};

typedef struct {
  real *x, *y, *integ, *flt, *neg, *sg; double *hc, *hz, *hsg[3];
} icpc_ws;

static int icpc_one(const float* wf, const ldsp_icpc_params* p, icpc_ws* w, double* o) {
  const int L = p->L;
  const double t0f = p->t_first, dt = p->dt, up = p->unit_per_us;
  real* x = w->x; real* y = w->y;
  int rc;
  for (int i = 0; i < C_NCOLS; ++i) o[i] = NAN;
  for (int i = 0; i < L; ++i) x[i] = (real)wf[i];

  int sat[4]; /* :93-95 */
  if ((rc = orc_saturation(x, L, 0, L - 1, p->sat_low, p->sat_high, sat))) return rc;
  o[C_n_sat_low] = sat[0]; o[C_n_sat_high] = sat[1]; o[C_n_sat_low_cons] = sat[2]; o[C_n_sat_high_cons] = sat[3];

  /* :102 */
  if ((rc = orc_signalstats(x, L, p->bl_from, p->bl_until, t0f, dt, &o[C_blmean], &o[C_blsigma], &o[C_blslope], &o[C_bloffset]))) return rc;
  for (int i = 0; i < L; ++i) x[i] -= o[C_blmean]; /* :105 */
  real wmax = x[0], wmin = x[0];                  /* :111-112 */
  for (int i = 1; i < L; ++i) { if (x[i] > wmax) wmax = x[i]; if (x[i] < wmin) wmin = x[i]; }
  o[C_e_max] = wmax; o[C_e_min] = wmin;
  /* :115 */
  if ((rc = orc_tailstats(x, L, p->tail_from, p->tail_until, t0f, dt, &o[C_tail_mean], &o[C_tail_sigma], &o[C_tail_tau]))) return rc;
  orc_invcr(x, L, p->pz_c, y); /* :119-120 */
  /* :123 */
  if ((rc = orc_signalstats(y, L, p->tail_from, p->tail_until, t0f, dt, &o[C_tailmean], &o[C_tailsigma], &o[C_tailslope], &o[C_tailoffset]))) return rc;
  /* :126 */
  if ((rc = get_t0(y, L, t0f, dt, up, p->t0_trap, p->t0_threshold, p->t0_mintot, w->flt, &o[C_t0]))) return rc;
  /* :132-136 */
  o[C_t10] = get_threshold(y, L, t0f, dt, up, wmax * 0.1, p->tx_mintot);
  o[C_t50] = get_threshold(y, L, t0f, dt, up, wmax * 0.5, p->tx_mintot);
  o[C_t80] = get_threshold(y, L, t0f, dt, up, wmax * 0.8, p->tx_mintot);
  o[C_t90] = get_threshold(y, L, t0f, dt, up, wmax * 0.9, p->tx_mintot);
  o[C_t99] = get_threshold(y, L, t0f, dt, up, wmax * 0.99, p->tx_mintot);
  o[C_drift_time] = (o[C_t90] - o[C_t0]) * up; /* :138 */
  /* :141,144 */
  orc_integrator(y, L, 1.0, w->integ);
  o[C_qdrift] = get_qdrift(w->integ, L, t0f, dt, o[C_t0] * up, p->qdrift_d1, p->qdrift_d2, p->int_est);
  o[C_lq] = get_qdrift(w->integ, L, t0f, dt, o[C_t80] * up, p->lq_d1, p->lq_d2, p->int_est);
  /* :147-154 */
  const int cols_fixed[3] = {C_e_10410, C_e_535, C_e_313};
  for (int f = 0; f < 3; ++f) {
    int no = orc_trap(y, L, p->trap_fixed[f].navg, p->trap_fixed[f].ngap, p->trap_fixed[f].navg2, w->flt);
    if (no < 0) return no;
    real m = w->flt[0];
    for (int i = 1; i < no; ++i) if (w->flt[i] > m) m = w->flt[i];
    o[cols_fixed[f]] = m;
  }
  /* :160-164 */
  {
    ldsp_trap tr = p->trap_opt;
    int no = orc_trap(y, L, tr.navg, tr.ngap, tr.navg2, w->flt);
    if (no < 0) return no;
    double tf = t0f + (tr.navg + tr.ngap + tr.navg2 - 1) * dt, vmin, tmin;
    orc_signal_estimator(w->flt, no, tf, dt, o[C_t50] * up + p->trap_pickoff, p->sig_est.npts, p->sig_est.degree, &o[C_e_trap]);
    orc_extremestats(w->flt, no, 0, no - 1, tf, dt, &vmin, &o[C_e_trap_max], &tmin, &o[C_t_trap_max]);
  }
  /* :167-171 */
  {
    int no = orc_fir(y, L, w->hc, p->cusp.length, w->flt);
    if (no < 0) return no;
    double tf = t0f + (p->cusp.length - 1) * dt, vmin, tmin;
    orc_signal_estimator(w->flt, no, tf, dt, o[C_t50] * up + p->cusp_pickoff, p->sig_est.npts, p->sig_est.degree, &o[C_e_cusp]);
    orc_extremestats(w->flt, no, 0, no - 1, tf, dt, &vmin, &o[C_e_cusp_max], &tmin, &o[C_t_cusp_max]);
  }
  /* :174-178 (the reference applies the ZAC filter twice; same result) */
  {
    int no = orc_fir(y, L, w->hz, p->zac.length, w->flt);
    if (no < 0) return no;
    double tf = t0f + (p->zac.length - 1) * dt, vmin, tmin;
    orc_signal_estimator(w->flt, no, tf, dt, o[C_t50] * up + p->zac_pickoff, p->sig_est.npts, p->sig_est.degree, &o[C_e_zac]);
    orc_extremestats(w->flt, no, 0, no - 1, tf, dt, &vmin, &o[C_e_zac_max], &tmin, &o[C_t_zac_max]);
  }
  /* :181-186 */
  const int cols_a[3] = {C_a_sg, C_a_60, C_a_100};
  for (int f = 2; f >= 0; --f) { /* f = 0 last so that w->sg holds the sg_wl filter output */
    int no = orc_fir(y, L, w->hsg[f], p->sg_npts[f], w->sg);
    if (no < 0) return no;
    double tf = t0f + (p->sg_npts[f] - 1) * dt;
    int from = win_idx(p->cur_left, tf, dt), until = win_idx(p->cur_right, tf, dt);
    if ((rc = orc_get_wvf_maximum(w->sg, no, from, until, &o[cols_a[f]]))) return rc;
  }
  {
    orc_derivative(y, L, 1.0, w->flt);
    int from = win_idx(p->cur_left, t0f, dt), until = win_idx(p->cur_right, t0f, dt);
    if ((rc = orc_get_wvf_maximum(w->flt, L, from, until, &o[C_a_raw]))) return rc;
  }
  /* :189 get_intracePileUp — src/dsp_routines.jl:72-82 */
  {
    int no = L - p->sg_npts[0] + 1;
    double tf = t0f + (p->sg_npts[0] - 1) * dt;
    int from = win_idx(p->bl_left + tf, tf, dt), until = win_idx(p->bl_right, tf, dt);
    double m, sg, sl, of;
    if ((rc = orc_signalstats(w->sg, no, from, until, tf, dt, &m, &sg, &sl, &of))) return rc;
    double thr = sg * p->intrace_nsigma;
    if (thr == 0) thr = 1; /* :77 */
    for (int i = 0; i < no; ++i) w->flt[i] = w->sg[no - 1 - i]; /* reverse_waveform :79 */
    double xr; int mult;
    orc_intersect(w->flt, no, tf, dt, thr, p->intrace_mintot, &xr, &mult);
    o[C_inTrace_intersect] = (tf + (no - 1) * dt) - xr; /* :81, NaN stays NaN */
    o[C_inTrace_n] = mult;
    /* :192-195 */
    real gmax = w->sg[0];
    for (int i = 1; i < no; ++i) if (w->sg[i] > gmax) gmax = w->sg[i];
    o[C_t50_current] = get_threshold(w->sg, no, tf, dt, up, gmax * 0.5, p->tx_mintot);
  }
  /* :199-207 */
  for (int i = 0; i < L; ++i) w->neg[i] = y[i] * -1.0;
  {
    int no = orc_trap(w->neg, L, p->trap_fixed[0].navg, p->trap_fixed[0].ngap, p->trap_fixed[0].navg2, w->flt);
    real m = w->flt[0];
    for (int i = 1; i < no; ++i) if (w->flt[i] > m) m = w->flt[i];
    o[C_e_10410_inv] = m;
    no = orc_trap(w->neg, L, p->trap_fixed[2].navg, p->trap_fixed[2].ngap, p->trap_fixed[2].navg2, w->flt);
    m = w->flt[0];
    for (int i = 1; i < no; ++i) if (w->flt[i] > m) m = w->flt[i];
    o[C_e_313_inv] = m;
  }
  if ((rc = get_t0(w->neg, L, t0f, dt, up, p->t0inv_trap, p->t0_threshold, p->t0_mintot, w->flt, &o[C_t0_inv]))) return rc;
  return ORC_OK;
}

static int icpc_ws_alloc(const ldsp_icpc_params* p, icpc_ws* w) {
  size_t L = (size_t)p->L;
  w->x = (real*)malloc(sizeof(real) * L); w->y = (real*)malloc(sizeof(real) * L); w->integ = (real*)malloc(sizeof(real) * L);
  w->flt = (real*)malloc(sizeof(real) * L); w->neg = (real*)malloc(sizeof(real) * L); w->sg = (real*)malloc(sizeof(real) * L);
  w->hc = (double*)malloc(8 * (size_t)p->cusp.length); w->hz = (double*)malloc(8 * (size_t)p->zac.length);
  int rc = orc_cusp_coeffs(&p->cusp, w->hc);
  if (!rc) rc = orc_zac_coeffs(&p->zac, w->hz);
  for (int f = 0; f < 3; ++f) {
    w->hsg[f] = (double*)malloc(8 * (size_t)(p->sg_npts[f] > 0 ? p->sg_npts[f] : 1));
    if (!rc) rc = orc_sg_coeffs(p->sg_npts[f], p->sg_degree, 1, w->hsg[f]);
  }
  return rc;
}
static void icpc_ws_free(icpc_ws* w) {
  free(w->x); free(w->y); free(w->integ); free(w->flt); free(w->neg); free(w->sg); free(w->hc); free(w->hz);
  for (int f = 0; f < 3; ++f) free(w->hsg[f]);
}

int orc_icpc_ncols(void) { return C_NCOLS; }

/* Batch driver: wf [n][L] float32 (host), out [n][C_NCOLS] float64 row-major.
 * status[n] (may be NULL) receives the per-trace error code.  nthreads <= 1:
 * single-threaded (the reference's execution model). */
int orc_dsp_icpc(const float* wf, long n, const ldsp_icpc_params* p, double* out, int* status, int nthreads) {
  int rc_all = ORC_OK;
  if (p->L < 8 || p->L > LDSP_MAX_L) return ORC_ERR_ARG;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
  {
    icpc_ws w;
    int rc0 = icpc_ws_alloc(p, &w);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long i = 0; i < n; ++i) {
      int rc = rc0 ? rc0 : icpc_one(wf + (size_t)i * (size_t)p->L, p, &w, out + (size_t)i * C_NCOLS);
      if (status) status[i] = rc;
      if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
        rc_all = rc;
      }
    }
    icpc_ws_free(&w);
  }
  return rc_all;
}

/* BASELINE config 2 sub-chain: blmean -> shift -> InvCR -> Trap(10us,4us) -> maximum
 * (src/dsp_icpc.jl:102-105,119-120,147-148). out [n][2] = (blmean, e_10410). */
int orc_icpc_pz_trap(const float* wf, long n, const ldsp_icpc_params* p, double* out) {
  int L = p->L;
  real* x = (real*)malloc(sizeof(real) * (size_t)L);
  real* y = (real*)malloc(sizeof(real) * (size_t)L);
  real* f = (real*)malloc(sizeof(real) * (size_t)L);
  int rc = ORC_OK;
  for (long i = 0; i < n && !rc; ++i) {
    for (int k = 0; k < L; ++k) x[k] = (real)wf[(size_t)i * L + k];
    double m, s, sl, of;
    rc = orc_signalstats(x, L, p->bl_from, p->bl_until, p->t_first, p->dt, &m, &s, &sl, &of);
    if (rc) break;
    for (int k = 0; k < L; ++k) x[k] -= m;
    orc_invcr(x, L, p->pz_c, y);
    int no = orc_trap(y, L, p->trap_fixed[0].navg, p->trap_fixed[0].ngap, p->trap_fixed[0].navg2, f);
    if (no < 0) { rc = no; break; }
    real mx = f[0];
    for (int k = 1; k < no; ++k) if (f[k] > mx) mx = f[k];
    out[2 * i] = m; out[2 * i + 1] = mx;
  }
  free(x); free(y); free(f);
  return rc;
}

/* ------------------------------------------------------------------------ */
/* dsp_sipm — src/dsp_sipm.jl:47-158, one trace.
 * scalars: 20 doubles in the order of ldsp_sipm_out's scalar columns.
 * trig: 4 groups (SG, DC, trap, DC_trap) x 4 fields (x, x_high, x_tot, max)
 *       x cap doubles; counts[4]. */
enum {
  S_t_max, S_t_min, S_t_max_lar, S_t_min_lar, S_e_max, S_e_min, S_e_max_lar, S_e_min_lar,
  S_blmean, S_blsigma, S_blslope, S_bloffset, S_wfmean, S_wfsigma, S_wfslope, S_wfoffset,
  S_threshold, S_threshold_DC, S_threshold_trap, S_threshold_DC_trap, S_NCOLS
};
int orc_sipm_ncols(void) { return S_NCOLS; }

static int sipm_one(const float* wf, const ldsp_sipm_params* p, real* ws, const double* hsg,
                    double* o, double* trig, int* counts, int cap) {
  const int L = p->L;
  const double t0f = p->t_first, dt = p->dt, up = p->unit_per_us;
  real *x = ws, *g = ws + L, *I = ws + 2 * L, *F = ws + 3 * L, *P = ws + 4 * L, *T = ws + 5 * L;
  int rc;
  for (int i = 0; i < L; ++i) x[i] = (real)wf[i] + (real)0; /* :88 */
  double vmin, vmax, tmin, tmax;
  if ((rc = orc_extremestats(x, L, 0, L - 1, t0f, dt, &vmin, &vmax, &tmin, &tmax))) return rc; /* :91 */
  o[S_e_min] = vmin; o[S_e_max] = vmax; o[S_t_min] = tmin / up; o[S_t_max] = tmax / up;
  /* TruncateFilter + extremestats :94-95 (same absolute times) */
  if ((rc = orc_extremestats(x, L, p->trunc_from, p->trunc_until, t0f, dt, &vmin, &vmax, &tmin, &tmax))) return rc;
  o[S_e_min_lar] = vmin; o[S_e_max_lar] = vmax; o[S_t_min_lar] = tmin / up; o[S_t_max_lar] = tmax / up;
  /* SG derivative :99-100 */
  int ng = orc_fir(x, L, hsg, p->sg_npts, g);
  if (ng < 0) return ng;
  double tg = t0f + (p->sg_npts - 1) * dt;
  /* :103-105 */
  double thr = orc_thresholdstats_mad(g, ng, p->sg_min_thr, p->sg_max_thr);
  o[S_threshold] = thr;
  double* tr0 = trig;
  counts[0] = orc_intersect_maximum(g, ng, tg, dt, p->sg_nsigma * thr, p->sg_mintot, p->sg_maxtot, cap,
                                    tr0, tr0 + cap, tr0 + 2 * cap, tr0 + 3 * cap);
  /* :108-109 */
  orc_integrator(g, ng, 1.0, I);
  /* :112-115 — minimum.(inters.x; init=0) folds the init into the min */
  double time_min = tg, d3 = 3 * dt, minx = 0.0;
  for (int k = 0; k < counts[0] && k < cap; ++k) if (tr0[k] < minx) minx = tr0[k];
  double stop = (minx < time_min + d3) ? time_min + d3 : minx;
  if ((rc = orc_signalstats(I, ng, win_idx(time_min, tg, dt), win_idx(stop, tg, dt), tg, dt,
                            &o[S_blmean], &o[S_blsigma], &o[S_blslope], &o[S_bloffset]))) return rc;
  if ((rc = orc_signalstats(I, ng, 0, ng - 1, tg, dt, &o[S_wfmean], &o[S_wfsigma], &o[S_wfslope], &o[S_wfoffset]))) return rc;
  /* :118-120 */
  for (int i = 0; i < ng; ++i) F[i] = I[i] * -1.0;
  double thr_dc = orc_thresholdstats_mad(F, ng, p->sg_min_dc_thr, p->sg_max_dc_thr);
  o[S_threshold_DC] = thr_dc;
  double* tr1 = trig + 4 * cap;
  counts[1] = orc_intersect_maximum(F, ng, tg, dt, p->sg_nsigma_dc * thr_dc, p->sg_mintot, p->sg_maxtot, cap,
                                    tr1, tr1 + cap, tr1 + 2 * cap, tr1 + 3 * cap);
  /* :124-129 */
  orc_invcr(I, ng, p->pz_c, P);
  int nt = orc_trap(P, ng, p->trap.navg, p->trap.ngap, p->trap.navg2, T);
  if (nt < 0) return nt;
  double tt = tg + (p->trap.navg + p->trap.ngap + p->trap.navg2 - 1) * dt;
  /* :132-134 */
  double thr_t = orc_thresholdstats_mad(T, nt, p->trap_min_thr, p->trap_max_thr);
  o[S_threshold_trap] = thr_t;
  double* tr2 = trig + 8 * cap;
  counts[2] = orc_intersect_maximum(T, nt, tt, dt, p->trap_nsigma * thr_t, p->trap_mintot, p->trap_maxtot, cap,
                                    tr2, tr2 + cap, tr2 + 2 * cap, tr2 + 3 * cap);
  /* :137-138 — uses intflt_sg (the SG functor), not intflt_trap */
  double thr_dct = orc_thresholdstats_mad(F, ng, p->trap_min_dc_thr, p->trap_max_dc_thr);
  o[S_threshold_DC_trap] = thr_dct;
  double* tr3 = trig + 12 * cap;
  counts[3] = orc_intersect_maximum(F, ng, tg, dt, p->trap_nsigma_dc * thr_dct, p->sg_mintot, p->sg_maxtot, cap,
                                    tr3, tr3 + cap, tr3 + 2 * cap, tr3 + 3 * cap);
  return ORC_OK;
}

/* out_scalars [n][S_NCOLS]; out_trig [n][16][cap]; counts [n][4] */
int orc_dsp_sipm(const float* wf, long n, const ldsp_sipm_params* p, double* out_scalars,
                 double* out_trig, int* counts, int cap, int* status, int nthreads) {
  if (p->L < 8 || p->L > LDSP_MAX_L) return ORC_ERR_ARG;
  int rc_all = ORC_OK;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
  {
    real* ws = (real*)malloc(sizeof(real) * (size_t)p->L * 6);
    double* hsg = (double*)malloc(8 * (size_t)(p->sg_npts > 0 ? p->sg_npts : 1));
    int rc0 = orc_sg_coeffs(p->sg_npts, p->sg_degree, 1, hsg);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long i = 0; i < n; ++i) {
      for (int k = 0; k < 16 * cap; ++k) out_trig[(size_t)i * 16 * cap + k] = NAN;
      int rc = rc0 ? rc0
                   : sipm_one(wf + (size_t)i * (size_t)p->L, p, ws, hsg, out_scalars + (size_t)i * S_NCOLS,
                              out_trig + (size_t)i * 16 * cap, counts + 4 * i, cap);
      if (status) status[i] = rc;
      if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
        rc_all = rc;
      }
    }
    free(ws); free(hsg);
  }
  return rc_all;
}
