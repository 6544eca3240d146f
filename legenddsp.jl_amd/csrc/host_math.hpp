// host_math.hpp — host-side (double / long double) derivation of the constants the
// kernels consume: least-squares polynomial bases, Savitzky-Golay taps, CUSP/ZAC
// shapes and their closed-form decomposition.  Product code: independent of oracle/.
#pragma once
#include <cmath>
#include <vector>
#include "../../include/ldsp.h"

namespace ldsp {
namespace hm {

// Weights of the least-squares polynomial of `degree` through npts equidistant
// points, in the centred/scaled coordinate u = (i - c)/s, c = (npts-1)/2,
// s = max(c, 1):  yhat(u) = sum_i y_i sum_j B[i*(degree+1)+j] u^j.
// (The projector V (V'V)^-1 of RadiationDetectorDSP `_lsq_fit_matrix`, whose
// convention is visible at reference src/multi_intersect.jl:80-84,115-123.)
inline bool lsq_basis(int npts, int degree, std::vector<double>& B, double& c, double& s) {
  if (npts < 1 || degree < 0 || degree >= npts || degree > 12) return false;
  const int d1 = degree + 1;
  c = 0.5 * (npts - 1);
  s = c > 1.0 ? c : 1.0;
  std::vector<long double> M((size_t)d1 * 2 * d1, 0.0L);
  auto at = [&](int r, int q) -> long double& { return M[(size_t)r * 2 * d1 + q]; };
  for (int i = 0; i < npts; ++i) {
    long double u = ((long double)i - c) / s;
    std::vector<long double> pw(2 * d1, 1.0L);
    for (int a = 1; a < 2 * d1; ++a) pw[a] = pw[a - 1] * u;
    for (int a = 0; a < d1; ++a)
      for (int b = 0; b < d1; ++b) at(a, b) += pw[a + b];
  }
  for (int a = 0; a < d1; ++a) at(a, d1 + a) = 1.0L;
  for (int col = 0; col < d1; ++col) {
    int piv = col;
    for (int r = col + 1; r < d1; ++r)
      if (fabsl(at(r, col)) > fabsl(at(piv, col))) piv = r;
    if (fabsl(at(piv, col)) < 1e-30L) return false;
    if (piv != col)
      for (int q = 0; q < 2 * d1; ++q) std::swap(at(col, q), at(piv, q));
    long double inv = 1.0L / at(col, col);
    for (int q = 0; q < 2 * d1; ++q) at(col, q) *= inv;
    for (int r = 0; r < d1; ++r) {
      if (r == col) continue;
      long double f = at(r, col);
      if (f == 0.0L) continue;
      for (int q = 0; q < 2 * d1; ++q) at(r, q) -= f * at(col, q);
    }
  }
  B.assign((size_t)npts * d1, 0.0);
  for (int i = 0; i < npts; ++i) {
    long double u = ((long double)i - c) / s;
    for (int j = 0; j < d1; ++j) {
      long double acc = 0, pu = 1;
      for (int a = 0; a < d1; ++a) { acc += pu * at(a, d1 + j); pu *= u; }
      B[(size_t)i * d1 + j] = (double)acc;
    }
  }
  return true;
}

// Savitzky-Golay taps in correlation form: out[k] = sum_i c[i] x[k+i]; the
// `deriv`-th derivative (per sample) of the LSQ polynomial at the window centre.
inline bool sg_corr_coeffs(int npts, int degree, int deriv, std::vector<double>& cc) {
  if (npts < 1 || (npts & 1) == 0 || deriv < 0 || deriv > degree) return false;
  std::vector<double> B;
  double c, s;
  if (!lsq_basis(npts, degree, B, c, s)) return false;
  double fact = 1;
  for (int k = 2; k <= deriv; ++k) fact *= k;
  const double scale = fact / std::pow(s, deriv);
  cc.resize(npts);
  for (int i = 0; i < npts; ++i) cc[i] = B[(size_t)i * (degree + 1) + deriv] * scale;
  return true;
}

struct CuspShape {
  int Lf, flat, lt, f1, ltp;
  double sigma, den;
};
inline bool cusp_shape_ok(const ldsp_cuspzac& p) {
  if (p.length < 5 || p.flat < 0 || !(p.sigma > 0) || !(p.tau > 0)) return false;
  int lt = (p.length - p.flat) / 2;
  int f1 = lt + p.flat + 1;
  return lt >= 2 && f1 <= p.length - 1;
}
inline CuspShape cusp_geometry(const ldsp_cuspzac& p) {
  CuspShape g;
  g.Lf = p.length; g.flat = p.flat; g.lt = (p.length - p.flat) / 2;
  g.f1 = g.lt + g.flat + 1; g.ltp = g.Lf - g.f1;
  g.sigma = p.sigma; g.den = std::sinh(g.lt / p.sigma);
  return g;
}
// cusp[j]: sinh flanks + flat top of flat+1 samples; par[j]: the ZAC parabolas
// (pygama cusp_filter / zac_filter shapes that RadiationDetectorDSP follows —
// DESIGN.md assumption A4)
inline void cusp_and_par(const ldsp_cuspzac& p, std::vector<double>& cusp, std::vector<double>& par) {
  CuspShape g = cusp_geometry(p);
  cusp.assign(g.Lf, 0.0);
  par.assign(g.Lf, 0.0);
  const double half = 0.5 * g.lt;
  for (int j = 0; j < g.Lf; ++j) {
    if (j < g.lt) {
      cusp[j] = std::sinh(j / g.sigma) / g.den;
      par[j] = (j - half) * (j - half) - half * half;
    } else if (j < g.f1) {
      cusp[j] = 1.0;
    } else {
      cusp[j] = std::sinh((g.Lf - j) / g.sigma) / g.den;
      par[j] = (g.Lf - j - half) * (g.Lf - j - half) - half * half;
    }
  }
}
// true-convolution taps: shape deconvolved with [1, -exp(-1/tau)] ("same"), scaled by beta/Lf
inline void cuspzac_taps(const ldsp_cuspzac& p, bool zac, std::vector<double>& h) {
  std::vector<double> cusp, par;
  cusp_and_par(p, cusp, par);
  std::vector<double> shape = cusp;
  if (zac) {
    double apar = 0, acusp = 0;
    for (int j = 0; j < p.length; ++j) { apar += par[j]; acusp += cusp[j]; }
    for (int j = 0; j < p.length; ++j) shape[j] = cusp[j] - par[j] / apar * acusp;
  }
  const double a = std::exp(-1.0 / p.tau), sc = p.beta / (double)p.length;
  h.resize(p.length);
  for (int j = 0; j < p.length; ++j) h[j] = sc * (shape[j] - (j > 0 ? a * shape[j - 1] : 0.0));
}

}  // namespace hm
}  // namespace ldsp
