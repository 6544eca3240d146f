// qdrift.hpp — get_qdrift (reference src/dsp_routines.jl:51-64) by ONE wave, shared by the dsp_icpc kernels.
//
// The reference integrates the trace (IntegratorFilter, I[i] = sum_{j<=i} y[j]), estimates I at t, t + d1, t + d2 with
// SignalEstimator(PolynomialDNI) and returns (E3 - E2) - (E2 - E1).  I reaches 1e8 on a 8192-sample trace: read back from a
// float32 prefix sum its three estimates carry +-8 each and the second difference +-40.  The weights of an LSQ estimate sum to
// one, so a common level cancels exactly in E1 - 2 E2 + E3: every window point is taken RELATIVE to the first point of the
// first window, D(i) = I[i] - I[ref] = sum_{ref < j <= i} y[j] — a float sum over at most a few hundred samples.  Each lane sums
// a chunk of consecutive samples, a wave scan gives the chunk offsets, lane l < npts evaluates point l of the three windows.  The result is valid in every lane.
#pragma once
#include "icpc_dev.hpp"
#include "wave_prims.hpp"

namespace ldsp {

// window start i0 and local coordinate u of the LSQ estimate at position (ip + fp) in a signal of nsig samples (assumption A3)
__device__ __forceinline__ void dni_window(const EstDev& E, int ip, float fp, int nsig, int* i0, float* u) {
  if (ip < 0) { ip = 0; fp = 0.f; }
  if (ip >= nsig - 1) { ip = nsig - 1; fp = 0.f; }
  int a = ip + (int)ceilf(fp - 0.5f * (float)E.npts);
  a = max(0, min(a, nsig - E.npts));
  *i0 = a;
  *u = ((float)(ip - a) + fp - E.c) * E.s_inv;
}
// Weight of window point l at local coordinate u: the polynomial sum_j B[l][j] u^j.  A row of the basis table holds
// LDSP_MAX_EST_DEG + 1 coefficients, zero above the estimator's degree (and for points beyond its window): all of them are read
// together (one wait; a Horner loop over the degree with a read per step was a chain of dependent LDS round trips) and the
// Horner chain starts from zero — fma(0, u, b[deg]) = b[deg], the same values as a chain that starts at the degree.
struct DniRow { float b[LDSP_MAX_EST_DEG + 1]; };
__device__ __forceinline__ DniRow dni_row(const float* Bt, int l) {
  DniRow r;
#pragma unroll
  for (int j = 0; j <= LDSP_MAX_EST_DEG; ++j) r.b[j] = Bt[l * (LDSP_MAX_EST_DEG + 1) + j];
  return r;
}
__device__ __forceinline__ float dni_poly(const DniRow& r, float u) {
  float w = 0.f;
#pragma unroll
  for (int j = LDSP_MAX_EST_DEG; j >= 0; --j) w = fmaf(w, u, r.b[j]);
  return w;
}
__device__ __forceinline__ float dni_weight(const EstDev& E, const float* Bt, int l, float u) { return dni_poly(dni_row(Bt, l), u); }

// Y: the trace in LDS (nsig samples); Bt: the estimator's basis table in LDS; (ip[k], fp[k]): the three positions t, t + d1, t + d2
// in samples.  Must be called by all 64 lanes of a wave.  NaN if the trace is shorter than the estimator window.
// scratch: 512 floats of LDS this wave may use, or nullptr.  With it (and a span of <= 512 samples) every sample is read once:
// a lane loads its <= 8 consecutive samples with independent reads, the prefix sums D(i) go to scratch and the window points
// fetch theirs with one read — instead of a chain of ~20 dependent LDS round trips (the waves of the workgroup that have
// nothing to do in this phase wait for this one).
__device__ __forceinline__ float qdrift_wave(const EstDev& E, const float* Bt, const float* Y, int nsig, const int (&ip)[3], const float (&fp)[3],
                                             float* scratch = nullptr) {
  if (nsig < E.npts) return NAN;   // (wave-uniform)
  const int lane = threadIdx.x & 63;
  int i0[3]; float u[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) dni_window(E, ip[k], fp[k], nsig, &i0[k], &u[k]);
  const int ref = min(i0[0], min(i0[1], i0[2]));
  const int span = max(i0[0], max(i0[1], i0[2])) + E.npts - 1 - ref;   // samples ref+1 .. ref+span
  const int ch = (span + 63) / 64;                                       // consecutive samples per lane
  constexpr int CH = 8;
  float t = 0.f;
  const DniRow row = dni_row(Bt, lane);   // (64 rows in the table; rows beyond the window are zero)
  if (scratch && ch <= CH) {   // (wave-uniform)
    float v[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int i = ref + 1 + lane * ch + j;
      v[j] = (j < ch && i <= ref + span) ? Y[i] : 0.f;
    }
#pragma unroll
    for (int j = 1; j < CH; ++j) v[j] += v[j - 1];
    float incl = v[CH - 1];
    LDSP_DPP_GROUP1("v_add_f32_dpp", incl);
    const float excl = incl - v[CH - 1];
#pragma unroll
    for (int j = 0; j < CH; ++j)
      if (j < ch) scratch[lane * ch + j] = excl + v[j];   // D(ref + 1 + o) at scratch[o]
    // (the wave's own LDS writes and reads are ordered)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = (lane < E.npts) ? i0[k] + lane : ref;
      const float d = (i > ref) ? scratch[i - ref - 1] : 0.f;
      if (lane < E.npts) t = fmaf(dni_poly(row, u[k]) * ((k == 1) ? -2.f : 1.f), d, t);   // E1 - 2 E2 + E3
    }
  } else {
    float loc = 0.f;
    for (int j = 0; j < ch; ++j) { const int i = ref + 1 + lane * ch + j; if (i <= ref + span) loc += Y[i]; }
    float incl = loc;
    LDSP_DPP_GROUP1("v_add_f32_dpp", incl);
    const float excl = incl - loc;   // sum of the chunks before this lane's
    // lane l < npts: point l of each window in turn.  D(i) for i = ref + 1 + c*ch + j: the chunks before c + the first j+1
    // samples of chunk c (every lane takes part in the shuffle)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = (lane < E.npts) ? i0[k] + lane : ref;
      const int o = max(i - ref - 1, 0), c = ch > 0 ? o / ch : 0, jj = o - c * ch;
      float d = __shfl(excl, c);
      for (int j = 0; j <= jj; ++j) d += Y[ref + 1 + c * ch + j];
      if (i <= ref) d = 0.f;
      if (lane < E.npts) t = fmaf(dni_poly(row, u[k]) * ((k == 1) ? -2.f : 1.f), d, t);   // E1 - 2 E2 + E3
    }
  }
  LDSP_DPP_GROUP1("v_add_f32_dpp", t);
  return readlane_f(t, 63);
}

}  // namespace ldsp
