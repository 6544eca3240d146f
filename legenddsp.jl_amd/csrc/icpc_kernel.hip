// icpc_kernel.hip — fused dsp_icpc for gfx950 (MI355X).
//
// One workgroup per trace.  The trace is read from HBM exactly once (coalesced
// 16 B/lane straight into registers) and every one of the 48 output columns of
// reference src/dsp_icpc.jl:62-230 is produced from registers / LDS; the only HBM
// writes are the 48 x 4 B of the output row.  No MFMA: these are 1-D recursions
// and sliding sums, not contractions.
//
//   phase 0  load (S4 view: thread t, row r holds samples 4(t+NT r)..+3)
//   phase 1  saturation, signalstats(bl), shift, max/min, tailstats,
//            pole-zero as prefix scan  y = x + c*cumsum(x),  signalstats(tail)
//   phase 2  T = prefix sum of y; y, T -> LDS (linear)
//   phase 3  LS sweep: 5 trapezoids from T (+ inverted outputs by linearity),
//            threshold bit-masks by wave ballot; Intersect scans on the masks
//   phase 3c signal estimators (e_trap, qdrift, lq)
//   phase 4  Savitzky-Golay derivative x3 + plain derivative, current maxima,
//            in-trace pile-up, t50_current
//   phase 5  CUSP / ZAC: closed-form sliding exponential / parabola sums
//            (causal + anti-causal one-pole scans, f64 double prefix sum), or the
//            direct-form FIR comparator (cusp_mode 0)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <type_traits>
#include "icpc_dev.hpp"
#include "ldsp_device.hpp"
#include "qdrift.hpp"

// LDSP_DEV_512: development builds instantiate the 512-thread kernels only (one fifth of the compile time); the dev
// library answers LDSP_ERR for other trace lengths.  LDSP_STAMPS: diagnostic build, every wave of the first blocks writes
// s_memtime at phase boundaries into IcpcDev::dbg_stamps (tools/stamp_map.py); never defined in the shipped library.
#ifdef LDSP_DEV_512
#define LDSP_ALL_CASES LDSP_CASE(512)
#else
#define LDSP_ALL_CASES LDSP_CASE(64) LDSP_CASE(128) LDSP_CASE(256) LDSP_CASE(512) LDSP_CASE(1024)
#endif
#ifdef LDSP_STAMPS
#define STAMP(id) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < LDSP_STAMP_BLOCKS && P.dbg_stamps) \
    P.dbg_stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * LDSP_STAMP_SLOTS + (id)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(id) do { } while (0)
#endif

namespace ldsp {

namespace {

constexpr int NMASK = 9;  // t10,t50,t80,t90,t99, t0, t0inv, sg50, intrace
enum { M_T10, M_T50, M_T80, M_T90, M_T99, M_T0, M_T0INV, M_SG50, M_INTR };

struct Pos {  // fractional sample position ip + fp
  int ip;
  float fp;
};
__device__ __forceinline__ Pos pos_norm(Pos p) {
  float f = floorf(p.fp);
  p.ip += (int)f;
  p.fp -= f;
  return p;
}
__device__ __forceinline__ Pos pos_add(Pos p, float d) {
  float di = floorf(d);
  p.ip += (int)di;
  p.fp += d - di;
  return pos_norm(p);
}

// reduction slots in LDS (each used once per trace; set to identities in phase 0)
struct Slots {
  unsigned long long vi[8];  // packed (value,index) maxima: opt trap, 4 current windows, cusp, zac
  uint32_t fmx[8];           // float maxima as ordered uints: e_max raw, 3 fixed traps, gmax
  uint32_t fmn[4];           // float minima as ordered uints: e_min raw, 2 inverted traps
  int isum[12];              // n_low, n_high, tail_bad, 9 intersect counts
  int imin[9];               // first crossing per mask
  int imax[1];               // last run end (in-trace pile-up)
};
enum { VI_OPT, VI_CUR0, VI_CUR1, VI_CUR2, VI_CUR3, VI_CUSP, VI_ZAC };
enum { FX_RAW, FX_F0, FX_F1, FX_F2, FX_G, FX_F0I, FX_F2I };  // ..I: maxima of the inverted trapezoid outputs
enum { FN_RAW };
enum { IS_LOW, IS_HIGH, IS_TAILBAD, IS_CNT0 };
constexpr int EST_TBL = LDSP_MAX_EST_PTS * (LDSP_MAX_EST_DEG + 1);
constexpr int NSUM = 16;  // deterministic f64 sum sites x NW wave partials

// LDS layout of one trace.  Between the two sample arrays lies a gap of `gap` floats: kernel 1
// keeps its threshold bit-masks there; the CUSP/ZAC stage zero-fills it as the Dp[i < 0] = 0
// margin in front of B1 (the ZAC parabola taps reach back Lf+2 samples).
template <int NT, int R, bool MASKS = true>
struct Smem {
  static constexpr int NW = NT / 64, SP = 4 * R, Lp = NT * SP, NWORDS = Lp / 32;
  static constexpr int MASK_FLOATS = MASKS ? NMASK * NWORDS : 0;
  float* B0;      // [Lp]      y, later scratch
  uint32_t* bm;   // [NMASK][NWORDS]   (inside the gap)
  float* B1;      // [Lp+64]   T = exclusive prefix sum of y, later scratch
  double* part;   // [2][R*NW] scan partials (alternating buffers)
  double* wsum;   // [NSUM][NW] per-wave partial sums, combined in fixed order
  Slots* sl;
  float* ylast;   // [R*NW]    (unused)
  float* outv;    // [C_NCOLS]  output row, filled as results become available
  float* misc;    // [16]       small broadcasts
  float* estB;    // [2][EST_TBL] LSQ basis tables of sig_est and int_est
  static constexpr size_t bytes(int gap) {
    return (size_t)(Lp + Lp + 64 + gap) * 4 + 2 * R * NW * 8 + NSUM * NW * 8 + sizeof(Slots) +
           R * NW * 4 + C_NCOLS * 4 + 16 * 4 + 2 * EST_TBL * 4 + 64;
  }
  __device__ explicit Smem(unsigned char* raw, int gap) {
    B0 = reinterpret_cast<float*>(raw);
    bm = reinterpret_cast<uint32_t*>(B0 + Lp);
    B1 = B0 + Lp + gap;
    part = reinterpret_cast<double*>(B1 + Lp + 64);
    wsum = part + 2 * R * NW;
    sl = reinterpret_cast<Slots*>(wsum + NSUM * NW);
    ylast = reinterpret_cast<float*>(sl + 1);
    outv = ylast + R * NW;
    misc = outv + C_NCOLS;
    estB = misc + 16;
  }
};

// the same with a first leg of <= 3 samples summed from the samples themselves (as in sweep A: on the tail a difference of two
// float prefix sums of 1e8 is good to a few counts only, the size of the t0 threshold)
__device__ __forceinline__ float trap_at_y(const float* T, const float* y, int k, const TrapDev& t) {
  const float a = T[k + t.flen] - T[k + t.n1 + t.g];
  float b;
  if (t.n1 <= 3) {
    b = y[k];
    if (t.n1 >= 2) b += y[k + 1];
    if (t.n1 >= 3) b += y[k + 2];
  } else {
    b = T[k + t.n1] - T[k];
  }
  return a * t.inv2 - b * t.inv1;
}
__device__ __forceinline__ float trap_at(const float* T, int k, const TrapDev& t) {
  float a = T[k + t.flen] - T[k + t.n1 + t.g];
  float b = T[k + t.n1] - T[k];
  return a * t.inv2 - b * t.inv1;
}

// Bt: the estimator's basis table staged in LDS at kernel start (a per-lane read from the
// parameter block in global memory would put ~2 us of latency on the critical path)
__device__ __forceinline__ float est_weight(const EstDev& E, const float* Bt, int l, float u) {
  return dni_weight(E, Bt, l, u);   // qdrift.hpp: the whole coefficient row in one read
}
// LSQ-polynomial estimate at position p (index space of a signal of length nsig whose
// samples are produced by getval(i)); computed redundantly by every wave, lane l
// handles window point l.  Assumption A3.
template <typename F>
__device__ __forceinline__ float estimate(const EstDev& E, const float* Bt, Pos p, int nsig, F getval) {
  if (nsig < E.npts) return NAN;
  if (p.ip < 0) { p.ip = 0; p.fp = 0.f; }
  if (p.ip >= nsig - 1) { p.ip = nsig - 1; p.fp = 0.f; }
  int i0 = p.ip + (int)ceilf(p.fp - 0.5f * (float)E.npts);  // ceil(p - npts/2)
  i0 = max(0, min(i0, nsig - E.npts));
  float u = ((float)(p.ip - i0) + p.fp - E.c) * E.s_inv;
  const int l = lane_id();
  float v = 0.f;
  if (l < E.npts) v = est_weight(E, Bt, l, u) * getval(i0 + l);
  return wave_sum_all(v);
}

// Window statistics (signalstats / tailstats sums).  All threads of a trace sum
// d = v - pivot about ONE block-uniform pivot (a sample of the window itself, so |d| is
// the spread of the window, not its level): float partial sums per thread and per wave
// (fused DPP adds), the <= 16 wave partials are combined in double in a fixed order and the
// pivot is added back in double.  Mathematically the oracle's  sum(Y^2)/n - mean^2.
struct WinAcc {   // sums over the window: sum d, sum d^2, sum xi*d  (xi = i - ic, d = v - pivot)
  double s1, s2, s3;
};
struct WinAccF {
  float s1, s2, sx;
};
__device__ __forceinline__ void winf_accum(WinAccF& a, const WinDev& w, int i, float xi, float pivot, float v) {
  const bool in = (i >= w.from) && (i <= w.until);
  const float d = in ? v - pivot : 0.f;
  a.s1 += d;
  a.s2 = fmaf(d, d, a.s2);
  a.sx = fmaf(xi, d, a.sx);
}
// four consecutive samples i0..i0+3 at once; quads outside the window cost one test,
// quads wholly inside run without masks
__device__ __forceinline__ void winf_accum4(WinAccF& a, const WinDev& w, int i0, float xi0, float pivot, float v0, float v1,
                                            float v2, float v3) {
  const int lo = w.from - i0, hi = w.until - i0;  // in-window e in [lo, hi]
  if (lo > 3 || hi < 0) return;
  float d[4] = {v0 - pivot, v1 - pivot, v2 - pivot, v3 - pivot};
  if (lo > 0 || hi < 3) {
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = (e >= lo && e <= hi) ? d[e] : 0.f;
  }
  a.s1 += (d[0] + d[1]) + (d[2] + d[3]);
  a.s2 = fmaf(d[0], d[0], fmaf(d[1], d[1], fmaf(d[2], d[2], fmaf(d[3], d[3], a.s2))));
  // sum (xi0+e) d_e = xi0 * sum d + (d1 + 2 d2 + 3 d3)
  a.sx = fmaf(xi0, (d[0] + d[1]) + (d[2] + d[3]), a.sx + fmaf(3.f, d[3], fmaf(2.f, d[2], d[1])));
}
// (mean, sigma, slope per time unit, offset) from window sums — the arithmetic of
// signalstats (RadiationDetectorDSP; restated in oracle/ldsp_oracle.c:orc_signalstats)
__device__ __forceinline__ void win_finish(const WinAcc& a, const WinDev& w, float pivot, float t_first, float dt, float* mean,
                                           float* sigma, float* slope, float* offset) {
  const double md = a.s1 * w.inv_n, m = (double)pivot + md;
  double var = a.s2 * w.inv_n - md * md;
  if (var < 0) var = 0;
  double cov = a.s3 * w.inv_n;  // mean of xi is 0, so the pivot drops out
  double sl_t = cov / w.var_i / (double)dt;
  double mean_x = (double)t_first + w.ic * (double)dt;
  *mean = (float)m;
  *sigma = (float)sqrt(var);
  *slope = (float)sl_t;
  *offset = (float)(m - sl_t * mean_x);
}
// per-wave partials of a window accumulator -> wsum[site..site+2][wave]
template <int NW>
__device__ __forceinline__ void win_publish(const WinAccF& a, double* wsum, int site) {
  const float s1 = wave_incl_scan_sum(a.s1), s2 = wave_incl_scan_sum(a.s2), s3 = wave_incl_scan_sum(a.sx);
  if (lane_id() == 63) {
    const int w = wave_id();
    wsum[(site + 0) * NW + w] = (double)s1;
    wsum[(site + 1) * NW + w] = (double)s2;
    wsum[(site + 2) * NW + w] = (double)s3;
  }
}
template <int NW>
__device__ __forceinline__ double win_collect1(const double* wsum, int slot) {  // one of a site's three sums
  double a = 0;
  for (int w = 0; w < NW; ++w) a += wsum[slot * NW + w];
  return a;
}
template <int NW>
__device__ __forceinline__ WinAcc win_collect(const double* wsum, int site) {
  WinAcc a = {0, 0, 0};
  for (int w = 0; w < NW; ++w) {
    a.s1 += wsum[(site + 0) * NW + w];
    a.s2 += wsum[(site + 1) * NW + w];
    a.s3 += wsum[(site + 2) * NW + w];
  }
  return a;
}

// filter f at output index k from y in LDS: f = 0..2 Savitzky-Golay derivative
// (correlation taps c), f = 3 DerivativeFilter(1): y[max(i,1)] - y[max(i-1,0)] (src/derivative.jl:47-55)
__device__ __forceinline__ float flt_eval(const float* y, const float* c, int np, int f, int k) {
  if (f < 3) {
    float g = 0.f;
    const float* yp = y + k;
    for (int i = 0; i < np; ++i) g = fmaf(c[i], yp[i], g);
    return g;
  }
  return y[max(k, 1)] - y[max(k - 1, 0)];
}
__device__ __forceinline__ float flt_eval_rare(const float* y, const float* c, int np, int f, int k) {
  return flt_eval(y, c, np, f, k);
}

}  // namespace

// ---------------------------------------------------------------------------
// zero-filled space in front of the Dp array: the ZAC parabola taps reach back Lf+2 samples
__host__ __device__ inline int cz_pad(const IcpcDev& P) { return ((P.cusp.Lf > P.zac.Lf ? P.cusp.Lf : P.zac.Lf) + 2 + 7) & ~3; }

// ---------------------------------------------------------------------------
// CUSP and ZAC (reference src/dsp_icpc.jl:167-178) on the register-resident, pole-zero corrected
// trace y (S4 view; B0 holds the same y).  Called at the end of icpc_kernel (fused, the normal
// path) or from icpc_cz_kernel (second launch: direct-form comparator, filters with different
// geometry, or a gap too large for two workgroups per CU).  Requires: the gap in front of B1
// zero-filled or about to be (a barrier follows inside), B1[Lp..Lp+63] = 0, slots fmx[0..1] = 0
// and imin[0..1] = INT_MAX.  Fills the six CUSP/ZAC entries of S.outv.
// WANT_C / WANT_Z: which filters this pass evaluates.  Both = they share sigma / flat /
// length / tau (one set of recursions).
template <int NT, int R, bool FULL, bool DIRECT, bool WANT_C, bool WANT_Z, bool T_READY, typename SM>
__device__ __forceinline__ void cz_body(SM& S, const IcpcDev& P, float (&y)[R][4], Pos ptx1, int& scan_buf) {
  constexpr int NW = SM::NW, SP = SM::SP, Lp = SM::Lp;
  const int L = FULL ? Lp : P.L, tid = threadIdx.x;
  const int lane = lane_id(), wave = wave_id();
  auto part_buf = [&]() { double* p = S.part + (scan_buf & 1) * R * NW; ++scan_buf; return p; };
  auto put = [&](int c, float v) { if (tid == 0) S.outv[c] = v; };
  Pos ptx[2];
  ptx[1] = ptx1;
  if (P.dbg_stop == 11) return;  // profiling aid (tools/gpu_phase_time.py): stops 11..15 inside this kernel
  // y just before each of the thread's chunks (for d[i] = y[i] - a*y[i-1]); B0 = y here
  float yprev[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i0 = 4 * (tid + NT * r);
    yprev[r] = (i0 > 0) ? S.B0[i0 - 1] : 0.f;
  }
  // Pivot (closed form only).  The parts of the closed form (flat top, sinh flanks, parabolas) each answer a constant level c
  // under the filter with a multiple of c * Lf that cancels between them only in exact arithmetic: a trace whose baseline
  // sits 2000 counts off zero (pile-up in the baseline window) loses 1e-4 of its ZAC energy to float rounding.  So the
  // stage runs on y - cpiv, cpiv = the level at the left edge of the pick-off window, and adds cpiv * hsum back at the end
  // (hsum = the sum of the direct-form taps; exact for any constant), which makes every trace look like a clean one.
  float cpiv = 0.f, y0_piv = 0.f;
  if constexpr (!DIRECT) {
    const int Lf0 = (WANT_C ? P.cusp : P.zac).Lf;
    const Pos pp = pos_add(ptx1, WANT_C ? P.cusp_pickoff : P.zac_pickoff);
    const int ic = max(0, min(pp.ip - (Lf0 - 1), L - 1));
    cpiv = S.B0[ic];
    y0_piv = S.B0[0] - cpiv;
    __syncthreads();   // every read of the unshifted y in B0 is done
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i0 = 4 * (tid + NT * r);
#pragma unroll
      for (int e = 0; e < 4; ++e) y[r][e] = (FULL || i0 + e < L) ? y[r][e] - cpiv : 0.f;
      yprev[r] = (i0 > 0) ? yprev[r] - cpiv : 0.f;
      *reinterpret_cast<float4*>(&S.B0[i0]) = make_float4(y[r][0], y[r][1], y[r][2], y[r][3]);
    }
  }
  // extremestats + SignalEstimator on filter outputs held in the LS view
  // (acc[m] = out[tid + NT*m]); f = 0 CUSP, 1 ZAC.          dsp_icpc.jl:170-171,177-178
  // Three steps with a barrier between them: finish_publish (per-wave estimator partial, the
  // maximum VALUE into an LDS slot), finish_locate (only threads holding that value look up its
  // first index), finish_collect.
  auto finish_publish = [&](int f, int Lf, const float (&acc)[SP], float* my_max) {
    const int nout = L - Lf + 1;
    float bv = -INFINITY;
#pragma unroll
    for (int m = 0; m < SP; m += 2) {
      if (NT * (m + 2) <= nout) bv = vmax3(bv, acc[m], acc[m + 1]);   // both rows wholly inside
      else if (NT * m < nout)
        bv = vmax3(bv, (tid + NT * m < nout) ? acc[m] : -INFINITY, (tid + NT * (m + 1) < nout) ? acc[m + 1] : -INFINITY);
    }
    *my_max = bv;
    // estimator window [i0, i0+npts): at most one output per thread (npts <= 64 <= NT)
    float part = 0.f;
    if (nout >= P.sig_est.npts) {
      Pos p = pos_add(ptx[1], f ? P.zac_pickoff : P.cusp_pickoff);
      p.ip -= (Lf - 1);
      if (p.ip < 0) { p.ip = 0; p.fp = 0.f; }
      if (p.ip >= nout - 1) { p.ip = nout - 1; p.fp = 0.f; }
      int i0 = p.ip + (int)ceilf(p.fp - 0.5f * (float)P.sig_est.npts);
      i0 = max(0, min(i0, nout - P.sig_est.npts));
      const float u = ((float)(p.ip - i0) + p.fp - P.sig_est.c) * P.sig_est.s_inv;
      const int ms = (i0 - tid + NT - 1) / NT;   // smallest m with tid + NT*m >= i0
      const int l = tid + NT * ms - i0;
      if (l >= 0 && l < P.sig_est.npts && ms >= 0 && ms < SP) {  // at most two waves get here
        float val = 0.f;
#pragma unroll
        for (int m = 0; m < SP; ++m) val = (m == ms) ? acc[m] : val;
        part = est_weight(P.sig_est, S.estB, l, u) * val;
      }
    }
    const float ps = wave_incl_scan_sum(part);   // <= 64 terms, pairwise; waves are combined in double
    const float bw = wave_max_all(bv);
    if (lane == 63) S.wsum[(12 + f) * NW + wave] = (double)ps;
    if (lane == 0) atomicMax(&S.sl->fmx[f], ford(bw));
  };
  auto finish_locate = [&](int f, int Lf, const float (&acc)[SP], float my_max) {
    const int nout = L - Lf + 1;
    const float vmax = ford_inv(S.sl->fmx[f]);
    if (my_max == vmax) {   // findmax: first occurrence
      int bi = 0x7fffffff;
#pragma unroll
      for (int m = SP - 1; m >= 0; --m)
        if (tid + NT * m < nout && acc[m] == vmax) bi = tid + NT * m;
      atomicMin(&S.sl->imin[f], bi);
    }
  };
  auto finish_collect = [&](int f, int Lf) {
    const int nout = L - Lf + 1;
    double s = 0;
    for (int ww = 0; ww < NW; ++ww) s += S.wsum[(12 + f) * NW + ww];
    const float v = ford_inv(S.sl->fmx[f]);
    const int i = S.sl->imin[f];
    const double back = (double)cpiv * (f ? P.zac.hsum : P.cusp.hsum);   // the pivot's share (estimator weights sum to one)
    put(f ? C_e_zac : C_e_cusp, (nout >= P.sig_est.npts) ? (float)(s + back) : NAN);
    put(f ? C_e_zac_max : C_e_cusp_max, (float)((double)v + back));
    put(f ? C_t_zac_max : C_t_cusp_max, P.t_first + P.dt * (float)(i + Lf - 1));
  };

  if constexpr (DIRECT) {
    // direct-form FIR comparator: out[k] = sum_j h[j] y[k+Lf-1-j]
    for (int f = WANT_C ? 0 : 1; f < (WANT_Z ? 2 : 1); ++f) {
      const CuspZacDev& Z = f ? P.zac : P.cusp;
      const float* h = f ? P.h_zac : P.h_cusp;
      const int Lf = Z.Lf, nout = L - Lf + 1;
      float acc[SP];
#pragma unroll
      for (int m = 0; m < SP; ++m) acc[m] = 0.f;
      for (int j = 0; j < Lf; ++j) {
        const float hj = h[j];
        const float* yp = &S.B0[tid + Lf - 1 - j];
#pragma unroll
        for (int m = 0; m < SP; ++m)
          if (tid + NT * m < nout) acc[m] = fmaf(hj, yp[NT * m], acc[m]);
      }
      float my_max;
      finish_publish(f, Lf, acc, &my_max);
      __syncthreads();
      finish_locate(f, Lf, acc, my_max);
    }
    __syncthreads();
    if (WANT_C) finish_collect(0, P.cusp.Lf);
    if (WANT_Z) finish_collect(1, P.zac.Lf);
  } else {
    // Closed form.  With d[i] = y[i] - a*y[i-1] (a = exp(-1/tau)):
    //   out[k] = sc * ( sum_{j<=Lf-2} w[j] d[n-j] + w[Lf-1] y[k] ),  n = k+Lf-1
    // and w = sinh flanks + flat top (+ parabolas for ZAC) splits into
    //   G[i] = sum_m q^m d[i-m]   causal one-pole      (S4 scan, forward)
    //   A[i] = sum_m q^m d[i+m]   anti-causal one-pole (S4 scan, backward)
    //   Dp[i] = sum_{m<=i} d[m] = y[i]-y[0]+eps*T[i]   (flat top; eps = 1-a)
    //   PRF = double prefix sum of a sparse combination of Dp (ZAC parabolas, in f64)
    // each read back in the LS view at a handful of fixed shifts.  B0/B1 are recycled.
    {
      constexpr bool want_c = WANT_C, want_z = WANT_Z;
      const CuspZacDev& Z = WANT_C ? P.cusp : P.zac;       // geometry + exponentials of this launch
      const CuspZacDev& ZZ = P.zac;                         // parabola constants
      const int Lf = Z.Lf, nout = L - Lf + 1, lt = Z.lt, f1 = Z.f1, ltp = Z.ltp;
      // ---- step A0 (S4): Dp[i] = y[i]-y[0]+eps*T[i] -> B1.  T_READY (fused): B1 still holds T, each
      // thread converts its own quads in place; otherwise T is re-derived by a scan of the
      // register-resident y.
      if constexpr (T_READY) {
        const float y0 = y0_piv;               // (B0[0] is being rewritten by thread 0)
        const float mec = -Z.eps * cpiv;       // B1 holds the exclusive prefix sum of the unshifted y: T'[i] = T[i] - cpiv*i
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float4 t = *reinterpret_cast<const float4*>(&S.B1[4 * (tid + NT * r)]);
          const float fi = (float)(4 * (tid + NT * r));
          t.x = fmaf(Z.eps, t.x, y[r][0] - y0) + mec * fi; t.y = fmaf(Z.eps, t.y, y[r][1] - y0) + mec * (fi + 1.f);
          t.z = fmaf(Z.eps, t.z, y[r][2] - y0) + mec * (fi + 2.f); t.w = fmaf(Z.eps, t.w, y[r][3] - y0) + mec * (fi + 3.f);
          *reinterpret_cast<float4*>(&S.B1[4 * (tid + NT * r)]) = t;
        }
      } else {
        if (tid == 0) S.misc[2] = y[0][0];  // y[0] for everyone
        float tot[R];
        double t_off[R];
#pragma unroll
        for (int r = 0; r < R; ++r) tot[r] = (y[r][0] + y[r][1]) + (y[r][2] + y[r][3]);
        s4_exscan_sum<NT, R>(tot, t_off, part_buf(), nullptr);  // barrier inside
        const float y0 = S.misc[2];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          double run = t_off[r];
          float4 v;
          float* pv = &v.x;
#pragma unroll
          for (int e = 0; e < 4; ++e) { pv[e] = (y[r][e] - y0) + Z.eps * (float)run; run += (double)y[r][e]; }
          *reinterpret_cast<float4*>(&S.B1[4 * (tid + NT * r)]) = v;
        }
      }
      __syncthreads();
      STAMP(16);
      // ---- step A1 (LS): flat top + last tap; ZAC: u[n] -> B0 in place
      float ac[SP], dz[SP];
      {
        const float dwl = want_c ? (ZZ.w_last - Z.w_last) : 0.f;  // shared pass: ZAC last tap relative to CUSP's
        const float wl = want_c ? Z.w_last : ZZ.w_last;
        const float sc = Z.sc;
#pragma unroll
        for (int m = 0; m < SP; ++m) {
          const int k = tid + NT * m;
          float a = 0.f, z = 0.f;
          if (k < nout) {
            const float yk = S.B0[k];
            const int n = k + Lf - 1;
            a = sc * (S.B1[n - lt] - S.B1[n - f1]) + wl * yk;
            z = dwl * yk;
          }
          ac[m] = a; dz[m] = z;
          if ((m & 1) == 1) asm volatile("" ::: "memory");  // keep at most two iterations of loads in flight
        }
        if (want_z) {
          // u[n] = sum_e coef_e * Dp[n - shift_e] (Dp = 0 at negative indices: the pad), for every n.
          // Tap-outer / row-inner: one address per tap, immediate row offsets.  Overwrites y[k] in
          // B0 — only this thread ever read that element (above), so in place is race-free.
          float u[SP];
#pragma unroll
          for (int m = 0; m < SP; ++m) u[m] = 0.f;
          const int nz = ZZ.zu_n;
          for (int e = 0; e < nz; ++e) {
            const float ce = ZZ.zu_coef[e];
            const float *dp = &S.B1[tid - ZZ.zu_shift[e]], *dq = &S.B1[tid - ZZ.zu_shift_b[e]];
#pragma unroll
            for (int m = 0; m < SP; ++m) u[m] = fmaf(ce, dp[NT * m] - dq[NT * m], u[m]);
          }
#pragma unroll
          for (int m = 0; m < SP; ++m) S.B0[tid + NT * m] = (tid + NT * m < L) ? u[m] : 0.f;
        }
      }
      if (P.dbg_stop == 12) return;
      STAMP(17);
      // ---- d[i] = (y[i]-y[i-1]) + eps*y[i-1] for 1 <= i < L, else 0   (S4)
      float d[R][4];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i0 = 4 * (tid + NT * r);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float yp = (e == 0) ? yprev[r] : y[r][(e + 3) & 3];
          const int i = i0 + e;
          d[r][e] = (i >= 1 && i < L) ? (y[r][e] - yp) + Z.eps * yp : 0.f;
        }
      }
      const float q1 = Z.qp1[1];
      // ---- step B: causal one-pole G -> B1, rise(-) and fall(+) exponentials
      {
        float gl[R][4], b[R], s_in[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float g = d[r][0]; gl[r][0] = g;
          g = fmaf(q1, g, d[r][1]); gl[r][1] = g;
          g = fmaf(q1, g, d[r][2]); gl[r][2] = g;
          g = fmaf(q1, g, d[r][3]); gl[r][3] = g;
          b[r] = g;
        }
        s4_exscan_affine_fwd<NT, R>(b, s_in, Z.qp4, Z.qpw, reinterpret_cast<float*>(part_buf()));  // barrier: A1 reads of B1 done
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float4 v;
          v.x = fmaf(Z.qp1[1], s_in[r], gl[r][0]);
          v.y = fmaf(Z.qp1[2], s_in[r], gl[r][1]);
          v.z = fmaf(Z.qp1[3], s_in[r], gl[r][2]);
          v.w = fmaf(Z.qp1[4], s_in[r], gl[r][3]);
          *reinterpret_cast<float4*>(&S.B1[4 * (tid + NT * r)]) = v;
        }
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < SP; ++m) {
        const int k = tid + NT * m;
        if (k < nout) {
          const int n = k + Lf - 1;
          const float pm = S.B1[n] - Z.q_lt * S.B1[n - lt];
          const float fp = Z.q_mltp * (S.B1[k + ltp - 1] - Z.q_ltp1 * S.B1[k]);
          ac[m] = fmaf(Z.sc_half_den, fp - pm, ac[m]);
        }
        if ((m & 1) == 1) asm volatile("" ::: "memory");
      }
      if (P.dbg_stop == 13) return;
      STAMP(18);
      // ---- step C: anti-causal one-pole A -> B1, rise(+) and fall(-) exponentials
      {
        float al[R][4], b[R], s_in[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float a = d[r][3]; al[r][3] = a;
          a = fmaf(q1, a, d[r][2]); al[r][2] = a;
          a = fmaf(q1, a, d[r][1]); al[r][1] = a;
          a = fmaf(q1, a, d[r][0]); al[r][0] = a;
          b[r] = a;
        }
        s4_exscan_affine_bwd<NT, R>(b, s_in, Z.qp4, Z.qpw, reinterpret_cast<float*>(part_buf()));  // barrier: G reads done
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float4 v;
          v.x = fmaf(Z.qp1[4], s_in[r], al[r][0]);
          v.y = fmaf(Z.qp1[3], s_in[r], al[r][1]);
          v.z = fmaf(Z.qp1[2], s_in[r], al[r][2]);
          v.w = fmaf(Z.qp1[1], s_in[r], al[r][3]);
          *reinterpret_cast<float4*>(&S.B1[4 * (tid + NT * r)]) = v;
        }
        if (tid == 0) S.B1[Lp] = 0.f;  // A[L] when L == Lp
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < SP; ++m) {
        const int k = tid + NT * m;
        if (k < nout) {
          const int n = k + Lf - 1;
          const float pp = Z.q_mlt1 * S.B1[n - lt + 1] - Z.q1 * S.B1[n + 1];
          const float fm = Z.q2 * (S.B1[k + 1] - Z.q_ltp1 * S.B1[k + ltp]);
          ac[m] = fmaf(Z.sc_half_den, pp - fm, ac[m]);
        }
        if ((m & 1) == 1) asm volatile("" ::: "memory");
      }
      if (P.dbg_stop == 14) return;
      STAMP(19);
      if (want_z) {
        // ---- step A2 (S4): PRF = cumsum(cumsum(u)).  Last, when y / d / G / A registers are dead.
        // (u was parked in B0 by step A1; B0 has not been touched since.)  Two levels: inside a
        // wave-row (256 samples) the single and double running sums l1, l2 start from zero and
        // stay in float (|l2| <= 2^15 |u|: its rounding is far below the float rounding of the
        // result); the state entering each wave-row, (C1, C2), is carried in double:
        //   c2[j] = C2 + (j+1)*C1 + l2[j],   C1' = C1 + l1[255],  C2' = C2 + 256*C1 + l2[255].
        {
          float ex1[R], ex2[R];
          double* pa = part_buf();
          double* pb = part_buf();
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float4 v = *reinterpret_cast<const float4*>(&S.B0[4 * (tid + NT * r)]);
            const float p0 = v.x, p1 = p0 + v.y, p2 = p1 + v.z, p3 = p2 + v.w;
            const float i1 = wave_incl_scan_sum(p3);
            ex1[r] = i1 - p3;
            const float wv = fmaf(4.f, ex1[r], (p0 + p1) + (p2 + p3));
            const float i2 = wave_incl_scan_sum(wv);
            ex2[r] = i2 - wv;
            if (lane == 63) { pa[r * NW + wave] = (double)i1; pb[r * NW + wave] = (double)i2; }
          }
          __syncthreads();
          // exclusive scans of the wave-row totals (R*NW <= 64 lanes), C2 with the 256*C1 term
          const double t1 = (lane < R * NW) ? pa[lane] : 0.0;
          const double c1x = wave_incl_scan_sum_f64(t1) - t1;
          const double t2 = (lane < R * NW) ? pb[lane] + 256.0 * c1x : 0.0;
          const double c2x = wave_incl_scan_sum_f64(t2) - t2;
          const double mrho = -(double)ZZ.rho_sc;
          const double jd0 = (double)(4 * lane + 1);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const double C1 = readlane_d(c1x, r * NW + wave), C2 = readlane_d(c2x, r * NW + wave);
            float4 v = *reinterpret_cast<const float4*>(&S.B0[4 * (tid + NT * r)]);  // u again (own chunk)
            float* pv = &v.x;
            double t = fma(C1, jd0, C2);   // C2 + (j+1)*C1 at the chunk's first sample
            float l1 = ex1[r], l2 = ex2[r];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              l1 += pv[e]; l2 += l1;
              pv[e] = (float)(mrho * (t + (double)l2));
              t += C1;
            }
            *reinterpret_cast<float4*>(&S.B0[4 * (tid + NT * r)]) = v;
          }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < SP; ++m) {
          const int k = tid + NT * m;
          if (k < nout) dz[m] += S.B0[k + Lf - 1];
          if ((m & 3) == 3) asm volatile("" ::: "memory");
        }
      }
      if (P.dbg_stop == 15) return;
      STAMP(20);
      float mxc = 0.f, mxz = 0.f;
      if (want_z) {
#pragma unroll
        for (int m = 0; m < SP; ++m) dz[m] += ac[m];
      }
      if (want_c) finish_publish(0, Lf, ac, &mxc);
      if (want_z) finish_publish(1, Lf, dz, &mxz);
      __syncthreads();
      if (want_c) finish_locate(0, Lf, ac, mxc);
      if (want_z) finish_locate(1, Lf, dz, mxz);
    }
    STAMP(21);
    __syncthreads();
    if (WANT_C) finish_collect(0, P.cusp.Lf);
    if (WANT_Z) finish_collect(1, P.zac.Lf);
  }
}

// Second-launch form of the CUSP/ZAC stage: re-reads the trace, takes blmean and the t50 position
// from kernel 1's side buffer and rebuilds y = x - blmean + c*cumsum exactly as kernel 1 does.
template <int NT, int R, bool FULL, bool DIRECT, bool WANT_C, bool WANT_Z>
__global__ void __launch_bounds__(NT, 4)
icpc_cz_kernel(const float* __restrict__ wf, const IcpcDev* __restrict__ Pp, const float* __restrict__ aux, IcpcOutDev out) {
  using SM = Smem<NT, R, false>;
  constexpr int NW = SM::NW, SP = SM::SP, Lp = SM::Lp;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const IcpcDev& P = *Pp;
  const int L = FULL ? Lp : P.L, tid = threadIdx.x;   // FULL: trace length == the tile, every bounds test folds
  const int lane = lane_id(), wave = wave_id();
  const int pad = cz_pad(P);
  SM S(smem_raw, pad);
  for (int i = tid; i < pad; i += NT) S.B1[i - pad] = 0.f;  // Dp[i < 0] = 0
  const float* w = wf + (size_t)blockIdx.x * (size_t)L;
  int scan_buf = 0;
  auto part_buf = [&]() { double* p = S.part + (scan_buf & 1) * R * NW; ++scan_buf; return p; };
  auto put = [&](int c, float v) { if (tid == 0) S.outv[c] = v; };
  const float blmean = aux[4 * (size_t)blockIdx.x + 0];
  Pos ptx[2];
  ptx[1].ip = __float_as_int(aux[4 * (size_t)blockIdx.x + 1]);
  ptx[1].fp = aux[4 * (size_t)blockIdx.x + 2];

  float y[R][4];
  if (P.in_u16) load_trace_s4_u16<NT, R, FULL>(reinterpret_cast<const uint16_t*>(wf) + (size_t)blockIdx.x * (size_t)L, L, tid, y);
  else load_trace_s4<NT, R, FULL>(w, L, tid, y);
  for (int i = tid; i < EST_TBL; i += NT) S.estB[i] = P.sig_est.B[i];
  if (tid < (int)(sizeof(Slots) / 4)) {  // fmx[0..1] (maxima: identity 0) and imin[0..1] (first index: identity INT_MAX) are used here
    const int o = tid * 4;
    reinterpret_cast<uint32_t*>(S.sl)[tid] = (o >= (int)offsetof(Slots, imin) && o < (int)offsetof(Slots, imax)) ? 0x7fffffffu : 0u;
  }
  if (tid < 64) S.B1[Lp + tid] = 0.f;
  // y = (x - blmean) + c*cumsum(x - blmean), exactly as kernel 1 computes it
  {
    float tot[R];
    double s_off[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i0 = 4 * (tid + NT * r);
#pragma unroll
      for (int e = 0; e < 4; ++e) y[r][e] = (i0 + e < L) ? y[r][e] - blmean : 0.f;
      tot[r] = (y[r][0] + y[r][1]) + (y[r][2] + y[r][3]);
    }
    s4_exscan_sum<NT, R>(tot, s_off, part_buf(), nullptr);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i0 = 4 * (tid + NT * r);
      const float coff = (float)(P.pz_c64 * s_off[r]);
      float run = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        run += y[r][e];
        y[r][e] = (i0 + e < L) ? (y[r][e] + coff) + P.pz_c * run : 0.f;
      }
      *reinterpret_cast<float4*>(&S.B0[i0]) = make_float4(y[r][0], y[r][1], y[r][2], y[r][3]);
    }
  }
  __syncthreads();
  cz_body<NT, R, FULL, DIRECT, WANT_C, WANT_Z, false>(S, P, y, ptx[1], scan_buf);
  if (tid < 6) {  // S.outv: stored and read by wave 0
    const int cols[6] = {C_e_cusp, C_e_zac, C_e_cusp_max, C_e_zac_max, C_t_cusp_max, C_t_zac_max};
    const int c = cols[tid];
    const bool mine = (tid & 1) ? WANT_Z : WANT_C;
    float* dst = reinterpret_cast<float*>(out.col[c]);
    if (dst && mine) dst[(size_t)blockIdx.x * (size_t)out.stride] = S.outv[c];
  }
}

// FUSE: the CUSP/ZAC stage runs at the end of this kernel on the same registers (one launch, one
// read of the trace, all 48 columns in one row store); otherwise blmean and the t50 position go
// to `aux` for icpc_cz_kernel.
template <int NT, int R, bool FULL, bool FUSE>
__global__ void __launch_bounds__(NT, (NT == 1024 && R == 2) ? 8 : 4)  // 4 waves/SIMD: <= 128 VGPRs, two 512-thread traces per CU; <1024, 2>: 8 waves/SIMD, <= 64 VGPRs, two 1024-thread traces per CU
icpc_kernel(const float* __restrict__ wf, const IcpcDev* __restrict__ Pp, float* __restrict__ aux, IcpcOutDev out,
            const float* __restrict__ ext_bl, float ext_bl_scale) {
  // ext_bl: dsp_icpc_compressed, windowed traces: the baseline handed over from the presummed traces (ext_bl[trace] *
  // ext_bl_scale replaces signalstats(bl).mean in shift_waveform, src/dsp_icpc.jl:353); NULL = the trace's own baseline
  using SM = Smem<NT, R>;
  constexpr int NW = SM::NW, SP = SM::SP, Lp = SM::Lp, NWORDS = SM::NWORDS;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const IcpcDev& P = *Pp;
  const int L = FULL ? Lp : P.L, tid = threadIdx.x;   // FULL: trace length == the tile, every bounds test folds
  const int lane = lane_id(), wave = wave_id();
  SM S(smem_raw, FUSE ? max(SM::MASK_FLOATS, cz_pad(P)) : SM::MASK_FLOATS);
  const float* w = wf + (size_t)blockIdx.x * (size_t)L;
  int scan_buf = 0;
  auto part_buf = [&]() { double* p = S.part + (scan_buf & 1) * R * NW; ++scan_buf; return p; };
  // results go to the LDS output row as soon as they exist (keeps them out of VGPRs)
  auto put = [&](int c, float v) { if (tid == 0) S.outv[c] = v; };
  auto puti = [&](int c, int v) { if (tid == 0) S.outv[c] = __int_as_float(v); };
  STAMP(0);

  // ------------------------------------------------------------ phase 0: load
  float x[R][4];
  const uint16_t* w16 = reinterpret_cast<const uint16_t*>(wf) + (size_t)blockIdx.x * (size_t)L;   // (in_u16)
  auto wv = [&](int i) { return P.in_u16 ? (float)w16[i] : w[i]; };
  if (P.in_u16) load_trace_s4_u16<NT, R, FULL>(w16, L, tid, x);
  else load_trace_s4<NT, R, FULL>(w, L, tid, x);
  for (int i = tid; i < 2 * EST_TBL; i += NT) S.estB[i] = (i < EST_TBL) ? P.sig_est.B[i] : P.int_est.B[i - EST_TBL];
  if (tid < (int)(sizeof(Slots) / 4)) {  // reduction slots: identities
    uint32_t* raw = reinterpret_cast<uint32_t*>(S.sl);
    const int o = tid * 4;
    uint32_t init = 0;
    if (o >= (int)offsetof(Slots, fmn) && o < (int)offsetof(Slots, isum)) init = 0xffffffffu;  // identity of min
    else if (o >= (int)offsetof(Slots, imin) && o < (int)offsetof(Slots, imax)) init = 0x7fffffffu;
    else if (o >= (int)offsetof(Slots, imax)) init = 0xffffffffu;  // -1
    raw[tid] = init;
  }
  if (tid < 64) S.B1[Lp + tid] = 0.f;

  // ------------------------------------------------- phase 1: raw-trace stats
  const float pv_bl = wv(P.bl.from);  // pivot of the baseline sums: the window's first sample (uniform scalar load)
  {
    float rmax = -INFINITY, rmin = INFINITY;
    WinAccF bl = {0, 0, 0};
    const float fic = (float)P.bl.ic;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i0 = 4 * (tid + NT * r);
      const float xi0 = (float)i0 - fic;
      if (FULL || i0 + 3 < L) {
        rmax = vmax3(rmax, x[r][0], x[r][1]); rmax = vmax3(rmax, x[r][2], x[r][3]);
        rmin = vmin3(rmin, x[r][0], x[r][1]); rmin = vmin3(rmin, x[r][2], x[r][3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = i0 + e < L;
          rmax = vmax(rmax, ok ? x[r][e] : -INFINITY);
          rmin = vmin(rmin, ok ? x[r][e] : INFINITY);
        }
      }
      winf_accum4(bl, P.bl, i0, xi0, pv_bl, x[r][0], x[r][1], x[r][2], x[r][3]);
    }
    STAMP(1);   // trace loaded, raw sums done
    win_publish<NW>(bl, S.wsum, 0);
    rmax = wave_max_all(rmax); rmin = wave_min_all(rmin);
    __syncthreads();  // slots initialised
    if (lane == 0) {
      atomicMax(&S.sl->fmx[FX_RAW], ford(rmax));
      atomicMin(&S.sl->fmn[FN_RAW], ford(rmin));
    }
  }
  __syncthreads();
  STAMP(2);
  // every thread needs the mean; sigma / slope / offset (double divisions, sqrt) only thread 0
  float blmean_ = (float)((double)pv_bl + win_collect1<NW>(S.wsum, 0) * P.bl.inv_n);
  if (ext_bl) blmean_ = ext_bl[blockIdx.x] * ext_bl_scale;   // windowed traces of dsp_icpc_compressed (dsp_icpc.jl:353)
  const float blmean = blmean_;
  if (tid == 0) {
    float m_, blsigma, blslope, bloffset;
    win_finish(win_collect<NW>(S.wsum, 0), P.bl, pv_bl, P.t_first, P.dt, &m_, &blsigma, &blslope, &bloffset);
    S.outv[C_blmean] = blmean; S.outv[C_blsigma] = blsigma; S.outv[C_blslope] = blslope; S.outv[C_bloffset] = bloffset;
  }
  const float raw_max = ford_inv(S.sl->fmx[FX_RAW]), raw_min = ford_inv(S.sl->fmn[FN_RAW]);
  const float e_max = raw_max - blmean;
  put(C_e_max, e_max); put(C_e_min, raw_min - blmean);

  // saturation runs (reference src/saturation.jl:28-65): rare path, only when a
  // saturated sample exists.  Flags through LDS -> ballots -> thread 0/1 walk the words.
  // A sample can only sit on a rail if the trace's extremes reach it: count (and look
  // for runs) only then.
  int n_low = 0, n_high = 0, cons_low = 0, cons_high = 0;
  if (raw_min <= P.sat_low || raw_max >= P.sat_high) {  // block-uniform
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = FULL || 4 * (tid + NT * r) + e < L;
        n_low += (ok && x[r][e] == P.sat_low);
        n_high += (ok && x[r][e] == P.sat_high);
      }
    n_low = wave_sum_all_i(n_low); n_high = wave_sum_all_i(n_high);
    if (lane == 0) {
      if (n_low) atomicAdd(&S.sl->isum[IS_LOW], n_low);
      if (n_high) atomicAdd(&S.sl->isum[IS_HIGH], n_high);
    }
    __syncthreads();
    n_low = S.sl->isum[IS_LOW]; n_high = S.sl->isum[IS_HIGH];
  }
  if (n_low > 0 || n_high > 0) {  // block-uniform
#pragma unroll
    for (int r = 0; r < R; ++r)
      *reinterpret_cast<float4*>(&S.B0[4 * (tid + NT * r)]) = make_float4(x[r][0], x[r][1], x[r][2], x[r][3]);
    __syncthreads();
    for (int m = 0; m < SP; ++m) {
      const int k = tid + NT * m;
      const float v = S.B0[k];
      ballot_store(k < L && v == P.sat_low, S.bm, (NT >> 5) * m + 2 * wave);
      ballot_store(k < L && v == P.sat_high, S.bm + NWORDS, (NT >> 5) * m + 2 * wave);
    }
    __syncthreads();
    if (tid < 2) {
      const uint32_t* b = S.bm + tid * NWORDS;
      int best = 0, run = 0;
      for (int wd = 0; wd < NWORDS; ++wd) {
        uint32_t v = b[wd];
        if (v == 0xffffffffu) { run += 32; continue; }
        if (v == 0) { best = max(best, run); run = 0; continue; }
        for (int bb = 0; bb < 32; ++bb) {
          if ((v >> bb) & 1u) ++run;
          else { best = max(best, run); run = 0; }
        }
      }
      S.misc[tid] = __int_as_float(max(best, run));
    }
    __syncthreads();
    cons_low = __float_as_int(S.misc[0]);
    cons_high = __float_as_int(S.misc[1]);
    __syncthreads();
  }
  puti(C_n_sat_low, n_low); puti(C_n_sat_high, n_high);
  puti(C_n_sat_low_cons, cons_low); puti(C_n_sat_high_cons, cons_high);
  if (P.dbg_stop == 1) return;

  STAMP(3);
  // shift_waveform(-blmean) (dsp_icpc.jl:105); tailstats on the shifted trace
  // (src/tailstats.jl:22-72); cumsum for the pole-zero correction
  double s_off[R];
  const float pv_tl = __logf(fmaxf(wv(P.tail.from) - blmean, 1e-30f));  // pivot of the log sums: the window's first sample
  {
    WinAccF tl = {0, 0, 0};
    int tail_bad = 0;
    float tot[R];
    const float ficl = (float)P.tail.ic;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i0 = 4 * (tid + NT * r);
      // wave-row [first, last] vs the tail window: skip the logs when disjoint (wave-uniform)
      const int wfirst = 4 * ((tid & ~63) + NT * r), wlast = wfirst + 255;
      const bool wave_in = (wfirst <= P.tail.until) && (wlast >= P.tail.from);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[r][e] = (i0 + e < L) ? x[r][e] - blmean : 0.f;
      if (wave_in && i0 + 3 >= P.tail.from && i0 <= P.tail.until) {
        float lv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = i0 + e;
          const float v = x[r][e];
          if (i >= P.tail.from && i <= P.tail.until && v <= 0.f) tail_bad = 1;
          lv[e] = __logf(fmaxf(v, 1e-30f));
        }
        winf_accum4(tl, P.tail, i0, (float)i0 - ficl, pv_tl, lv[0], lv[1], lv[2], lv[3]);
      }
      tot[r] = (x[r][0] + x[r][1]) + (x[r][2] + x[r][3]);
    }
    win_publish<NW>(tl, S.wsum, 3);
    if (__ballot(tail_bad) != 0ull && lane == 0) atomicAdd(&S.sl->isum[IS_TAILBAD], 1);
    s4_exscan_sum<NT, R>(tot, s_off, part_buf(), nullptr);  // barrier inside: tail sums published too
  }
  if (tid == 0) {
    float tail_mean = 0.f, tail_sigma = 0.f, tail_tau = 0.f;
    if (S.sl->isum[IS_TAILBAD] == 0) {
      float sl, of;
      win_finish(win_collect<NW>(S.wsum, 3), P.tail, pv_tl, P.t_first, P.dt, &tail_mean, &tail_sigma, &sl, &of);
      tail_tau = -1.f / sl;
    }
    S.outv[C_tail_tau] = tail_tau; S.outv[C_tail_mean] = tail_mean; S.outv[C_tail_sigma] = tail_sigma;
  }
  STAMP(4);
  // InvCRFilter: y = x + c*cumsum(x)  (dsp_icpc.jl:119-120);  x[][] becomes y
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i0 = 4 * (tid + NT * r);
    const float coff = (float)(P.pz_c64 * s_off[r]);
    float run = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      run += x[r][e];
      x[r][e] = (i0 + e < L) ? (x[r][e] + coff) + P.pz_c * run : 0.f;
    }
  }
  auto& y = x;
  if (P.dbg_stop == 7) return;
  // get_threshold at 10 / 50 / 80 / 90 / 99 % of the pre-PZ maximum (dsp_icpc.jl:132-136): Intersect needs the FIRST up-crossing
  // that holds for tx_mintot samples, and on a pulse that is the first sample at or above the threshold.  Each thread
  // finds that sample among its own 16 (S4 view, registers: a compare + select per sample and threshold, no ballots, no
  // mask words), the workgroup takes the minimum; the candidate is confirmed further down (the samples after it, and that
  // it is not sample 0) and only a trace that fails the confirmation runs the general bit-mask scan.
  const float thr_tx[5] = {e_max * 0.1f, e_max * 0.5f, e_max * 0.8f, e_max * 0.9f, e_max * 0.99f};
  if (e_max > 0.f) {   // thresholds ascend; otherwise the general scan handles the trace
    auto first_code = [&](float thr) {   // 4r+e of the thread's first sample >= thr, 16 = none  (y is 0 beyond L, thr > 0)
      int code = 16;
#pragma unroll
      for (int r = R - 1; r >= 0; --r)
#pragma unroll
        for (int e = 3; e >= 0; --e) code = (y[r][e] < thr) ? code : (4 * r + e);
      return code;
    };
    int code[5];
    code[0] = first_code(thr_tx[0]);
    code[4] = first_code(thr_tx[4]);
    if (__ballot(code[0] != code[4]) == 0ull) {   // wave-uniform: no sample of this wave lies between the lowest and the highest threshold
      code[1] = code[2] = code[3] = code[0];
    } else {
      code[1] = first_code(thr_tx[1]); code[2] = first_code(thr_tx[2]); code[3] = first_code(thr_tx[3]);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const uint32_t key = (code[q] == 16) ? 0x7fffffffu : (uint32_t)(4 * (tid + NT * (code[q] >> 2)) + (code[q] & 3));
      const uint32_t kmin = wave_min_u32(key);
      if (lane == 0 && kmin != 0x7fffffffu) atomicMin(&S.sl->imin[q], (int)kmin);
    }
  }

  // ----------------------------------- phase 2: SG derivatives, current maxima
  // (runs before the prefix sum T is built: its full-length output is parked in B1, which T then
  // takes over and keeps until the CUSP/ZAC stage turns it into Dp in place)
  // LS view: g[k] = sum_i c[i] y[k+i] (valid mode, trailing time axis).  The SG(sg_wl)
  // output is needed in full (pile-up scan, t50_current) and is parked in B1; SG(60ns), SG(100ns) and the plain derivative only in the current window.
  const int ng = L - P.sg_npts[0] + 1;
  auto flt_at = [&](int f, int k) -> float { return flt_eval(S.B0, P.sg_c[f < 3 ? f : 0], P.sg_npts[f < 3 ? f : 0], f, k); };
  auto flt_rare = [&](int f, int k) -> float { return flt_eval_rare(S.B0, P.sg_c[f < 3 ? f : 0], P.sg_npts[f < 3 ? f : 0], f, k); };
  static_assert(M_INTR == M_SG50 + 1, "mask order");
  for (int i = tid; i < 2 * NWORDS / 4; i += NT) reinterpret_cast<uint4*>(S.bm + M_SG50 * NWORDS)[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int r = 0; r < R; ++r)
    *reinterpret_cast<float4*>(&S.B0[4 * (tid + NT * r)]) = make_float4(y[r][0], y[r][1], y[r][2], y[r][3]);
  __syncthreads();  // B0 = y visible to every wave (halo and LS reads below)
  STAMP(5);
  {
    float gmax = -INFINITY;
    WinAccF sgb = {0, 0, 0};   // pivot 0: the SG derivative of a baseline has no level
    const float ficg = (float)P.sgbl.ic;
    float bv[4]; int bi[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { bv[f] = -INFINITY; bi[f] = 0x7fffffff; }
    // S4 evaluation from the register-resident y plus a halo of the next samples (ds_read_b128):
    // all four filters share one window; taps in SGPRs, zero-padded to M.
    auto sg_pass_s4 = [&](auto mtag) {
      constexpr int M = decltype(mtag)::value;
      constexpr int NH = (M - 1 + 3) / 4;
      float c0[M], c1[M], c2[M];
#pragma unroll
      for (int i = 0; i < M; ++i) {
        c0[i] = (i < P.sg_npts[0]) ? P.sg_c[0][i] : 0.f;
        c1[i] = (i < P.sg_npts[1]) ? P.sg_c[1][i] : 0.f;
        c2[i] = (i < P.sg_npts[2]) ? P.sg_c[2][i] : 0.f;
      }
      const int wlo = min(min(P.cur_from[1], P.cur_from[2]), P.cur_from[3]);
      const int whi = max(max(P.cur_until[1], P.cur_until[2]), P.cur_until[3]);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i0 = 4 * (tid + NT * r);
        float win[4 + 4 * NH];
#pragma unroll
        for (int e = 0; e < 4; ++e) win[e] = y[r][e];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const float4 v = *reinterpret_cast<const float4*>(&S.B0[i0 + 4 + 4 * h]);  // past Lp this runs into B1: masked below
          win[4 + 4 * h] = v.x; win[5 + 4 * h] = v.y; win[6 + 4 * h] = v.z; win[7 + 4 * h] = v.w;
        }
        const int wfirst = 4 * ((tid & ~63) + NT * r), wlast = wfirst + 255;  // this wave-row's sample range
        float go[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float g = 0.f;
#pragma unroll
          for (int i = 0; i < M; ++i) g = fmaf(c0[i], win[e + i], g);
          go[e] = g;
        }
        if (wlast >= ng) {   // only the wave-row holding the end of the output axis
#pragma unroll
          for (int e = 0; e < 4; ++e) go[e] = (i0 + e < ng) ? go[e] : -INFINITY;
        }
        gmax = vmax3(vmax3(gmax, go[0], go[1]), go[2], go[3]);
        if (wfirst <= P.cur_until[0] && wlast >= P.cur_from[0]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = i0 + e;
            if (k >= P.cur_from[0] && k <= P.cur_until[0] && go[e] > bv[0]) { bv[0] = go[e]; bi[0] = k; }
          }
        }
        *reinterpret_cast<float4*>(&S.B1[i0]) = make_float4(go[0], go[1], go[2], go[3]);
        winf_accum4(sgb, P.sgbl, i0, (float)i0 - ficg, 0.f, go[0], go[1], go[2], go[3]);  // sgbl.until <= ng-1: -inf never enters
        // SG(60 ns), SG(100 ns), plain derivative: only wave-rows that touch the current window
        if (wfirst <= whi && wlast >= wlo) {
          const float ypv = (i0 > 0) ? S.B0[i0 - 1] : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = i0 + e;
            float g1 = 0.f, g2 = 0.f;
#pragma unroll
            for (int i = 0; i < M; ++i) { g1 = fmaf(c1[i], win[e + i], g1); g2 = fmaf(c2[i], win[e + i], g2); }
            float g3 = win[e] - (e > 0 ? win[(e + 3) & 3] : ypv);  // y[k] - y[k-1]
            if (k == 0) g3 = win[1] - win[0];                        // y[max(i,1)] - y[max(i-1,0)] at i = 0
            if (k >= P.cur_from[1] && k <= P.cur_until[1] && g1 > bv[1]) { bv[1] = g1; bi[1] = k; }
            if (!P.sg_same_02 && k >= P.cur_from[2] && k <= P.cur_until[2] && g2 > bv[2]) { bv[2] = g2; bi[2] = k; }
            if (k >= P.cur_from[3] && k <= P.cur_until[3] && g3 > bv[3]) { bv[3] = g3; bi[3] = k; }
          }
        }
      }
    };
    const int npmax = max(P.sg_npts[0], max(P.sg_npts[1], P.sg_npts[2]));
    if (npmax <= 7) {
      sg_pass_s4(std::integral_constant<int, 7>{});
    } else if (npmax <= 13) {
      sg_pass_s4(std::integral_constant<int, 13>{});
    } else {
      // generic LS evaluation for long windows
      for (int m = 0; m < SP; ++m) {
        const int k = tid + NT * m;
        float g0 = -INFINITY;
        if (k < ng) {
          g0 = flt_at(0, k);
          gmax = fmaxf(gmax, g0);
          winf_accum(sgb, P.sgbl, k, (float)k - ficg, 0.f, g0);
          if (k >= P.cur_from[0] && k <= P.cur_until[0] && g0 > bv[0]) { bv[0] = g0; bi[0] = k; }
        }
        S.B1[k] = g0;
  #pragma unroll
        for (int f = 1; f < 4; ++f) {
          if (f == 2 && P.sg_same_02) continue;
          if (k >= P.cur_from[f] && k <= P.cur_until[f]) {
            const float g = flt_at(f, k);
            if (g > bv[f]) { bv[f] = g; bi[f] = k; }
          }
        }
      }
    }
    STAMP(6);   // SG pass done
    win_publish<NW>(sgb, S.wsum, 9);
    gmax = wave_max_all(gmax);
    if (lane == 0) atomicMax(&S.sl->fmx[FX_G], ford(gmax));
    // only the waves whose samples touch a current window hold a candidate (wave-uniform test)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (__ballot(bi[f] != 0x7fffffff) != 0ull) {
        const unsigned long long bk = wave_max_u64(pack_vi(bv[f], bi[f]));
        if (lane == 0) atomicMax(&S.sl->vi[VI_CUR0 + f], bk);
      }
    }
  }
  __syncthreads();
  STAMP(7);
  // get_wvf_maximum (src/interpolation.jl:30-46): parabola through the three samples about the
  // maximum if it is strictly interior.  Wave 0 only; lane 3f+d+1 evaluates filter f at i_f+d.
  if (wave == 0) {
    const int f = min(lane / 3, 3), d = lane - 3 * f - 1;
    const int fs = (f == 2 && P.sg_same_02) ? 0 : f;
    float v; int i;
    unpack_vi(S.sl->vi[VI_CUR0 + fs], &v, &i);
    const bool interior = i > P.cur_from[fs] && i < P.cur_until[fs];
    float ev = 0.f;
    if (lane < 12 && interior) ev = flt_rare(fs, i + d);
    const float em = __shfl(ev, 3 * f), e0 = __shfl(ev, 3 * f + 1), ep = __shfl(ev, 3 * f + 2);
    if (interior) v = extrema3points(em, e0, ep);
    if (lane < 12 && d == -1) S.outv[f == 0 ? C_a_sg : f == 1 ? C_a_60 : f == 2 ? C_a_100 : C_a_raw] = v;
  }
  // in-trace pile-up threshold (dsp_routines.jl:75-77) and t50_current threshold (dsp_icpc.jl:192)
  float thr_intr, thr_sg50;
  {
    // sigma of the SG output over the baseline window: only the first two sums
    const double m_ = win_collect1<NW>(S.wsum, 9) * P.sgbl.inv_n;
    double var_ = win_collect1<NW>(S.wsum, 10) * P.sgbl.inv_n - m_ * m_;
    if (var_ < 0) var_ = 0;
    const float sg_ = sqrtf((float)var_);
    thr_intr = sg_ * P.intrace_nsigma;
    if (thr_intr == 0.f) thr_intr = 1.f;
    thr_sg50 = ford_inv(S.sl->fmx[FX_G]) * 0.5f;
  }
  // t50_current (dsp_icpc.jl:192-195) is the first up-crossing of thr_sg50 held for tx_mintot samples.  With tx_mintot = 2
  // (the usual 32 ns at 16 ns sampling) a crossing at k is  g[k-1] < thr <= min(g[k], g[k+1]):  three reads per sample and a
  // running select give each thread its first crossing, no ballot and no mask words; other values of tx_mintot take the
  // bit-mask scan.  The in-trace pile-up mask (a count of runs is needed there) stays a ballot per row.
  const bool sg50_direct = P.tx_mintot == 2;   // block-uniform
  int code50 = SP;
#pragma unroll
  for (int m = SP - 1; m >= 0; --m) {
    const float g = S.B1[tid + NT * m];  // LS view of the SG output (-inf beyond its last sample)
    if (sg50_direct) {
      float gm = S.B1[max(tid + NT * m - 1, 0)];
      const float gp = S.B1[tid + NT * m + 1];
      if (m == 0) gm = (tid == 0) ? INFINITY : gm;            // a run that starts the trace is no crossing
      const float w = (gm < thr_sg50) ? vmin(g, gp) : -INFINITY;
      code50 = (w >= thr_sg50) ? m : code50;
    }
    const unsigned long long bin = __ballot(g >= thr_intr);
    unsigned long long b50 = 0ull;
    if (!sg50_direct) b50 = __ballot(g >= thr_sg50);
    if (lane == 0) {  // words were zeroed before the SG pass: only non-zero ballots are stored
      const int wb = (NT >> 5) * m + 2 * wave;
      if (b50) *reinterpret_cast<unsigned long long*>(&S.bm[M_SG50 * NWORDS + wb]) = b50;
      if (bin) *reinterpret_cast<unsigned long long*>(&S.bm[M_INTR * NWORDS + wb]) = bin;
    }
  }
  if (sg50_direct) {
    const uint32_t kmin = wave_min_u32(code50 == SP ? 0x7fffffffu : (uint32_t)(tid + NT * code50));
    if (lane == 0 && kmin != 0x7fffffffu) atomicMin(&S.sl->imin[M_SG50], (int)kmin);
  }
  STAMP(8);
  // (the run scans of these two masks and the crossing interpolations follow in phase 4b, together with
  //  the seven masks of the sweep: no barrier here — the next one is inside the prefix-sum scan, which
  //  also orders these B1 reads before T overwrites B1)
  if (P.dbg_stop == 2) return;

  // ------------------------------------------------- phase 3: T = prefix sum of y
  float pv_pz;
  {
    float tot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i0 = 4 * (tid + NT * r);
      tot[r] = (y[r][0] + y[r][1]) + (y[r][2] + y[r][3]);
    }
    double t_off[R], tot_all;
    s4_exscan_sum<NT, R>(tot, t_off, part_buf(), &tot_all);   // barrier inside: the SG stage's reads of B1 are done
    // signalstats of the pole-zero corrected tail (dsp_icpc.jl:122), pivot = its first sample
    WinAccF pz = {0, 0, 0};
    pv_pz = S.B0[P.tail.from];
    const float ficp = (float)P.tail.ic;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i0 = 4 * (tid + NT * r);
      winf_accum4(pz, P.tail, i0, (float)i0 - ficp, pv_pz, y[r][0], y[r][1], y[r][2], y[r][3]);
      double run = t_off[r];
      float4 t;
      float* pt = &t.x;
#pragma unroll
      for (int e = 0; e < 4; ++e) { pt[e] = (float)run; run += (double)y[r][e]; }
      *reinterpret_cast<float4*>(&S.B1[i0]) = t;
    }
    win_publish<NW>(pz, S.wsum, 6);
    if (tid == 0) S.B1[Lp] = (float)tot_all;  // T[Lp] (= T[L] when L == Lp; y is 0 beyond L)
    static_assert(M_T0INV == M_T0 + 1, "mask order");
    for (int i = tid; i < 2 * NWORDS / 4; i += NT)  // mask words of the t0 trapezoid: only non-zero ballots are stored
      reinterpret_cast<uint4*>(S.bm + M_T0 * NWORDS)[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  STAMP(9);   // T in LDS
  if (tid == 0) {
    float tailmean, tailsigma, tailslope, tailoffset;
    win_finish(win_collect<NW>(S.wsum, 6), P.tail, pv_pz, P.t_first, P.dt, &tailmean, &tailsigma, &tailslope, &tailoffset);
    S.outv[C_tailmean] = tailmean; S.outv[C_tailsigma] = tailsigma; S.outv[C_tailslope] = tailslope; S.outv[C_tailoffset] = tailoffset;
  }
  if (P.dbg_stop == 3) return;

  // ------------------------------------------------ phase 4: lane-strided sweep
  {
    float mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY, mn0 = INFINITY, mn2 = INFINITY;
    float bo_v = -INFINITY; int bo_i = 0x7fffffff;
    const int nout_t0 = L - P.t0.flen + 1, nout_t0i = L - P.t0inv.flen + 1;
    const int nout_f0 = L - P.fixed[0].flen + 1, nout_f1 = L - P.fixed[1].flen + 1,
              nout_f2 = L - P.fixed[2].flen + 1, nout_opt = L - P.opt.flen + 1;
    const TrapDev t0 = P.t0, t0i = P.t0inv, f0 = P.fixed[0], f1 = P.fixed[1], f2 = P.fixed[2], fo = P.opt;
    const bool inv_same = P.t0inv_same != 0;
    // One base register per shifted read; the row offset NT*m is an immediate.  A trapezoid is
    // evaluated unscaled, o' = (T[k+flen]-T[k+n1+g])*(inv2/inv1) - (T[k+n1]-T[k]) (two subtractions and
    // one fma), thresholds are divided by inv1 and maxima multiplied by it after the sweep.
    // Rows wholly inside an output range skip the per-lane range test (scalar branch on the row).
    const float* tb = &S.B1[tid];
    auto traw = [&](const float* a, const float* b, const float* c, float rr, float Tk, int m) {
      return fmaf(c[NT * m] - b[NT * m], rr, -(a[NT * m] - Tk));
    };
    // ---- sweep A: threshold bit-masks of the t0 trapezoid (t0, inverted t0).  (The five thresholds on y are found from
    // registers, see the candidate search after the pole-zero stage.)
    {
      const float *t0a = tb + t0.n1, *t0b = tb + t0.n1 + t0.g, *t0c = tb + t0.flen;
      const float *tia = tb + t0i.n1, *tib = tb + t0i.n1 + t0i.g, *tic = tb + t0i.flen;
      const float rr0 = t0.rr, rri = t0i.rr;
      const float thr0 = P.t0_thr * t0.navg, thr0i = -P.t0_thr * t0i.navg;
      // ballots of one row and the stores of the non-zero ones (the mask words were zeroed in phase 3)
      auto emit = [&](int m, float o0, float o0i) {
        const unsigned long long b0 = __ballot(o0 >= thr0);
        const unsigned long long bi = __ballot(o0i <= thr0i);   // -trap >= thr
        if (lane == 0) {
          const int wb = (NT >> 5) * m + 2 * wave;
          if (b0) *reinterpret_cast<unsigned long long*>(&S.bm[M_T0 * NWORDS + wb]) = b0;
          if (bi) *reinterpret_cast<unsigned long long*>(&S.bm[M_T0INV * NWORDS + wb]) = bi;
        }
      };
      // A first leg of <= 3 samples (get_t0's 40 ns) is summed from y itself (B0): on the tail T is 1e7..1e8 and a difference
      // of two of its float values is good to 1..8 counts — the size of the threshold the INVERTED trace is tested against there.
      const float* ya = &S.B0[tid];
      auto leg1 = [&](int n1, const float* a, float Tk, int m) {   // sum of the first leg's n1 samples of output row m
        if (n1 <= 3) {
          float l = ya[NT * m];
          if (n1 >= 2) l += ya[NT * m + 1];
          if (n1 >= 3) l += ya[NT * m + 2];
          return l;
        }
        return a[NT * m] - Tk;
      };
      auto traw1 = [&](int n1, const float* a, const float* b, const float* c, float rr, float Tk, int m) {
        return fmaf(c[NT * m] - b[NT * m], rr, -leg1(n1, a, Tk, m));
      };
      static_assert(SP % 4 == 0, "row groups");
      const int nout_grp = inv_same ? nout_t0 : min(nout_t0, nout_t0i);
#pragma unroll
      for (int m0 = 0; m0 < SP; m0 += 4) {
        if (NT * (m0 + 4) <= nout_grp) {
          // four rows wholly inside the trace and both output ranges: all their reads go out before the first comparison
          // (compiler fence), the LDS latency is paid once per group
          float tk[4], ra[4], rb[4], rc[4], ia[4], ib[4], ic[4];   // ra / ia: the first leg's sum
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            tk[q] = (t0.n1 <= 3 && (inv_same || t0i.n1 <= 3)) ? 0.f : tb[NT * (m0 + q)];
            ra[q] = leg1(t0.n1, t0a, tk[q], m0 + q); rb[q] = t0b[NT * (m0 + q)]; rc[q] = t0c[NT * (m0 + q)];
            if (!inv_same) { ia[q] = leg1(t0i.n1, tia, tk[q], m0 + q); ib[q] = tib[NT * (m0 + q)]; ic[q] = tic[NT * (m0 + q)]; }
          }
          asm volatile("" ::: "memory");
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float o0 = fmaf(rc[q] - rb[q], rr0, -ra[q]);
            const float o0i = inv_same ? o0 : fmaf(ic[q] - ib[q], rri, -ia[q]);
            emit(m0 + q, o0, o0i);
          }
          continue;
        }
#pragma unroll
        for (int m = m0; m < m0 + 4; ++m) {
          const int k = tid + NT * m;
          const float Tk = tb[NT * m];
          float o0 = NAN, o0i = NAN;   // NaN: both comparisons false for rows/lanes outside the output range
          if (NT * (m + 1) <= nout_t0) o0 = traw1(t0.n1, t0a, t0b, t0c, rr0, Tk, m);
          else if (NT * m < nout_t0) { o0 = traw1(t0.n1, t0a, t0b, t0c, rr0, Tk, m); o0 = (k < nout_t0) ? o0 : NAN; }
          if (inv_same) o0i = o0;
          else if (NT * (m + 1) <= nout_t0i) o0i = traw1(t0i.n1, tia, tib, tic, rri, Tk, m);
          else if (NT * m < nout_t0i) { o0i = traw1(t0i.n1, tia, tib, tic, rri, Tk, m); o0i = (k < nout_t0i) ? o0i : NAN; }
          emit(m, o0, o0i);
        }
      }
    }
    STAMP(10);  // sweep A done
    // ---- sweep B: extrema of the three fixed trapezoids and the arg-max of the optimised one
    {
      const float *f0a = tb + f0.n1, *f0b = tb + f0.n1 + f0.g, *f0c = tb + f0.flen;
      const float *f1a = tb + f1.n1, *f1b = tb + f1.n1 + f1.g, *f1c = tb + f1.flen;
      const float *f2a = tb + f2.n1, *f2b = tb + f2.n1 + f2.g, *f2c = tb + f2.flen;
      const float *foa = tb + fo.n1, *fob = tb + fo.n1 + fo.g, *foc = tb + fo.flen;
      const float rr0 = f0.rr, rr1 = f1.rr, rr2 = f2.rr, rro = fo.rr;
      // Two rows per step: a pair's reads share their base registers (ds_read2st64_b32), its
      // arithmetic packs (v_pk_add/fma) and one v_max3 / v_min3 folds both rows.  Pairs wholly
      // inside an output range take the unmasked branch (scalar test); the one straddling pair
      // of each trapezoid runs row by row with lane masks.
      static_assert(SP % 2 == 0, "row pairs");
      const int nout_all = min(min(nout_f0, nout_f1), min(nout_f2, nout_opt));
#pragma unroll
      for (int m = 0; m < SP; m += 2) {
        if (NT * (m + 2) <= nout_all) {
          // pair wholly inside every output range (most pairs): ALL 26 reads are issued before the arithmetic starts
          // (compiler fence), so the LDS latency is paid once per pair instead of once per read or two
          float a[4][2], b[4][2], c[4][2], t[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            t[q] = tb[NT * (m + q)];
            a[0][q] = f0a[NT * (m + q)]; b[0][q] = f0b[NT * (m + q)]; c[0][q] = f0c[NT * (m + q)];
            a[1][q] = f1a[NT * (m + q)]; b[1][q] = f1b[NT * (m + q)]; c[1][q] = f1c[NT * (m + q)];
            a[2][q] = f2a[NT * (m + q)]; b[2][q] = f2b[NT * (m + q)]; c[2][q] = f2c[NT * (m + q)];
            a[3][q] = foa[NT * (m + q)]; b[3][q] = fob[NT * (m + q)]; c[3][q] = foc[NT * (m + q)];
          }
          asm volatile("" ::: "memory");
          auto tv = [&](int j, float rr, int q) { return fmaf(c[j][q] - b[j][q], rr, -(a[j][q] - t[q])); };
          const float o00 = tv(0, rr0, 0), o01 = tv(0, rr0, 1);
          mx0 = vmax3(mx0, o00, o01); mn0 = vmin3(mn0, o00, o01);
          mx1 = vmax3(mx1, tv(1, rr1, 0), tv(1, rr1, 1));
          const float o20 = tv(2, rr2, 0), o21 = tv(2, rr2, 1);
          mx2 = vmax3(mx2, o20, o21); mn2 = vmin3(mn2, o20, o21);
          const float oo0 = tv(3, rro, 0), oo1 = tv(3, rro, 1);
          if (oo0 > bo_v) { bo_v = oo0; bo_i = tid + NT * m; }
          if (oo1 > bo_v) { bo_v = oo1; bo_i = tid + NT * (m + 1); }
          continue;
        }
        const float Tk0 = tb[NT * m], Tk1 = tb[NT * (m + 1)];
        auto row = [&](int mm) { return mm == m ? Tk0 : Tk1; };
        if (NT * (m + 2) <= nout_f0) {
          const float o0 = traw(f0a, f0b, f0c, rr0, Tk0, m), o1 = traw(f0a, f0b, f0c, rr0, Tk1, m + 1);
          mx0 = vmax3(mx0, o0, o1); mn0 = vmin3(mn0, o0, o1);
        } else {
#pragma unroll
          for (int mm = m; mm < m + 2; ++mm)
            if (NT * mm < nout_f0) {
              const float o = traw(f0a, f0b, f0c, rr0, row(mm), mm);
              const bool in = tid + NT * mm < nout_f0;
              mx0 = vmax(mx0, in ? o : -INFINITY); mn0 = vmin(mn0, in ? o : INFINITY);
            }
        }
        if (NT * (m + 2) <= nout_f1) {
          mx1 = vmax3(mx1, traw(f1a, f1b, f1c, rr1, Tk0, m), traw(f1a, f1b, f1c, rr1, Tk1, m + 1));
        } else {
#pragma unroll
          for (int mm = m; mm < m + 2; ++mm)
            if (NT * mm < nout_f1) {
              const float o = traw(f1a, f1b, f1c, rr1, row(mm), mm);
              mx1 = vmax(mx1, (tid + NT * mm < nout_f1) ? o : -INFINITY);
            }
        }
        if (NT * (m + 2) <= nout_f2) {
          const float o0 = traw(f2a, f2b, f2c, rr2, Tk0, m), o1 = traw(f2a, f2b, f2c, rr2, Tk1, m + 1);
          mx2 = vmax3(mx2, o0, o1); mn2 = vmin3(mn2, o0, o1);
        } else {
#pragma unroll
          for (int mm = m; mm < m + 2; ++mm)
            if (NT * mm < nout_f2) {
              const float o = traw(f2a, f2b, f2c, rr2, row(mm), mm);
              const bool in = tid + NT * mm < nout_f2;
              mx2 = vmax(mx2, in ? o : -INFINITY); mn2 = vmin(mn2, in ? o : INFINITY);
            }
        }
        if (NT * (m + 2) <= nout_opt) {
          const float o0 = traw(foa, fob, foc, rro, Tk0, m), o1 = traw(foa, fob, foc, rro, Tk1, m + 1);
          if (o0 > bo_v) { bo_v = o0; bo_i = tid + NT * m; }
          if (o1 > bo_v) { bo_v = o1; bo_i = tid + NT * (m + 1); }
        } else {
#pragma unroll
          for (int mm = m; mm < m + 2; ++mm)
            if (NT * mm < nout_opt) {
              const float o = traw(foa, fob, foc, rro, row(mm), mm);
              if (tid + NT * mm < nout_opt && o > bo_v) { bo_v = o; bo_i = tid + NT * mm; }
            }
        }
      }
      mx0 *= f0.inv1; mn0 *= f0.inv1; mx1 *= f1.inv1; mx2 *= f2.inv1; mn2 *= f2.inv1; bo_v *= fo.inv1;
    }
    STAMP(11);  // sweep B done
    mx0 = wave_max_all(mx0); mx1 = wave_max_all(mx1); mx2 = wave_max_all(mx2);
    mn0 = wave_min_all(mn0); mn2 = wave_min_all(mn2);
    unsigned long long bo = wave_max_u64(pack_vi(bo_v, bo_i));
    if (lane == 0) {
      atomicMax(&S.sl->fmx[FX_F0], ford(mx0));
      atomicMax(&S.sl->fmx[FX_F1], ford(mx1));
      atomicMax(&S.sl->fmx[FX_F2], ford(mx2));
      // max(trap(-y)) = -min(trap(y)): negate BEFORE the order map (negating the decoded
      // slot value was folded into a wrong sign by hipcc 7.2)
      atomicMax(&S.sl->fmx[FX_F0I], ford(-mn0));
      atomicMax(&S.sl->fmx[FX_F2I], ford(-mn2));
      atomicMax(&S.sl->vi[VI_OPT], bo);
    }
  }
  __syncthreads();
  if (P.dbg_stop == 4) return;
  STAMP(12);
  // Intersect scans on the bit-masks (thread w <-> word w): the two masks of the sweep and the two
  // of the SG stage (phase 2)
  for (int j = tid; j < 2 * NWORDS; j += NT) {
    const int q = M_T0 + j / NWORDS, wd = j % NWORDS;
    int c, f;
    intersect_word(S.bm + q * NWORDS, wd, NWORDS, P.t0_mintot, &c, &f);
    if (c) { atomicAdd(&S.sl->isum[IS_CNT0 + q], c); atomicMin(&S.sl->imin[q], f); }
  }
  for (int wd = tid; wd < NWORDS; wd += NT) {
    int c, f;
    if (P.tx_mintot != 2) {   // otherwise found without a mask (phase 2)
      intersect_word(S.bm + M_SG50 * NWORDS, wd, NWORDS, P.tx_mintot, &c, &f);
      if (c) atomicMin(&S.sl->imin[M_SG50], f);
    }
    intersect_word_rev(S.bm + M_INTR * NWORDS, wd, NWORDS, ng, P.intrace_mintot, &c, &f);
    if (c) { atomicAdd(&S.sl->isum[IS_CNT0 + M_INTR], c); atomicMax(&S.sl->imax[0], f); }
  }
  __syncthreads();
  STAMP(13);
  // crossing interpolations: wave 0, lanes 0..3 evaluate the four SG samples they need
  if (wave == 0) {
    const int intr_n = S.sl->isum[IS_CNT0 + M_INTR];
    const int p = S.sl->imin[M_SG50], e = S.sl->imax[0];
    const bool has50 = p != 0x7fffffff;
    const int at = (lane == 0) ? p - 1 : (lane == 1) ? p : (lane == 2) ? e + 1 : e;
    float ev = 0.f;
    if (lane < 4 && ((lane < 2) ? has50 : intr_n > 0)) ev = flt_rare(0, at);
    const float yl5 = __shfl(ev, 0), yh5 = __shfl(ev, 1), yli = __shfl(ev, 2), yhi = __shfl(ev, 3);
    if (lane == 0) {
      const float tg_first = P.t_first + P.dt * (float)(P.sg_npts[0] - 1);  // trailing alignment (A1)
      float t50cur_us = 0.f, intr_x = NAN;
      if (has50) t50cur_us = (tg_first + P.dt * ((float)(p - 1) + (thr_sg50 - yl5) / (yh5 - yl5))) * P.inv_unit_per_us;
      if (intr_n > 0) {
        // reversed index pos' = ng-1-e ; r[pos'-1] = g[e+1], r[pos'] = g[e]
        const int pr = ng - 1 - e;
        const float xl = tg_first + P.dt * (float)(pr - 1);
        const float xr_ = (thr_intr - yli) * P.dt / (yhi - yli) + xl;
        intr_x = (tg_first + P.dt * (float)(ng - 1)) - xr_;  // last(time) - x   (dsp_routines.jl:81)
      }
      S.outv[C_t50_current] = t50cur_us; S.outv[C_inTrace_intersect] = intr_x; S.outv[C_inTrace_n] = __int_as_float(intr_n);
    }
  }
  {
    float mx_opt_v; int mx_opt_i;
    unpack_vi(S.sl->vi[VI_OPT], &mx_opt_v, &mx_opt_i);
    put(C_e_trap_max, mx_opt_v); put(C_t_trap_max, P.t_first + P.dt * (float)(mx_opt_i + P.opt.flen - 1));
    put(C_e_10410, ford_inv(S.sl->fmx[FX_F0])); put(C_e_535, ford_inv(S.sl->fmx[FX_F1])); put(C_e_313, ford_inv(S.sl->fmx[FX_F2]));
    // trap(-y) = -trap(y)  (dsp_icpc.jl:199-204)
    put(C_e_10410_inv, ford_inv(S.sl->fmx[FX_F0I])); put(C_e_313_inv, ford_inv(S.sl->fmx[FX_F2I]));
  }
  // Confirmation of the five threshold candidates (imin[0..4] = first sample at or above the threshold): a candidate is the
  // crossing Intersect reports if it is not sample 0 (a run that starts the trace is no crossing) and the tx_mintot - 1 samples
  // after it stay at or above the threshold.  Every wave checks all five (lane q <-> threshold q) and reaches the same
  // verdict; a trace that fails (e.g. a noise spike in front of the pulse) — or whose thresholds do not ascend — runs the
  // general scan: bit-masks of y by ballot, run scan on the words (reference scan: src/intersect_maximum.jl:41-56).
  {
    const int q = min(lane, 4);
    const int p = S.sl->imin[q];
    bool ok = e_max > 0.f;
    if (ok && p != 0x7fffffff) {
      const float thrq = e_max * ((q == 0) ? 0.1f : (q == 1) ? 0.5f : (q == 2) ? 0.8f : (q == 3) ? 0.9f : 0.99f);   // = thr_tx[q]
      ok = p >= 1 && p + P.tx_mintot <= L;
      for (int j = 1; ok && j < P.tx_mintot; ++j) ok = S.B0[p + j] >= thrq;
    }
    if (__ballot(lane < 5 && !ok) != 0ull) {   // block-uniform
      __syncthreads();                          // every wave has read the candidates
      if (tid < 5) S.sl->imin[tid] = 0x7fffffff;
      for (int m = 0; m < SP; ++m) {
        const int k = tid + NT * m;
        float yv = S.B0[k];
        if (!FULL) yv = (k < L) ? yv : -INFINITY;
#pragma unroll
        for (int qq = 0; qq < 5; ++qq) {
          const unsigned long long b = __ballot(yv >= thr_tx[qq]);
          if (lane == 0) *reinterpret_cast<unsigned long long*>(&S.bm[qq * NWORDS + (NT >> 5) * m + 2 * wave]) = b;
        }
      }
      __syncthreads();
      for (int j = tid; j < 5 * NWORDS; j += NT) {
        const int qq = j / NWORDS, wd = j % NWORDS;
        int c, f;
        intersect_word(S.bm + qq * NWORDS, wd, NWORDS, P.tx_mintot, &c, &f);
        if (c) atomicMin(&S.sl->imin[qq], f);
      }
      __syncthreads();
    }
  }
  // crossing positions (sample units, split int + frac); NaN -> 0 us (dsp_routines.jl:24,41).
  // Seven interpolations, one per lane (lane q < 5: threshold q of y; 5: t0; 6: inverted t0),
  // evaluated once per wave and handed out by readlane.
  Pos ptx[3];   // [1] = t50, [2] = t80 (the only ones used further down); pt0
  Pos pt0;
  {
    const int q = min(lane, 6);
    const int p = S.sl->imin[q];
    const bool has = (q < 5) ? p != 0x7fffffff : S.sl->isum[IS_CNT0 + q] > 0;
    const float frac = (q == 0) ? 0.1f : (q == 1) ? 0.5f : (q == 2) ? 0.8f : (q == 3) ? 0.9f : 0.99f;
    const float thr = (q < 5) ? e_max * frac : P.t0_thr;   // same products as thr_tx[]
    Pos pp; pp.ip = 0; pp.fp = -P.t_first / P.dt;            // sample position of t = 0
    float us = 0.f;
    if (has) {
      float yl, yh; int base;
      if (q < 5) {
        yl = S.B0[p - 1]; yh = S.B0[p]; base = p - 1;
      } else {
        const bool inv = (q == 6);
        const TrapDev& t = (inv && !P.t0inv_same) ? P.t0inv : P.t0;
        yl = trap_at_y(S.B1, S.B0, p - 1, t); yh = trap_at_y(S.B1, S.B0, p, t);
        if (inv) { yl = -yl; yh = -yh; }
        base = p - 1 + (t.flen - 1);  // trailing alignment (A1): back to input index space
      }
      pp.ip = base; pp.fp = (thr - yl) / (yh - yl);
      us = (P.t_first + P.dt * ((float)base + pp.fp)) * P.inv_unit_per_us;
    } else {
      pp = pos_norm(pp);
    }
    ptx[1].ip = __builtin_amdgcn_readlane(pp.ip, 1); ptx[1].fp = readlane_f(pp.fp, 1);
    ptx[2].ip = __builtin_amdgcn_readlane(pp.ip, 2); ptx[2].fp = readlane_f(pp.fp, 2);
    pt0.ip = __builtin_amdgcn_readlane(pp.ip, 5); pt0.fp = readlane_f(pp.fp, 5);
    if (wave == 0) {
      if (lane < 7) S.outv[lane == 0 ? C_t10 : lane == 1 ? C_t50 : lane == 2 ? C_t80 : lane == 3 ? C_t90 : lane == 4 ? C_t99 : lane == 5 ? C_t0 : C_t0_inv] = us;
      const float t90 = readlane_f(us, 3), t0u = readlane_f(us, 5);
      if (lane == 0) S.outv[C_drift_time] = (t90 - t0u) * P.unit_per_us;
    }
  }
  if (P.dbg_stop == 5) return;

  STAMP(14);
  // ------------------------------------------------ phase 3c: signal estimators
  // e_trap = SignalEstimator(trap_opt output, t50 + rt + ft/2)     dsp_icpc.jl:163
  // the seven estimates are spread over the waves (each needs a full wave: lane l = window point l)
  {
    float* eslot = S.misc + 4;
    auto I = [&](int i) { return S.B1[i + 1]; };  // integrator output I[i] = T[i+1]  (dsp_routines.jl:53)
    // qdrift / lq (get_qdrift, dsp_routines.jl:51-64; lq from t80 with lq_int_length, dsp_icpc.jl:144): one wave each, the
    // integrator output taken relative to the first window point (qdrift.hpp) — exact where a difference of two float32
    // prefix sums of 1e8 would carry +-40
    for (int e = wave; e < 3; e += NW) {
      float v;
      if (e == 0) {
        Pos p = pos_add(ptx[1], P.trap_pickoff);
        p.ip -= (P.opt.flen - 1);
        v = estimate(P.sig_est, S.estB, p, L - P.opt.flen + 1, [&](int i) { return trap_at(S.B1, i, P.opt); });
      } else {
        const Pos base = (e == 1) ? pt0 : ptx[2];
        const Pos p1 = pos_add(base, e == 1 ? P.qdrift_d1 : P.lq_d1), p2 = pos_add(base, e == 1 ? P.qdrift_d2 : P.lq_d2);
        const int ips[3] = {base.ip, p1.ip, p2.ip};
        const float fps[3] = {base.fp, p1.fp, p2.fp};
        v = qdrift_wave(P.int_est, S.estB + EST_TBL, S.B0, L, ips, fps);
      }
      if (lane == 0) eslot[e] = v;
    }
    __syncthreads();
    if (tid == 0) {
      S.outv[C_e_trap] = eslot[0];
      S.outv[C_qdrift] = eslot[1];
      S.outv[C_lq] = eslot[2];
    }
  }
  if (P.dbg_stop == 6) return;


  STAMP(15);
  if constexpr (FUSE) {
    // ------------------------------------------- phase 5: CUSP / ZAC (dsp_icpc.jl:167-178)
    // (behind the estimators' barrier: nobody reads the mask words, B1[Lp..] or slots 0..1 any more)
    const int pad = cz_pad(P);
    for (int i = tid; i < pad; i += NT) S.B1[i - pad] = 0.f;  // the gap (dead mask words) becomes Dp[i < 0] = 0
    if (tid < 64) S.B1[Lp + tid] = 0.f;
    if (tid < 2) { S.sl->fmx[tid] = 0u; S.sl->imin[tid] = 0x7fffffff; }
    cz_body<NT, R, FULL, false, true, true, true>(S, P, y, ptx[1], scan_buf);  // barriers inside order the above
  } else {
    // CUSP / ZAC run in icpc_cz_kernel; hand over blmean and the t50 position
    if (tid == 0) {
      aux[4 * (size_t)blockIdx.x + 0] = blmean;
      aux[4 * (size_t)blockIdx.x + 1] = __int_as_float(ptx[1].ip);
      aux[4 * (size_t)blockIdx.x + 2] = ptx[1].fp;
    }
  }
  STAMP(22);
  // ---------------------------------------------------------------- outputs
  // every S.outv entry was stored by a lane of wave 0 and is read here by wave 0: program order suffices
  static_assert(C_NCOLS <= 64, "the output row is stored by wave 0");
  if (tid < C_NCOLS) {
    const bool cz_col = tid == C_e_cusp || tid == C_e_zac || tid == C_e_cusp_max || tid == C_e_zac_max ||
                        tid == C_t_cusp_max || tid == C_t_zac_max;
    float* dst = reinterpret_cast<float*>(out.col[tid]);
    if (dst && (FUSE || !cz_col)) dst[(size_t)blockIdx.x * (size_t)out.stride] = S.outv[tid];
  }
}

// ---------------------------------------------------------------------------
// BASELINE config 2: blmean -> shift -> InvCR -> Trap(10us,4us) -> maximum
// (reference src/dsp_icpc.jl:102-105,119-120,147-148).  Same load and scans as the
// fused kernel, nothing else: 4L+8 algorithmic bytes per trace.
template <int NT, int R, bool FULL>
__global__ void __launch_bounds__(NT)
pz_trap_kernel(const float* __restrict__ wf, const IcpcDev* __restrict__ Pp, float* __restrict__ o_blmean,
               float* __restrict__ o_e10410) {
  constexpr int NW = NT / 64, SP = 4 * R, Lp = NT * SP;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const IcpcDev& P = *Pp;
  const int L = FULL ? Lp : P.L, tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  float* B = reinterpret_cast<float*>(smem_raw);              // [Lp+64]  T
  double* part = reinterpret_cast<double*>(B + Lp + 64);      // [2][R*NW]
  double* wsum = part + 2 * R * NW;                           // [NW]
  float* fred = reinterpret_cast<float*>(wsum + NW);          // [NW]
  const float* w = wf + (size_t)blockIdx.x * (size_t)L;
  const uint16_t* w16 = reinterpret_cast<const uint16_t*>(wf) + (size_t)blockIdx.x * (size_t)L;   // (in_u16)
  float x[R][4];
  if (P.in_u16) load_trace_s4_u16<NT, R, FULL>(w16, L, tid, x);
  else load_trace_s4<NT, R, FULL>(w, L, tid, x);
  // baseline mean exactly as icpc_kernel forms it (same pivot, same summation order)
  const float pv_bl = P.in_u16 ? (float)w16[P.bl.from] : w[P.bl.from];
  WinAccF bl = {0, 0, 0};
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i0 = 4 * (tid + NT * r);
    winf_accum4(bl, P.bl, i0, 0.f, pv_bl, x[r][0], x[r][1], x[r][2], x[r][3]);
  }
  const float s1w = wave_incl_scan_sum(bl.s1);
  if (lane == 63) wsum[wave] = (double)s1w;
  __syncthreads();
  double s1 = 0;
  for (int ww = 0; ww < NW; ++ww) s1 += wsum[ww];
  const float blmean = (float)((double)pv_bl + s1 * P.bl.inv_n);
  float tot[R];
  double off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i0 = 4 * (tid + NT * r);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[r][e] = (i0 + e < L) ? x[r][e] - blmean : 0.f;
    tot[r] = (x[r][0] + x[r][1]) + (x[r][2] + x[r][3]);
  }
  s4_exscan_sum<NT, R>(tot, off, part, nullptr);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i0 = 4 * (tid + NT * r);
    const float coff = (float)(P.pz_c64 * off[r]);
    float run = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      run += x[r][e];
      x[r][e] = (i0 + e < L) ? (x[r][e] + coff) + P.pz_c * run : 0.f;
    }
    tot[r] = (x[r][0] + x[r][1]) + (x[r][2] + x[r][3]);
  }
  double tot_all;
  s4_exscan_sum<NT, R>(tot, off, part + R * NW, &tot_all);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double run = off[r];
    float4 t;
    float* pt = &t.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) { pt[e] = (float)run; run += (double)x[r][e]; }
    *reinterpret_cast<float4*>(&B[4 * (tid + NT * r)]) = t;
  }
  if (tid == 0) B[Lp] = (float)tot_all;
  __syncthreads();
  const TrapDev tr = P.fixed[0];
  const int nout = L - tr.flen + 1;
  float mx = -INFINITY;
  // unscaled trapezoid exactly as icpc_kernel's sweep B evaluates it; rows wholly inside the
  // output range skip the per-lane test
  const float* tb = &B[tid];
  const float *fa = tb + tr.n1, *fb = tb + tr.n1 + tr.g, *fc = tb + tr.flen;
#pragma unroll
  for (int m = 0; m < SP; ++m) {
    if (NT * (m + 1) <= nout) {
      mx = vmax(mx, fmaf(fc[NT * m] - fb[NT * m], tr.rr, -(fa[NT * m] - tb[NT * m])));
    } else if (NT * m < nout) {
      const float o = fmaf(fc[NT * m] - fb[NT * m], tr.rr, -(fa[NT * m] - tb[NT * m]));
      mx = vmax(mx, (tid + NT * m < nout) ? o : -INFINITY);
    }
  }
  mx *= tr.inv1;
  mx = wave_max_all(mx);
  if (lane == 0) fred[wave] = mx;
  __syncthreads();
  if (tid == 0) {
    for (int ww = 1; ww < NW; ++ww) mx = fmaxf(mx, fred[ww]);
    o_blmean[blockIdx.x] = blmean;
    o_e10410[blockIdx.x] = mx;
  }
}

// ---------------------------------------------------------------------------
// Trapezoid filter-optimisation grid scans (reference src/dsp_filter_optimization.jl:102-133 and
// :241-274): baseline mean -> shift -> InvCR -> T = prefix sum (as pz_trap_kernel), then for every
// grid point g the SignalEstimator of TrapezoidalChargeFilter_g's output at the pick-off, i.e.
// npts trapezoid samples (4 reads of T each) per grid point instead of a filtered trace.  One wave
// per grid point, round robin.  pick_mode 1 needs t50 first: half the maximum of y, Intersect.
template <int NT, int R, bool FULL>
__global__ void __launch_bounds__(NT)
trap_grid_kernel(const float* __restrict__ wf, const TrapGridDev* __restrict__ Pp, float* __restrict__ out, int64_t n) {
  constexpr int NW = NT / 64, SP = 4 * R, Lp = NT * SP, NWORDS = Lp / 32;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const TrapGridDev& P = *Pp;
  const int L = FULL ? Lp : P.L, tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  // ONE trace-sized array: it holds y while the t50 crossing is looked up (pick_mode 1), then T.  35 KB per workgroup at
  // L = 8192: four workgroups per CU keep four traces' loads in flight (with separate y and T arrays: two).
  float* T = reinterpret_cast<float*>(smem_raw);                                   // [Lp+64]
  float* Y = T;
  uint32_t* bm = reinterpret_cast<uint32_t*>(T + Lp + 64);                          // [NWORDS]
  double* part = reinterpret_cast<double*>(smem_raw + (size_t)(Lp + 64 + NWORDS) * 4);   // [2][R*NW]
  double* wsum = part + 2 * R * NW;                                                 // [NW]
  float* estB = reinterpret_cast<float*>(wsum + NW);                                // [EST_TBL]
  uint32_t* slot = reinterpret_cast<uint32_t*>(estB + EST_TBL);                     // [4] max(y), first crossing, count
  const float* w = wf + (size_t)blockIdx.x * (size_t)L;
  float x[R][4];
  load_trace_s4<NT, R, FULL>(w, L, tid, x);
  for (int i = tid; i < EST_TBL; i += NT) estB[i] = P.est.B[i];
  if (tid == 0) { slot[0] = 0u; slot[1] = 0x7fffffffu; slot[2] = 0u; }
  // baseline mean exactly as icpc_kernel forms it
  const float pv_bl = w[P.bl.from];
  WinAccF bl = {0, 0, 0};
#pragma unroll
  for (int r = 0; r < R; ++r) winf_accum4(bl, P.bl, 4 * (tid + NT * r), 0.f, pv_bl, x[r][0], x[r][1], x[r][2], x[r][3]);
  const float s1w = wave_incl_scan_sum(bl.s1);
  if (lane == 63) wsum[wave] = (double)s1w;
  __syncthreads();
  double s1 = 0;
  for (int ww = 0; ww < NW; ++ww) s1 += wsum[ww];
  const float blmean = (float)((double)pv_bl + s1 * P.bl.inv_n);
  float tot[R];
  double off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i0 = 4 * (tid + NT * r);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[r][e] = (i0 + e < L) ? x[r][e] - blmean : 0.f;
    tot[r] = (x[r][0] + x[r][1]) + (x[r][2] + x[r][3]);
  }
  s4_exscan_sum<NT, R>(tot, off, part, nullptr);
  float ymax = -INFINITY;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i0 = 4 * (tid + NT * r);
    const float coff = (float)(P.pz_c64 * off[r]);
    float run = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      run += x[r][e];
      x[r][e] = (i0 + e < L) ? (x[r][e] + coff) + P.pz_c * run : 0.f;
      ymax = vmax(ymax, (i0 + e < L) ? x[r][e] : -INFINITY);
    }
    tot[r] = (x[r][0] + x[r][1]) + (x[r][2] + x[r][3]);
    if (P.pick_mode == 1) *reinterpret_cast<float4*>(&Y[i0]) = make_float4(x[r][0], x[r][1], x[r][2], x[r][3]);
  }
  ymax = wave_max_all(ymax);
  if (lane == 0) atomicMax(&slot[0], ford(ymax));
  double tot_all;
  s4_exscan_sum<NT, R>(tot, off, part + R * NW, &tot_all);   // barrier inside: y (pick_mode 1) and the maximum are published
  // pick-off position (samples, int + frac)
  Pos base;
  base.ip = P.pick_ip; base.fp = P.pick_fp;
  if (P.pick_mode == 1) {  // t50 = get_threshold(wvfs, 0.5 * maximum; mintot)   dsp_filter_optimization.jl:260
    const float thr = 0.5f * ford_inv(slot[0]);
#pragma unroll
    for (int m = 0; m < SP; ++m) {
      const int k = tid + NT * m;
      const unsigned long long bq = __ballot(k < L && Y[k] >= thr);
      if (lane == 0) *reinterpret_cast<unsigned long long*>(&bm[(NT >> 5) * m + 2 * wave]) = bq;
    }
    __syncthreads();
    for (int wd = tid; wd < NWORDS; wd += NT) {
      int c, f;
      intersect_word(bm, wd, NWORDS, P.tx_mintot, &c, &f);
      if (c) { atomicAdd(&slot[2], (uint32_t)c); atomicMin(reinterpret_cast<int*>(&slot[1]), f); }
    }
    __syncthreads();
    if (slot[2] > 0) {
      const int p = (int)slot[1];
      const float yl = Y[p - 1], yh = Y[p];
      base.ip = p - 1; base.fp = (thr - yl) / (yh - yl);
    } else {               // no crossing: NaN -> 0 us (dsp_routines.jl:41), i.e. the position of t = 0
      base.ip = 0; base.fp = -P.t_first / P.dt;
      base = pos_norm(base);
    }
    __syncthreads();       // every read of y is done: the array becomes T
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double run = off[r];
    float4 t;
    float* pt = &t.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) { pt[e] = (float)run; run += (double)x[r][e]; }
    *reinterpret_cast<float4*>(&T[4 * (tid + NT * r)]) = t;
  }
  if (tid == 0) T[Lp] = (float)tot_all;
  if (tid < 63) T[Lp + 1 + tid] = 0.f;
  __syncthreads();
  for (int g = wave; g < P.G; g += NW) {
    const TrapDev tr = P.trap[g];
    Pos p = (P.pick_mode == 1) ? pos_add(base, P.offs[g]) : base;
    p.ip -= (tr.flen - 1);   // trailing alignment of the filter output (A1)
    const float v = estimate(P.est, estB, p, L - tr.flen + 1, [&](int i) { return trap_at(T, i, tr); });
    if (lane == 0) out[(size_t)g * (size_t)n + blockIdx.x] = v;
  }
}

template <int NT, int R, bool FULL>
static hipError_t launch_grid_t(const float* wf, int64_t n, const TrapGridDev* dP, float* out, hipStream_t st) {
  constexpr int NW = NT / 64, Lp = 16 * NT;
  const size_t smem = (size_t)(Lp + 64 + Lp / 32) * 4 + (2 * R * NW + NW) * 8 + EST_TBL * 4 + 32;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&trap_grid_kernel<NT, R, FULL>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((trap_grid_kernel<NT, R, FULL>), dim3((unsigned)n), dim3(NT), smem, st, wf, dP, out, n);
  return hipGetLastError();
}
hipError_t launch_trap_grid(const float* wf, int64_t n, int NT, bool full, const TrapGridDev* dP, float* out, hipStream_t st) {
#define LDSP_CASE(N) \
  case N: return full ? launch_grid_t<N, 4, true>(wf, n, dP, out, st) : launch_grid_t<N, 4, false>(wf, n, dP, out, st);
  switch (NT) {
    LDSP_ALL_CASES
    default: return hipErrorInvalidValue;
  }
#undef LDSP_CASE
}

// ---------------------------------------------------------------------------
// FIR filter-optimisation grid scans (CUSP / ZAC: reference src/dsp_filter_optimization.jl:145-229, 286-374).
// Front end as trap_grid_kernel (y only: no prefix sum); per grid point one wave evaluates, in direct form,
// just the npts filter outputs under the SignalEstimator window: lane l <-> output i0+l, consecutive lanes on
// consecutive LDS words, one scalar tap load per 16 taps.
template <int NT, int R, bool FULL>
__global__ void __launch_bounds__(NT, 6)   // <= 80 VGPRs: three 512-thread workgroups per CU (64 VGPRs would spill the accumulators)
fir_grid_kernel(const float* __restrict__ wf, const FirGridDev* __restrict__ Pp, float* __restrict__ out, int64_t n) {
  constexpr int NW = NT / 64, SP = 4 * R, Lp = NT * SP, NWORDS = Lp / 32;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const FirGridDev& P = *Pp;
  const int L = FULL ? Lp : P.L, tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  float* Y = reinterpret_cast<float*>(smem_raw);                                    // [Lp+64]
  uint32_t* bm = reinterpret_cast<uint32_t*>(Y + Lp + 64);                          // [NWORDS]
  double* part = reinterpret_cast<double*>(smem_raw + (size_t)(Lp + 64 + NWORDS) * 4);   // [R*NW]
  double* wsum = part + R * NW;                                                     // [NW]
  float* estB = reinterpret_cast<float*>(wsum + NW);                                // [EST_TBL]
  uint32_t* slot = reinterpret_cast<uint32_t*>(estB + EST_TBL);                     // [4]
  const float* w = wf + (size_t)blockIdx.x * (size_t)L;
  float x[R][4];
  load_trace_s4<NT, R, FULL>(w, L, tid, x);
  for (int i = tid; i < EST_TBL; i += NT) estB[i] = P.est.B[i];
  if (tid < 64) Y[Lp + tid] = 0.f;
  if (tid == 0) { slot[0] = 0u; slot[1] = 0x7fffffffu; slot[2] = 0u; }
  const float pv_bl = w[P.bl.from];
  WinAccF bl = {0, 0, 0};
#pragma unroll
  for (int r = 0; r < R; ++r) winf_accum4(bl, P.bl, 4 * (tid + NT * r), 0.f, pv_bl, x[r][0], x[r][1], x[r][2], x[r][3]);
  const float s1w = wave_incl_scan_sum(bl.s1);
  if (lane == 63) wsum[wave] = (double)s1w;
  __syncthreads();
  double s1 = 0;
  for (int ww = 0; ww < NW; ++ww) s1 += wsum[ww];
  const float blmean = (float)((double)pv_bl + s1 * P.bl.inv_n);
  float tot[R];
  double off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i0 = 4 * (tid + NT * r);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[r][e] = (i0 + e < L) ? x[r][e] - blmean : 0.f;
    tot[r] = (x[r][0] + x[r][1]) + (x[r][2] + x[r][3]);
  }
  s4_exscan_sum<NT, R>(tot, off, part, nullptr);
  float ymax = -INFINITY;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i0 = 4 * (tid + NT * r);
    const float coff = (float)(P.pz_c64 * off[r]);
    float run = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      run += x[r][e];
      x[r][e] = (i0 + e < L) ? (x[r][e] + coff) + P.pz_c * run : 0.f;
      ymax = vmax(ymax, (i0 + e < L) ? x[r][e] : -INFINITY);
    }
    *reinterpret_cast<float4*>(&Y[i0]) = make_float4(x[r][0], x[r][1], x[r][2], x[r][3]);
  }
  ymax = wave_max_all(ymax);
  if (lane == 0) atomicMax(&slot[0], ford(ymax));
  __syncthreads();
  Pos base;
  base.ip = P.pick_ip; base.fp = P.pick_fp;
  if (P.pick_mode == 1) {  // t50 = get_threshold(wvfs, 0.5 * maximum; mintot)
    const float thr = 0.5f * ford_inv(slot[0]);
#pragma unroll
    for (int m = 0; m < SP; ++m) {
      const int k = tid + NT * m;
      const unsigned long long bq = __ballot(k < L && Y[k] >= thr);
      if (lane == 0) *reinterpret_cast<unsigned long long*>(&bm[(NT >> 5) * m + 2 * wave]) = bq;
    }
    __syncthreads();
    for (int wd = tid; wd < NWORDS; wd += NT) {
      int c, f;
      intersect_word(bm, wd, NWORDS, P.tx_mintot, &c, &f);
      if (c) { atomicAdd(&slot[2], (uint32_t)c); atomicMin(reinterpret_cast<int*>(&slot[1]), f); }
    }
    __syncthreads();
    if (slot[2] > 0) {
      const int p = (int)slot[1];
      const float yl = Y[p - 1], yh = Y[p];
      base.ip = p - 1; base.fp = (thr - yl) / (yh - yl);
    } else {
      base.ip = 0; base.fp = -P.t_first / P.dt;
      base = pos_norm(base);
    }
  }
  const int Lf = P.Lf, nout = L - Lf + 1;
  if (P.same_offs) {
    // Every grid point picks off at the same position, and the estimator is linear in its window:
    //   E[g] = sum_l w[l] * (sum_j c_g[j] y[i0+l+j]) = sum_j c_g[j] * z[j],   z[j] = sum_l w[l] y[i0+l+j].
    // z is ONE npts-tap pass over the trace (thread <-> j, consecutive lanes on consecutive LDS words); each grid point
    // then costs one dot product with z (taps read coalesced from L2): (npts + G) * Lf multiply-adds per trace instead of
    // npts * G * Lf.  Partial sums: per thread over its j, DPP inside the wave, wave partials combined in fixed order.
    float* wpart = reinterpret_cast<float*>(slot + 8);                                // [NW][LDSP_MAX_GRID]
    Pos p = (P.pick_mode == 1) ? pos_add(base, P.offs[0]) : base;
    p.ip -= (Lf - 1);   // trailing alignment of the filter output (A1)
    const int npts = P.est.npts;
    if (nout < npts) {                                   // estimate(): window longer than the filter output
      if (tid < P.G) out[(size_t)tid * (size_t)n + blockIdx.x] = NAN;
      return;
    }
    if (p.ip < 0) { p.ip = 0; p.fp = 0.f; }              // the clamps of estimate()
    if (p.ip >= nout - 1) { p.ip = nout - 1; p.fp = 0.f; }
    int i0 = p.ip + (int)ceilf(p.fp - 0.5f * (float)npts);
    i0 = max(0, min(i0, nout - npts));
    const float u = ((float)(p.ip - i0) + p.fp - P.est.c) * P.est.s_inv;
    const float wv = (lane < npts) ? est_weight(P.est, estB, lane, u) : 0.f;   // every wave holds the weights, lane l <-> point l
    float acc[LDSP_MAX_GRID];
#pragma unroll
    for (int g = 0; g < LDSP_MAX_GRID; ++g) acc[g] = 0.f;
    const int G = P.G;
    for (int j0 = 0; j0 < Lf; j0 += NT) {
      const int j = j0 + tid;
      const bool in = j < Lf;
      const float* yp = &Y[i0 + (in ? j : 0)];
      float z0 = 0.f, z1 = 0.f;
      int l = 0;
      for (; l + 4 <= npts; l += 4) {
        z0 = fmaf(readlane_f(wv, l), yp[l], z0); z1 = fmaf(readlane_f(wv, l + 1), yp[l + 1], z1);
        z0 = fmaf(readlane_f(wv, l + 2), yp[l + 2], z0); z1 = fmaf(readlane_f(wv, l + 3), yp[l + 3], z1);
      }
      for (; l < npts; ++l) z0 = fmaf(readlane_f(wv, l), yp[l], z0);
      const float z = in ? z0 + z1 : 0.f;
      const float* taps = P.taps;
      uint32_t idx = in ? (uint32_t)j : 0u;   // one 32-bit running index (G * Lf <= 2^19) on ONE scalar base: no per-point base pointers
#pragma unroll
      for (int g = 0; g < LDSP_MAX_GRID; ++g)
        if (g < G) { acc[g] = fmaf(taps[idx], z, acc[g]); idx += (uint32_t)Lf; }   // (g < G: scalar test, the tail of the unrolled body is skipped)
    }
#pragma unroll
    for (int g = 0; g < LDSP_MAX_GRID; ++g)
      if (g < G) {
        const float s = wave_sum_all(acc[g]);
        if (lane == 0) wpart[wave * LDSP_MAX_GRID + g] = s;
      }
    __syncthreads();
    if (tid < G) {
      float e = 0.f;
      for (int ww = 0; ww < NW; ++ww) e += wpart[ww * LDSP_MAX_GRID + tid];
      out[(size_t)tid * (size_t)n + blockIdx.x] = e;
    }
    return;
  }
  for (int g = wave; g < P.G; g += NW) {
    const float* c = P.taps + (size_t)g * (size_t)Lf;   // wave-uniform pointer: the tap reads are scalar loads
    Pos p = (P.pick_mode == 1) ? pos_add(base, P.offs[g]) : base;
    p.ip -= (Lf - 1);   // trailing alignment of the filter output (A1)
    const float v = estimate(P.est, estB, p, nout, [&](int i) {
      const float* yp = &Y[i];
      float a0 = 0.f, a1 = 0.f;
      int j = 0;
      for (; j + 16 <= Lf; j += 16) {
#pragma unroll
        for (int u = 0; u < 16; u += 2) { a0 = fmaf(c[j + u], yp[j + u], a0); a1 = fmaf(c[j + u + 1], yp[j + u + 1], a1); }
      }
      for (; j < Lf; ++j) a0 = fmaf(c[j], yp[j], a0);
      return a0 + a1;
    });
    if (lane == 0) out[(size_t)g * (size_t)n + blockIdx.x] = v;
  }
}

template <int NT, int R, bool FULL>
static hipError_t launch_fir_grid_t(const float* wf, int64_t n, const FirGridDev* dP, float* out, hipStream_t st) {
  constexpr int NW = NT / 64, Lp = 16 * NT;
  const size_t smem = (size_t)(Lp + 64 + Lp / 32) * 4 + (R * NW + NW) * 8 + EST_TBL * 4 + 32 + (size_t)NW * LDSP_MAX_GRID * 4;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fir_grid_kernel<NT, R, FULL>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((fir_grid_kernel<NT, R, FULL>), dim3((unsigned)n), dim3(NT), smem, st, wf, dP, out, n);
  return hipGetLastError();
}
hipError_t launch_fir_grid(const float* wf, int64_t n, int NT, bool full, const FirGridDev* dP, float* out, hipStream_t st) {
#define LDSP_CASE(N) \
  case N: return full ? launch_fir_grid_t<N, 4, true>(wf, n, dP, out, st) : launch_fir_grid_t<N, 4, false>(wf, n, dP, out, st);
  switch (NT) {
    LDSP_ALL_CASES
    default: return hipErrorInvalidValue;
  }
#undef LDSP_CASE
}

// ---------------------------------------------------------------------------
// dsp_sg_optimization (reference src/dsp_filter_optimization.jl:393-441): front end as trap_grid_kernel (blmean +
// slope, pole-zero, T, t50), the trapezoid energy at t50 + rt + ft/2, then one wave per SG window length: the
// derivative filter only inside the current window (lane-strided), first-occurrence arg-max, parabola vertex.
template <int NT, int R, bool FULL>
__global__ void __launch_bounds__(NT, 8)   // <= 64 VGPRs: with the single trace-sized LDS array, four 512-thread workgroups per CU
sg_grid_kernel(const float* __restrict__ wf, const SgGridDev* __restrict__ Pp, float* __restrict__ amax, float* __restrict__ energy,
               float* __restrict__ t50_us, float* __restrict__ o_blmean, float* __restrict__ o_blslope, int64_t n) {
  constexpr int NW = NT / 64, SP = 4 * R, Lp = NT * SP, NWORDS = Lp / 32;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const SgGridDev& P = *Pp;
  const int L = FULL ? Lp : P.L, tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  // ONE trace-sized array: y for the t50 crossing and the window-length scan, then T for the trapezoid energy (36 KB per
  // workgroup at L = 8192: four workgroups per CU instead of two)
  float* T = reinterpret_cast<float*>(smem_raw);                                   // [Lp+64]
  float* Y = T;
  uint32_t* bm = reinterpret_cast<uint32_t*>(T + Lp + 64);                          // [NWORDS]
  double* part = reinterpret_cast<double*>(smem_raw + (size_t)(Lp + 64 + NWORDS) * 4);   // [2][R*NW]
  double* wsum = part + 2 * R * NW;                                                 // [3][NW]
  float* estB = reinterpret_cast<float*>(wsum + 3 * NW);                            // [EST_TBL]
  uint32_t* slot = reinterpret_cast<uint32_t*>(estB + EST_TBL);                     // [4]
  unsigned long long* vi = reinterpret_cast<unsigned long long*>(slot + 8);         // [LDSP_MAX_GRID] arg-max per window length
  const float* w = wf + (size_t)blockIdx.x * (size_t)L;
  float x[R][4];
  load_trace_s4<NT, R, FULL>(w, L, tid, x);
  for (int i = tid; i < EST_TBL; i += NT) estB[i] = P.est.B[i];
  if (tid < 64) Y[Lp + tid] = 0.f;
  if (tid < LDSP_MAX_GRID) vi[tid] = 0ull;
  if (tid == 0) { slot[0] = 0u; slot[1] = 0x7fffffffu; slot[2] = 0u; }
  const float pv_bl = w[P.bl.from];
  WinAccF bl = {0, 0, 0};
  const float fic = (float)P.bl.ic;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i0 = 4 * (tid + NT * r);
    winf_accum4(bl, P.bl, i0, (float)i0 - fic, pv_bl, x[r][0], x[r][1], x[r][2], x[r][3]);
  }
  win_publish<NW>(bl, wsum, 0);
  __syncthreads();
  const float blmean = (float)((double)pv_bl + win_collect1<NW>(wsum, 0) * P.bl.inv_n);
  if (tid == 0) {
    float m_, sg_, sl_, of_;
    win_finish(win_collect<NW>(wsum, 0), P.bl, pv_bl, P.t_first, P.dt, &m_, &sg_, &sl_, &of_);
    if (o_blmean) o_blmean[blockIdx.x] = blmean;
    if (o_blslope) o_blslope[blockIdx.x] = sl_;
  }
  float tot[R];
  double off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i0 = 4 * (tid + NT * r);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[r][e] = (i0 + e < L) ? x[r][e] - blmean : 0.f;
    tot[r] = (x[r][0] + x[r][1]) + (x[r][2] + x[r][3]);
  }
  s4_exscan_sum<NT, R>(tot, off, part, nullptr);
  float ymax = -INFINITY;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i0 = 4 * (tid + NT * r);
    const float coff = (float)(P.pz_c64 * off[r]);
    float run = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      run += x[r][e];
      x[r][e] = (i0 + e < L) ? (x[r][e] + coff) + P.pz_c * run : 0.f;
      ymax = vmax(ymax, (i0 + e < L) ? x[r][e] : -INFINITY);
    }
    tot[r] = (x[r][0] + x[r][1]) + (x[r][2] + x[r][3]);
    *reinterpret_cast<float4*>(&Y[i0]) = make_float4(x[r][0], x[r][1], x[r][2], x[r][3]);
  }
  ymax = wave_max_all(ymax);
  if (lane == 0) atomicMax(&slot[0], ford(ymax));
  double tot_all;
  s4_exscan_sum<NT, R>(tot, off, part + R * NW, &tot_all);   // barrier inside: y and the maximum are published
  // t50 = get_threshold(wvfs, 0.5 * maximum; mintot)   :421
  const float thr = 0.5f * ford_inv(slot[0]);
#pragma unroll
  for (int m = 0; m < SP; ++m) {
    const int k = tid + NT * m;
    const unsigned long long bq = __ballot(k < L && Y[k] >= thr);
    if (lane == 0) *reinterpret_cast<unsigned long long*>(&bm[(NT >> 5) * m + 2 * wave]) = bq;
  }
  __syncthreads();
  for (int wd = tid; wd < NWORDS; wd += NT) {
    int c, f;
    intersect_word(bm, wd, NWORDS, P.tx_mintot, &c, &f);
    if (c) { atomicAdd(&slot[2], (uint32_t)c); atomicMin(reinterpret_cast<int*>(&slot[1]), f); }
  }
  __syncthreads();
  Pos base;
  float t50 = 0.f;
  if (slot[2] > 0) {
    const int p = (int)slot[1];
    const float yl = Y[p - 1], yh = Y[p];
    base.ip = p - 1; base.fp = (thr - yl) / (yh - yl);
    t50 = (P.t_first + P.dt * ((float)base.ip + base.fp)) * P.inv_unit_per_us;
  } else {
    base.ip = 0; base.fp = -P.t_first / P.dt;
    base = pos_norm(base);
  }
  if (tid == 0 && t50_us) t50_us[blockIdx.x] = t50;
  // SG window-length scan: every wave works on every window length (thread <-> output sample of the current window, four
  // independent multiply-add chains per output so that the LDS reads of a tap group are in flight together); the arg-max of a
  // window length is combined through an LDS slot, its parabola refinement is done by one wave per window length.
  auto sg_at = [&](const float* c, int np, int k) {
    const float* yp = &Y[k];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
    for (; i + 4 <= np; i += 4) {
      a0 = fmaf(c[i], yp[i], a0); a1 = fmaf(c[i + 1], yp[i + 1], a1);
      a2 = fmaf(c[i + 2], yp[i + 2], a2); a3 = fmaf(c[i + 3], yp[i + 3], a3);
    }
    for (; i < np; ++i) a0 = fmaf(c[i], yp[i], a0);
    return (a0 + a1) + (a2 + a3);
  };
  for (int g = 0; g < P.W; ++g) {
    const int np = P.np[g], from = P.from[g], until = P.until[g];
    const float* c = P.c[g];
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int k = from + tid; k <= until; k += NT) {
      const float v = sg_at(c, np, k);
      if (v > bv) { bv = v; bi = k; }
    }
    if (__ballot(bi != 0x7fffffff) != 0ull) {   // wave-uniform: this wave holds samples of the window
      const unsigned long long best = wave_max_u64(pack_vi(bv, bi));
      if (lane == 0) atomicMax(&vi[g], best);
    }
  }
  __syncthreads();
  for (int g = wave; g < P.W; g += NW) {
    const int np = P.np[g], from = P.from[g], until = P.until[g];
    const float* c = P.c[g];
    float v; int i;
    unpack_vi(vi[g], &v, &i);
    if (i > from && i < until) {   // get_wvf_maximum (src/interpolation.jl:30-46): lanes 0..2 evaluate the three samples
      const float e = (lane < 3) ? sg_at(c, np, i - 1 + lane) : 0.f;
      v = extrema3points(__shfl(e, 0), __shfl(e, 1), __shfl(e, 2));
    }
    if (lane == 0 && amax) amax[(size_t)g * (size_t)n + blockIdx.x] = v;
  }
  __syncthreads();   // every read of y is done: the array becomes T
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double run = off[r];
    float4 t;
    float* pt = &t.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) { pt[e] = (float)run; run += (double)x[r][e]; }
    *reinterpret_cast<float4*>(&T[4 * (tid + NT * r)]) = t;
  }
  if (tid == 0) T[Lp] = (float)tot_all;
  __syncthreads();
  if (wave == 0) {   // energy: trapezoid output at t50 + rt + ft/2 through the estimator
    Pos p = pos_add(base, P.trap_off);
    p.ip -= (P.trap.flen - 1);
    const TrapDev tr = P.trap;
    const float e = estimate(P.est, estB, p, L - tr.flen + 1, [&](int i) { return trap_at(T, i, tr); });
    if (lane == 0 && energy) energy[blockIdx.x] = e;
  }
}

template <int NT, int R, bool FULL>
static hipError_t launch_sg_grid_t(const float* wf, int64_t n, const SgGridDev* dP, float* amax, float* energy, float* t50, float* blm, float* bls,
                                   hipStream_t st) {
  constexpr int NW = NT / 64, Lp = 16 * NT;
  const size_t smem = (size_t)(Lp + 64 + Lp / 32) * 4 + (2 * R * NW + 3 * NW) * 8 + EST_TBL * 4 + 32 + LDSP_MAX_GRID * 8;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sg_grid_kernel<NT, R, FULL>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((sg_grid_kernel<NT, R, FULL>), dim3((unsigned)n), dim3(NT), smem, st, wf, dP, amax, energy, t50, blm, bls, n);
  return hipGetLastError();
}
hipError_t launch_sg_grid(const float* wf, int64_t n, int NT, bool full, const SgGridDev* dP, float* amax, float* energy, float* t50, float* blm,
                          float* bls, hipStream_t st) {
#define LDSP_CASE(N)                                                                                   \
  case N:                                                                                              \
    return full ? launch_sg_grid_t<N, 4, true>(wf, n, dP, amax, energy, t50, blm, bls, st)             \
                : launch_sg_grid_t<N, 4, false>(wf, n, dP, amax, energy, t50, blm, bls, st);
  switch (NT) {
    LDSP_ALL_CASES
    default: return hipErrorInvalidValue;
  }
#undef LDSP_CASE
}

// Largest dynamic LDS size that still lets two workgroups share a CU (160 KiB, 1280-byte granules)
constexpr size_t LDS_TWO_PER_CU = 80640;

template <int NT, int R, bool FULL>
static hipError_t launch_icpc_t(const float* wf, int64_t n, const IcpcDev* dP, float* aux, const IcpcOutDev& out,
                                const float* ext_bl, float ext_bl_scale, bool direct, bool cz_shared, bool fuse_ok, int stop_after_main, int cz_pad_floats, hipStream_t st,
                                hipEvent_t mid, int* stages) {
  using SM = Smem<NT, R>;
  // fused single launch: the normal path (both filters share their geometry, closed form, the gap fits)
  const size_t smem_fused = SM::bytes(std::max(SM::MASK_FLOATS, cz_pad_floats));
  if (fuse_ok && !direct && cz_shared && smem_fused <= LDS_TWO_PER_CU) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&icpc_kernel<NT, R, FULL, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_fused);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((icpc_kernel<NT, R, FULL, true>), dim3((unsigned)n), dim3(NT), smem_fused, st, wf, dP, aux, out, ext_bl, ext_bl_scale);
    *stages = 1;
    return hipGetLastError();
  }
  const size_t smem = SM::bytes(SM::MASK_FLOATS);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&icpc_kernel<NT, R, FULL, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((icpc_kernel<NT, R, FULL, false>), dim3((unsigned)n), dim3(NT), smem, st, wf, dP, aux, out, ext_bl, ext_bl_scale);
  e = hipGetLastError();
  *stages = 1;
  if (e == hipSuccess && mid) e = hipEventRecord(mid, st);  // stage boundary for per-kernel timing
  if (e != hipSuccess || stop_after_main) return e;
  *stages = 2;
  const size_t smem_cz = Smem<NT, R, false>::bytes(cz_pad_floats);
  auto launch_cz = [&](auto kern) -> hipError_t {
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_cz);
    if (e2 != hipSuccess) return e2;
    hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(NT), smem_cz, st, wf, dP, (const float*)aux, out);
    return hipGetLastError();
  };
  if (direct) return launch_cz(&icpc_cz_kernel<NT, R, FULL, true, true, true>);
  if (cz_shared) return launch_cz(&icpc_cz_kernel<NT, R, FULL, false, true, true>);
  e = launch_cz(&icpc_cz_kernel<NT, R, FULL, false, true, false>);
  if (e != hipSuccess) return e;
  return launch_cz(&icpc_cz_kernel<NT, R, FULL, false, false, true>);
}
// `full`: the trace length equals the tile (L == 16*NT), the specialisation without bounds tests.
// *stages: number of timed stages of this call (1 = fused launch, 2 = icpc_kernel + icpc_cz_kernel)
hipError_t launch_icpc(const float* wf, int64_t n, int NT, int R, bool full, const IcpcDev* dP, float* aux, const IcpcOutDev& out,
                       const float* ext_bl, float ext_bl_scale, bool direct, bool cz_shared, bool fuse_ok, int stop_after_main, int cz_pad_floats, hipStream_t st, hipEvent_t mid,
                       int* stages) {
  if (R == 2) {   // 8192 samples on 1024 threads (8 samples per thread)
    if (NT != 1024) return hipErrorInvalidValue;
    return full ? launch_icpc_t<1024, 2, true>(wf, n, dP, aux, out, ext_bl, ext_bl_scale, direct, cz_shared, fuse_ok, stop_after_main, cz_pad_floats, st, mid, stages)
                : launch_icpc_t<1024, 2, false>(wf, n, dP, aux, out, ext_bl, ext_bl_scale, direct, cz_shared, fuse_ok, stop_after_main, cz_pad_floats, st, mid, stages);
  }
#define LDSP_CASE(N)                                                                                                                 \
  case N:                                                                                                                            \
    return full ? launch_icpc_t<N, 4, true>(wf, n, dP, aux, out, ext_bl, ext_bl_scale, direct, cz_shared, fuse_ok, stop_after_main, cz_pad_floats, st, mid, stages) \
                : launch_icpc_t<N, 4, false>(wf, n, dP, aux, out, ext_bl, ext_bl_scale, direct, cz_shared, fuse_ok, stop_after_main, cz_pad_floats, st, mid, stages);
  switch (NT) {
    LDSP_ALL_CASES
    default: return hipErrorInvalidValue;
  }
#undef LDSP_CASE
}

template <int NT, int R, bool FULL>
static hipError_t launch_pz_t(const float* wf, int64_t n, const IcpcDev* dP, float* a, float* b, hipStream_t st) {
  constexpr int NW = NT / 64;
  const size_t smem = (size_t)(NT * 4 * R + 64) * 4 + 2 * R * NW * 8 + NW * 8 + NW * 4 + 16;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pz_trap_kernel<NT, R, FULL>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((pz_trap_kernel<NT, R, FULL>), dim3((unsigned)n), dim3(NT), smem, st, wf, dP, a, b);
  return hipGetLastError();
}
hipError_t launch_pz_trap(const float* wf, int64_t n, int NT, bool full, const IcpcDev* dP, float* blmean, float* e10410, hipStream_t st) {
#define LDSP_CASE(N) \
  case N: return full ? launch_pz_t<N, 4, true>(wf, n, dP, blmean, e10410, st) : launch_pz_t<N, 4, false>(wf, n, dP, blmean, e10410, st);
  switch (NT) {
    LDSP_ALL_CASES
    default: return hipErrorInvalidValue;
  }
#undef LDSP_CASE
}

size_t icpc_smem_bytes(int NT) { return 0; }

}  // namespace ldsp
