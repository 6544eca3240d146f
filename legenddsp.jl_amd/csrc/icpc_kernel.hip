// icpc_kernel.hip — fused dsp_icpc for gfx950 (MI355X).
//
// One workgroup per trace, NT = L/32 threads.  The trace is read from HBM
// exactly once (coalesced 16 B/lane), staged in LDS, and every one of the 48
// output columns of reference src/dsp_icpc.jl:62-230 is produced from LDS /
// registers; the only HBM writes are the 48 x 4 B of the output row.
//
//   phase 0  load -> LDS (swizzled) -> thread-blocked registers
//   phase 1  saturation, signalstats(bl), shift, max/min, tailstats,
//            pole-zero as prefix scan (y = x + c*cumsum(x)), signalstats(tail)
//   phase 2  prefix sum T of y -> LDS      (all trapezoids + integrator share it)
//   phase 3  lane-strided sweep: 5 trapezoids (+inverted outputs by linearity),
//            threshold bit-masks by wave ballot; Intersect scans on bit-masks
//   phase 3c signal estimators (e_trap, qdrift, lq)
//   phase 4  Savitzky-Golay derivative x3 + plain derivative, current maxima,
//            in-trace pile-up, t50_current
//   phase 5  CUSP / ZAC: closed-form sliding exponential / parabola sums
//            (forward + backward one-pole scans, double prefix sum), or the
//            direct-form FIR comparator (cusp_mode 0)
//
// No MFMA: these are 1-D recursions and sliding sums, not contractions.
#include <hip/hip_runtime.h>
#include <math.h>
#include "icpc_dev.hpp"
#include "ldsp_device.hpp"

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

struct Smem {
  float* B0;      // [Lp]      y (later scratch)
  float* B1;      // [Lp+32]   T = exclusive prefix sum of y (later scratch)
  uint32_t* bm;   // [NMASK][NT]
  double* red;    // [MAX_WAVES*8]
  float* outv;    // [C_NCOLS]
  float* misc;    // [64]
};

__device__ __forceinline__ float trap_at(const float* T, int k, const TrapDev& t) {
  float a = T[sw(k + t.flen)] - T[sw(k + t.n1 + t.g)];
  float b = T[sw(k + t.n1)] - T[sw(k)];
  return a * t.inv2 - b * t.inv1;
}

// LSQ-polynomial estimate at position p (index space of a signal of length
// nsig whose samples are produced by getval(i)); computed redundantly by every
// wave, lane l handles window point l.  Assumption A3.
template <typename F>
__device__ __forceinline__ float estimate(const EstDev& E, Pos p, int nsig, F getval) {
  if (nsig < E.npts) return NAN;
  if (p.ip < 0) { p.ip = 0; p.fp = 0.f; }
  if (p.ip >= nsig - 1) { p.ip = nsig - 1; p.fp = 0.f; }
  // i0 = ceil(p - npts/2) with p = ip + fp
  float h = 0.5f * (float)E.npts;
  float r = p.fp - h;  // in (-h, 1-h)
  int i0 = p.ip + (int)ceilf(r);
  i0 = max(0, min(i0, nsig - E.npts));
  float u = ((float)(p.ip - i0) + p.fp - E.c) * E.s_inv;
  const int l = lane_id();
  float v = 0.f;
  if (l < E.npts) {
    const float* b = &E.B[l * (LDSP_MAX_EST_DEG + 1)];
    float w = b[E.deg];
    for (int j = E.deg - 1; j >= 0; --j) w = fmaf(w, u, b[j]);
    v = w * getval(i0 + l);
  }
  return wave_reduce(v, OpSum());
}

struct WinAcc {
  double s1, s2, s3;
};
__device__ __forceinline__ void win_accum(WinAcc& a, const WinDev& w, int i, float v) {
  if (i >= w.from && i <= w.until) {
    double d = (double)v;
    a.s1 += d;
    a.s2 = fma(d, d, a.s2);
    a.s3 = fma((double)i - w.ic, d, a.s3);
  }
}
// (mean, sigma, slope per time unit, offset) from window sums — the arithmetic of
// signalstats (RadiationDetectorDSP; restated in oracle/ldsp_oracle.c:orc_signalstats)
__device__ __forceinline__ void win_finish(const WinAcc& a, const WinDev& w, float t_first, float dt,
                                           float* mean, float* sigma, float* slope, float* offset) {
  double m = a.s1 * w.inv_n;
  double var = a.s2 * w.inv_n - m * m;
  if (var < 0) var = 0;
  double cov = a.s3 * w.inv_n;  // mean of xi is 0
  double sl_i = cov / w.var_i;
  double sl_t = sl_i / (double)dt;
  double mean_x = (double)t_first + w.ic * (double)dt;
  *mean = (float)m;
  *sigma = (float)sqrt(var);
  *slope = (float)sl_t;
  *offset = (float)(m - sl_t * mean_x);
}

}  // namespace

template <int NT_MAX>
__global__ void __launch_bounds__(NT_MAX)
icpc_kernel(const float* __restrict__ wf, const IcpcDev* __restrict__ Pp, IcpcOutDev out) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const IcpcDev& P = *Pp;
  const int L = P.L, NT = blockDim.x, tid = threadIdx.x;
  const int lane = lane_id(), wave = wave_id();
  const int Lp = NT * SPT;

  Smem S;
  S.B0 = reinterpret_cast<float*>(smem_raw);
  S.B1 = S.B0 + Lp;
  S.bm = reinterpret_cast<uint32_t*>(S.B1 + Lp + 32);
  S.red = reinterpret_cast<double*>(S.bm + NMASK * NT);
  S.outv = reinterpret_cast<float*>(S.red + MAX_WAVES * 8);
  S.misc = S.outv + C_NCOLS;

  const float* w = wf + (size_t)blockIdx.x * (size_t)L;

  // ------------------------------------------------------------ phase 0: load
  if ((L & 3) == 0) {
    const float4* w4 = reinterpret_cast<const float4*>(w);
#pragma unroll
    for (int m = 0; m < SPT / 4; ++m) {
      int i4 = tid + NT * m;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (4 * i4 < L) v = w4[i4];
      *reinterpret_cast<float4*>(&S.B0[sw(4 * i4)]) = v;
    }
  } else {
    for (int m = 0; m < SPT; ++m) {
      int i = tid + NT * m;
      S.B0[sw(i)] = (i < L) ? w[i] : 0.f;
    }
  }
  if (tid < 32) S.B1[Lp + tid] = 0.f;
  __syncthreads();

  float xr[SPT];
  const int i0t = SPT * tid;  // first sample of this thread
#pragma unroll
  for (int c = 0; c < SPT / 4; ++c) {
    float4 v = *reinterpret_cast<const float4*>(&S.B0[sw(i0t + 4 * c)]);
    xr[4 * c] = v.x; xr[4 * c + 1] = v.y; xr[4 * c + 2] = v.z; xr[4 * c + 3] = v.w;
  }
  const int nv = max(0, min(SPT, L - i0t));  // valid samples of this thread

  // ------------------------------------------------- phase 1: raw-trace stats
  int n_low = 0, n_high = 0;
  float rmax = -INFINITY, rmin = INFINITY;
  WinAcc bl = {0, 0, 0};
  {
    const bool in_bl = (i0t <= P.bl.until) && (i0t + SPT - 1 >= P.bl.from);
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      if (j < nv) {
        float v = xr[j];
        n_low += (v == P.sat_low);
        n_high += (v == P.sat_high);
        rmax = fmaxf(rmax, v);
        rmin = fminf(rmin, v);
        if (in_bl) win_accum(bl, P.bl, i0t + j, v);
      }
    }
  }
  {
    double v[5] = {bl.s1, bl.s2, bl.s3, (double)n_low, (double)n_high};
    block_reduce<5>(v, S.red, OpSum());
    bl.s1 = v[0]; bl.s2 = v[1]; bl.s3 = v[2];
    n_low = (int)v[3]; n_high = (int)v[4];
    float mm[1] = {rmax};
    block_reduce<1>(mm, reinterpret_cast<float*>(S.red), OpMax());
    rmax = mm[0];
    mm[0] = rmin;
    block_reduce<1>(mm, reinterpret_cast<float*>(S.red), OpMin());
    rmin = mm[0];
  }
  // saturation runs (reference src/saturation.jl:28-65): rare path, only when a
  // saturated sample exists.  Bit-pack per thread, thread 0 walks the words.
  int cons_low = 0, cons_high = 0;
  if (n_low > 0 || n_high > 0) {  // block-uniform
    uint32_t wl = 0, wh = 0;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      if (j < nv) {
        wl |= (xr[j] == P.sat_low) ? (1u << j) : 0u;
        wh |= (xr[j] == P.sat_high) ? (1u << j) : 0u;
      }
    }
    S.bm[tid] = wl;
    S.bm[NT + tid] = wh;
    __syncthreads();
    if (tid < 2) {
      const uint32_t* b = S.bm + tid * NT;
      int best = 0, run = 0;
      for (int wd = 0; wd < NT; ++wd) {
        uint32_t x = b[wd];
        if (x == 0xffffffffu) { run += 32; continue; }
        if (x == 0) { best = max(best, run); run = 0; continue; }
        for (int bb = 0; bb < 32; ++bb) {
          if ((x >> bb) & 1u) ++run;
          else { best = max(best, run); run = 0; }
        }
      }
      best = max(best, run);
      S.misc[40 + tid] = __int_as_float(best);
    }
    __syncthreads();
    cons_low = __float_as_int(S.misc[40]);
    cons_high = __float_as_int(S.misc[41]);
    __syncthreads();
  }

  float blmean, blsigma, blslope, bloffset;
  win_finish(bl, P.bl, P.t_first, P.dt, &blmean, &blsigma, &blslope, &bloffset);
  const float e_max = rmax - blmean, e_min = rmin - blmean;

  // shift_waveform(wvfs, -blmean)                                 dsp_icpc.jl:105
#pragma unroll
  for (int j = 0; j < SPT; ++j) xr[j] = (j < nv) ? xr[j] - blmean : 0.f;

  // tailstats on the shifted trace (src/tailstats.jl:22-72) + cumsum for pole-zero
  WinAcc tl = {0, 0, 0};
  int tail_bad = 0;
  float loc[SPT];  // local inclusive prefix
  {
    const bool in_tl = (i0t <= P.tail.until) && (i0t + SPT - 1 >= P.tail.from);
    float run = 0.f;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      run += xr[j];
      loc[j] = run;
      if (in_tl) {
        int i = i0t + j;
        if (i >= P.tail.from && i <= P.tail.until) {
          float v = xr[j];
          if (v <= 0.f) tail_bad = 1;
          else win_accum(tl, P.tail, i, logf(v));
        }
      }
    }
  }
  double s_off = block_exscan_f64((double)loc[SPT - 1], S.red, nullptr);
  {
    double v[4] = {tl.s1, tl.s2, tl.s3, (double)tail_bad};
    block_reduce<4>(v, S.red, OpSum());
    tl.s1 = v[0]; tl.s2 = v[1]; tl.s3 = v[2];
    tail_bad = v[3] > 0;
  }
  float tail_mean = 0.f, tail_sigma = 0.f, tail_tau = 0.f;
  if (!tail_bad) {
    float sl, of;
    win_finish(tl, P.tail, P.t_first, P.dt, &tail_mean, &tail_sigma, &sl, &of);
    tail_tau = -1.f / sl;
  }

  // InvCRFilter: y = x + c*cumsum(x)                               dsp_icpc.jl:119-120
  {
    const float coff = (float)(P.pz_c64 * s_off);
#pragma unroll
    for (int j = 0; j < SPT; ++j) xr[j] = (j < nv) ? (xr[j] + coff) + P.pz_c * loc[j] : 0.f;
  }
  float* yr = xr;  // from here on the registers hold the PZ-corrected trace y

  if (P.dbg_stop == 1) return;
  // ------------------------------------------------- phase 2: prefix sum of y
  WinAcc pz = {0, 0, 0};
  {
    const bool in_tl = (i0t <= P.tail.until) && (i0t + SPT - 1 >= P.tail.from);
    float run = 0.f;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      loc[j] = run;  // local EXCLUSIVE prefix
      run += yr[j];
      if (in_tl) win_accum(pz, P.tail, i0t + j, yr[j]);
    }
    double tot;
    double t_off = block_exscan_f64((double)run, S.red, &tot);
    // y -> B0, T -> B1 (T[i] = sum_{j<i} y[j], i = 0..L)
    __syncthreads();  // all reads of raw x in B0 are long done; keep the barrier explicit
#pragma unroll
    for (int c = 0; c < SPT / 4; ++c) {
      float4 vy = make_float4(yr[4 * c], yr[4 * c + 1], yr[4 * c + 2], yr[4 * c + 3]);
      *reinterpret_cast<float4*>(&S.B0[sw(i0t + 4 * c)]) = vy;
      float4 vt;
      vt.x = (float)(t_off + (double)loc[4 * c]);
      vt.y = (float)(t_off + (double)loc[4 * c + 1]);
      vt.z = (float)(t_off + (double)loc[4 * c + 2]);
      vt.w = (float)(t_off + (double)loc[4 * c + 3]);
      *reinterpret_cast<float4*>(&S.B1[sw(i0t + 4 * c)]) = vt;
    }
    if (tid == NT - 1) S.B1[sw(Lp)] = (float)tot;  // T[Lp] (== T[L] when L == Lp; y is 0 beyond L)
  }
  {
    double v[3] = {pz.s1, pz.s2, pz.s3};
    block_reduce<3>(v, S.red, OpSum());  // also publishes B0 / B1
    pz.s1 = v[0]; pz.s2 = v[1]; pz.s3 = v[2];
  }
  float tailmean, tailsigma, tailslope, tailoffset;
  win_finish(pz, P.tail, P.t_first, P.dt, &tailmean, &tailsigma, &tailslope, &tailoffset);
  // y[32t-1], needed again in phase 5 after B0 has been recycled
  const float yprev = (i0t > 0) ? S.B0[sw(i0t - 1)] : 0.f;

  if (P.dbg_stop == 2) return;
  // ------------------------------------------------ phase 3: lane-strided sweep
  const float thr_tx[5] = {e_max * 0.1f, e_max * 0.5f, e_max * 0.8f, e_max * 0.9f, e_max * 0.99f};
  float mx_f[3] = {-INFINITY, -INFINITY, -INFINITY}, mn_f[3] = {INFINITY, INFINITY, INFINITY};
  ValIdx mx_opt = {-INFINITY, 0x7fffffff};
  {
    const int nout_t0 = L - P.t0.flen + 1, nout_t0i = L - P.t0inv.flen + 1;
    const int nout_f0 = L - P.fixed[0].flen + 1, nout_f1 = L - P.fixed[1].flen + 1,
              nout_f2 = L - P.fixed[2].flen + 1, nout_opt = L - P.opt.flen + 1;
    const int wstep = NT >> 5;
    for (int m = 0; m < SPT; ++m) {
      const int k = tid + NT * m;
      const int wb = wstep * m + 2 * wave;
      const float yv = (k < L) ? S.B0[sw(k)] : -INFINITY;
#pragma unroll
      for (int q = 0; q < 5; ++q) ballot_store(yv >= thr_tx[q], S.bm + q * NT, wb);
      float o0 = -INFINITY;
      if (k < nout_t0) o0 = trap_at(S.B1, k, P.t0);
      ballot_store(o0 >= P.t0_thr, S.bm + M_T0 * NT, wb);
      float o0i;
      if (P.t0inv_same) o0i = (k < nout_t0) ? -o0 : -INFINITY;
      else o0i = (k < nout_t0i) ? -trap_at(S.B1, k, P.t0inv) : -INFINITY;
      ballot_store(o0i >= P.t0_thr, S.bm + M_T0INV * NT, wb);
      if (k < nout_f0) { float o = trap_at(S.B1, k, P.fixed[0]); mx_f[0] = fmaxf(mx_f[0], o); mn_f[0] = fminf(mn_f[0], o); }
      if (k < nout_f1) { float o = trap_at(S.B1, k, P.fixed[1]); mx_f[1] = fmaxf(mx_f[1], o); }
      if (k < nout_f2) { float o = trap_at(S.B1, k, P.fixed[2]); mx_f[2] = fmaxf(mx_f[2], o); mn_f[2] = fminf(mn_f[2], o); }
      if (k < nout_opt) { float o = trap_at(S.B1, k, P.opt); if (o > mx_opt.v) { mx_opt.v = o; mx_opt.i = k; } }
    }
  }
  {
    float v[3] = {mx_f[0], mx_f[1], mx_f[2]};
    block_reduce<3>(v, reinterpret_cast<float*>(S.red), OpMax());  // barrier: bit-masks visible
    mx_f[0] = v[0]; mx_f[1] = v[1]; mx_f[2] = v[2];
    float u[2] = {mn_f[0], mn_f[2]};
    block_reduce<2>(u, reinterpret_cast<float*>(S.red), OpMin());
    mn_f[0] = u[0]; mn_f[2] = u[1];
    mx_opt = block_reduce_vimax(mx_opt, S.red);
  }

  if (P.dbg_stop == 3) return;
  // Intersect scans on the bit-masks (thread w <-> word w)
  int cnt7[7], first7[7];
  {
    int mins[7], sums[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int min_n = (q < 5) ? P.tx_mintot : P.t0_mintot;
      int c = 0, f = 0x7fffffff;
      intersect_word(S.bm + q * NT, tid, NT, min_n, &c, &f);
      sums[q] = c; mins[q] = f;
    }
    block_reduce<7>(sums, reinterpret_cast<int*>(S.red), OpSum());
    block_reduce<7>(mins, reinterpret_cast<int*>(S.red), OpMin());
#pragma unroll
    for (int q = 0; q < 7; ++q) { cnt7[q] = sums[q]; first7[q] = mins[q]; }
  }
  // crossing positions (sample units, split int + frac); NaN -> 0 us (dsp_routines.jl:24,41)
  Pos ptx[5];
  float ttx[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    if (cnt7[q] > 0) {
      int p = first7[q];
      float yl = S.B0[sw(p - 1)], yh = S.B0[sw(p)];
      ptx[q].ip = p - 1;
      ptx[q].fp = (thr_tx[q] - yl) / (yh - yl);
      ttx[q] = (P.t_first + P.dt * ((float)(p - 1) + ptx[q].fp)) * P.inv_unit_per_us;
    } else {
      ttx[q] = 0.f;
      float p0 = -P.t_first / P.dt;  // position of t = 0
      ptx[q].ip = 0; ptx[q].fp = p0;
      ptx[q] = pos_norm(ptx[q]);
    }
  }
  Pos pt0;
  float t0_us, t0inv_us;
  {
    if (cnt7[M_T0] > 0) {
      int p = first7[M_T0];
      float yl = trap_at(S.B1, p - 1, P.t0), yh = trap_at(S.B1, p, P.t0);
      float fr = (P.t0_thr - yl) / (yh - yl);
      pt0.ip = p - 1 + (P.t0.flen - 1);  // trailing alignment (A1): back to input index space
      pt0.fp = fr;
      t0_us = (P.t_first + P.dt * ((float)pt0.ip + fr)) * P.inv_unit_per_us;
    } else {
      t0_us = 0.f;
      pt0.ip = 0; pt0.fp = -P.t_first / P.dt;
      pt0 = pos_norm(pt0);
    }
    if (cnt7[M_T0INV] > 0) {
      int p = first7[M_T0INV];
      const TrapDev& ti = P.t0inv_same ? P.t0 : P.t0inv;
      float yl = -trap_at(S.B1, p - 1, ti), yh = -trap_at(S.B1, p, ti);
      float fr = (P.t0_thr - yl) / (yh - yl);
      t0inv_us = (P.t_first + P.dt * ((float)(p - 1 + ti.flen - 1) + fr)) * P.inv_unit_per_us;
    } else {
      t0inv_us = 0.f;
    }
  }

  if (P.dbg_stop == 4) return;
  // ------------------------------------------------ phase 3c: signal estimators
  // e_trap = SignalEstimator(trap_opt output, t50 + rt + ft/2)     dsp_icpc.jl:163
  float e_trap;
  {
    Pos p = pos_add(ptx[1], P.trap_pickoff);
    p.ip -= (P.opt.flen - 1);
    const int nout = L - P.opt.flen + 1;
    e_trap = estimate(P.sig_est, p, nout, [&](int i) { return trap_at(S.B1, i, P.opt); });
  }
  // get_qdrift (dsp_routines.jl:51-64): integrator output I[i] = T[i+1]
  float qdrift, lq;
  {
    auto I = [&](int i) { return S.B1[sw(i + 1)]; };
    float a0 = estimate(P.int_est, pt0, L, I);
    float a1 = estimate(P.int_est, pos_add(pt0, P.qdrift_d1), L, I);
    float a2 = estimate(P.int_est, pos_add(pt0, P.qdrift_d2), L, I);
    qdrift = (a2 - a1) - (a1 - a0);
    float b0 = estimate(P.int_est, ptx[2], L, I);
    float b1 = estimate(P.int_est, pos_add(ptx[2], P.lq_d1), L, I);
    float b2 = estimate(P.int_est, pos_add(ptx[2], P.lq_d2), L, I);
    lq = (b2 - b1) - (b1 - b0);
  }

  if (P.dbg_stop == 5) return;
  // ----------------------------------- phase 4: SG derivatives, current maxima
  // lane-strided: g[k] = sum_i c[i] y[k+i] from LDS (valid mode, trailing time axis).
  // The SG(sg_wl) output is needed in full (pile-up scan, t50_current) and is
  // parked in B1 (T is dead after phase 3c; phase 5 regenerates it from registers);
  // SG(60ns), SG(100ns) and the plain derivative only inside the current window.
  float a_cur[4];
  float gmax = -INFINITY;
  WinAcc sgb = {0, 0, 0};
  const int ng = L - P.sg_npts[0] + 1;
  auto flt_at = [&](int f, int k) -> float {
    float g = 0.f;
    if (f < 3) {
      const int np = P.sg_npts[f];
      for (int i = 0; i < np; ++i) g = fmaf(P.sg_c[f][i], S.B0[sw(k + i)], g);
    } else {  // DerivativeFilter(1): y[max(i,1)] - y[max(i-1,0)]   (src/derivative.jl:47-55)
      g = S.B0[sw(max(k, 1))] - S.B0[sw(max(k - 1, 0))];
    }
    return g;
  };
  __syncthreads();  // phase 3c reads of T complete before B1 is recycled
  {
    ValIdx best[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { best[f].v = -INFINITY; best[f].i = 0x7fffffff; }
    for (int m = 0; m < SPT; ++m) {
      const int k = tid + NT * m;
      float g0 = -INFINITY;
      if (k < ng) {
        g0 = flt_at(0, k);
        gmax = fmaxf(gmax, g0);
        win_accum(sgb, P.sgbl, k, g0);
        if (k >= P.cur_from[0] && k <= P.cur_until[0] && g0 > best[0].v) { best[0].v = g0; best[0].i = k; }
      }
      S.B1[sw(k)] = g0;
#pragma unroll
      for (int f = 1; f < 4; ++f) {
        if (f == 2 && P.sg_same_02) continue;
        if (k >= P.cur_from[f] && k <= P.cur_until[f]) {
          float g = flt_at(f, k);
          if (g > best[f].v) { best[f].v = g; best[f].i = k; }
        }
      }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (f == 2 && P.sg_same_02) { a_cur[2] = a_cur[0]; continue; }
      ValIdx b = block_reduce_vimax(best[f], S.red);
      // get_wvf_maximum (src/interpolation.jl:30-46): parabola if strictly interior
      float amax = b.v;
      if (b.i > P.cur_from[f] && b.i < P.cur_until[f])
        amax = extrema3points(flt_at(f, b.i - 1), flt_at(f, b.i), flt_at(f, b.i + 1));
      a_cur[f] = amax;
    }
  }
  // in-trace pile-up threshold (dsp_routines.jl:75-77) and t50_current threshold (dsp_icpc.jl:192)
  float thr_intr, thr_sg50;
  {
    double v[3] = {sgb.s1, sgb.s2, sgb.s3};
    block_reduce<3>(v, S.red, OpSum());
    sgb.s1 = v[0]; sgb.s2 = v[1]; sgb.s3 = v[2];
    float mm[1] = {gmax};
    block_reduce<1>(mm, reinterpret_cast<float*>(S.red), OpMax());
    gmax = mm[0];
    float m_, sg_, sl_, of_;
    win_finish(sgb, P.sgbl, 0.f, P.dt, &m_, &sg_, &sl_, &of_);
    thr_intr = sg_ * P.intrace_nsigma;
    if (thr_intr == 0.f) thr_intr = 1.f;
    thr_sg50 = gmax * 0.5f;
  }
  {
    const int wstep = NT >> 5;
    for (int m = 0; m < SPT; ++m) {
      const int k = tid + NT * m;
      const float g = S.B1[sw(k)];
      ballot_store(g >= thr_sg50, S.bm + M_SG50 * NT, wstep * m + 2 * wave);
      ballot_store(g >= thr_intr, S.bm + M_INTR * NT, wstep * m + 2 * wave);
    }
  }
  __syncthreads();
  float t50cur_us, intr_x;
  int intr_n;
  {
    int c50, f50, ci, ei;
    intersect_word(S.bm + M_SG50 * NT, tid, NT, P.tx_mintot, &c50, &f50);
    intersect_word_rev(S.bm + M_INTR * NT, tid, NT, ng, P.intrace_mintot, &ci, &ei);
    int sums[2] = {c50, ci};
    block_reduce<2>(sums, reinterpret_cast<int*>(S.red), OpSum());
    int mn[1] = {f50};
    block_reduce<1>(mn, reinterpret_cast<int*>(S.red), OpMin());
    int mxv[1] = {ei};
    block_reduce<1>(mxv, reinterpret_cast<int*>(S.red), OpMax());
    const int np = P.sg_npts[0];
    auto gval = [&](int k) { return flt_at(0, k); };
    const float tg_first = P.t_first + P.dt * (float)(np - 1);  // trailing alignment (A1)
    if (sums[0] > 0) {
      int p = mn[0];
      float yl = gval(p - 1), yh = gval(p);
      t50cur_us = (tg_first + P.dt * ((float)(p - 1) + (thr_sg50 - yl) / (yh - yl))) * P.inv_unit_per_us;
    } else {
      t50cur_us = 0.f;
    }
    intr_n = sums[1];
    if (intr_n > 0) {
      // reversed index pos' = ng-1-e ; r[pos'-1] = g[e+1], r[pos'] = g[e]
      int e = mxv[0];
      int pr = ng - 1 - e;
      float yl = gval(e + 1), yh = gval(e);
      float xl = tg_first + P.dt * (float)(pr - 1);
      float xr_ = (thr_intr - yl) * P.dt / (yh - yl) + xl;
      intr_x = (tg_first + P.dt * (float)(ng - 1)) - xr_;  // last(time) - x   (dsp_routines.jl:81)
    } else {
      intr_x = NAN;
    }
  }

  if (P.dbg_stop == 6) return;
  // ------------------------------------------------------ phase 5: CUSP / ZAC
  float e_cz[2] = {NAN, NAN}, mx_cz[2] = {NAN, NAN}, tmx_cz[2] = {NAN, NAN};
  // extremestats + SignalEstimator on filter outputs held lane-strided in registers
  // (acc[m] = out[tid + NT*m]); f = 0 CUSP, 1 ZAC.          dsp_icpc.jl:170-171,177-178
  auto finish = [&](int f, int Lf, const float (&acc)[SPT]) {
    const int nout = L - Lf + 1;
    ValIdx best = {-INFINITY, 0x7fffffff};
    Pos p = pos_add(ptx[1], f ? P.zac_pickoff : P.cusp_pickoff);
    p.ip -= (Lf - 1);
    float esum = NAN;
    if (nout >= P.sig_est.npts) {
      if (p.ip < 0) { p.ip = 0; p.fp = 0.f; }
      if (p.ip >= nout - 1) { p.ip = nout - 1; p.fp = 0.f; }
      int i0 = p.ip + (int)ceilf(p.fp - 0.5f * (float)P.sig_est.npts);
      i0 = max(0, min(i0, nout - P.sig_est.npts));
      float u = ((float)(p.ip - i0) + p.fp - P.sig_est.c) * P.sig_est.s_inv;
      float part = 0.f;
#pragma unroll
      for (int m = 0; m < SPT; ++m) {
        int k = tid + NT * m;
        if (k < nout) {
          if (acc[m] > best.v) { best.v = acc[m]; best.i = k; }
          int l = k - i0;
          if (l >= 0 && l < P.sig_est.npts) {
            const float* b = &P.sig_est.B[l * (LDSP_MAX_EST_DEG + 1)];
            float wgt = b[P.sig_est.deg];
            for (int jj = P.sig_est.deg - 1; jj >= 0; --jj) wgt = fmaf(wgt, u, b[jj]);
            part = fmaf(wgt, acc[m], part);
          }
        }
      }
      float pv[1] = {part};
      block_reduce<1>(pv, reinterpret_cast<float*>(S.red), OpSum());
      esum = pv[0];
    } else {
#pragma unroll
      for (int m = 0; m < SPT; ++m) {
        int k = tid + NT * m;
        if (k < nout && acc[m] > best.v) { best.v = acc[m]; best.i = k; }
      }
    }
    best = block_reduce_vimax(best, S.red);
    e_cz[f] = esum;
    mx_cz[f] = best.v;
    tmx_cz[f] = P.t_first + P.dt * (float)(best.i + Lf - 1);
  };

  if (P.cusp_mode == 0) {
    // direct-form FIR comparator: out[k] = sum_j h[j] y[k+Lf-1-j]
    for (int f = 0; f < 2; ++f) {
      const CuspZacDev& Z = f ? P.zac : P.cusp;
      const float* h = f ? P.h_zac : P.h_cusp;
      const int Lf = Z.Lf, nout = L - Lf + 1;
      float acc[SPT];
#pragma unroll
      for (int m = 0; m < SPT; ++m) acc[m] = 0.f;
      for (int j = 0; j < Lf; ++j) {
        const float hj = h[j];
        const int sh = Lf - 1 - j;
#pragma unroll
        for (int m = 0; m < SPT; ++m) {
          int k = tid + NT * m;
          if (k < nout) acc[m] = fmaf(hj, S.B0[sw(k + sh)], acc[m]);
        }
      }
      finish(f, Lf, acc);
    }
  } else {
    // Closed form.  With d[i] = y[i] - a*y[i-1] (a = exp(-1/tau)):
    //   out[k] = sc * ( sum_{j<=Lf-2} w[j] d[n-j] + w[Lf-1] y[k] ),  n = k+Lf-1
    // and w = sinh flanks + flat top (+ parabolas for ZAC) splits into
    //   G[i] = sum_m q^m d[i-m]   causal one-pole      (thread-blocked scan, forward)
    //   A[i] = sum_m q^m d[i+m]   anti-causal one-pole (thread-blocked scan, backward)
    //   Dp[i] = sum_{m<=i} d[m] = y[i]-y[0]+eps*T[i]   (flat top; eps = 1-a)
    //   PRF = double prefix sum of a sparse combination of Dp (ZAC parabolas, in f64)
    // each read back lane-strided at a handful of fixed shifts.  B0/B1 are recycled.
    const int npass = P.cz_shared ? 1 : 2;
    for (int pass = 0; pass < npass; ++pass) {
      const bool want_c = P.cz_shared || pass == 0;
      const bool want_z = P.cz_shared || pass == 1;
      const CuspZacDev& Z = (pass == 0) ? P.cusp : P.zac;  // geometry + exponentials of this pass
      const CuspZacDev& ZZ = P.zac;                         // parabola constants
      const int Lf = Z.Lf, nout = L - Lf + 1, lt = Z.lt, f1 = Z.f1, ltp = Z.ltp;
      if (pass == 1) {  // restore y in B0 (recycled by pass 0)
        __syncthreads();
#pragma unroll
        for (int c = 0; c < SPT / 4; ++c)
          *reinterpret_cast<float4*>(&S.B0[sw(i0t + 4 * c)]) = make_float4(yr[4 * c], yr[4 * c + 1], yr[4 * c + 2], yr[4 * c + 3]);
      }
      // ---- step A0: Dp -> B1 (thread-blocked)
      {
        float run = 0.f;
#pragma unroll
        for (int j = 0; j < SPT; ++j) { loc[j] = run; run += yr[j]; }
        const double t_off = block_exscan_f64((double)run, S.red, nullptr);  // barriers: B1 (g) reads done, B0 restored
        const float y0 = S.B0[0];
#pragma unroll
        for (int c = 0; c < SPT / 4; ++c) {
          float4 v;
          float* pv = &v.x;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = 4 * c + e;
            const float Ti = (float)(t_off + (double)loc[j]);
            pv[e] = (yr[j] - y0) + Z.eps * Ti;
          }
          *reinterpret_cast<float4*>(&S.B1[sw(i0t + 4 * c)]) = v;
        }
      }
      __syncthreads();
      // ---- step A1 (lane-strided): flat top + last tap; ZAC: u[n] -> B0 in place
      float ac[SPT], dz[SPT];
      {
        const float dwl = want_c ? (ZZ.w_last - Z.w_last) : 0.f;  // shared pass: ZAC last tap relative to CUSP's
        const float wl = want_c ? Z.w_last : ZZ.w_last;
        for (int m = 0; m < SPT; ++m) {
          const int k = tid + NT * m;
          float a = 0.f, z = 0.f;
          const float yk = S.B0[sw(k)];
          if (k < nout) {
            const int n = k + Lf - 1;
            a = Z.sc * (S.B1[sw(n - lt)] - S.B1[sw(n - f1)]) + wl * yk;
            z = dwl * yk;
          }
          ac[m] = a; dz[m] = z;
          if (want_z) {
            float u = 0.f;
            for (int e = 0; e < ZZ.zu_n; ++e) {
              const int i = k - ZZ.zu_shift[e];
              if (i > 0 && k < L) u = fmaf(ZZ.zu_coef[e], S.B1[sw(i)], u);
            }
            S.B0[sw(k)] = u;  // same index this thread just read: in-place is race-free
          }
        }
      }
      if (want_z) {
        // ---- step A2 (thread-blocked): PRF = cumsum(cumsum(u)) in f64, two sweeps
        __syncthreads();
        float ur[SPT];
#pragma unroll
        for (int c = 0; c < SPT / 4; ++c) {
          float4 v = *reinterpret_cast<const float4*>(&S.B0[sw(i0t + 4 * c)]);
          ur[4 * c] = v.x; ur[4 * c + 1] = v.y; ur[4 * c + 2] = v.z; ur[4 * c + 3] = v.w;
        }
        double c1 = 0, V = 0;
#pragma unroll
        for (int j = 0; j < SPT; ++j) { c1 += (double)ur[j]; V += c1; }
        const double O1 = block_exscan_f64(c1, S.red, nullptr);
        const double O2 = block_exscan_f64((double)SPT * O1 + V, S.red, nullptr);
        c1 = O1;
        double c2 = O2;
        const double mrho = -(double)ZZ.rho_sc;
#pragma unroll
        for (int c = 0; c < SPT / 4; ++c) {
          float4 v;
          float* pv = &v.x;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            c1 += (double)ur[4 * c + e];
            c2 += c1;
            pv[e] = (float)(mrho * c2);
          }
          *reinterpret_cast<float4*>(&S.B0[sw(i0t + 4 * c)]) = v;
        }
        __syncthreads();
        for (int m = 0; m < SPT; ++m) {
          const int k = tid + NT * m;
          if (k < nout) dz[m] += S.B0[sw(k + Lf - 1)];
        }
      }
      // ---- step B: causal one-pole G -> B1, rise(-) and fall(+) exponentials
      // d[i] = (y[i]-y[i-1]) + eps*y[i-1] for 1 <= i < L, else 0
      float dr[SPT];
#pragma unroll
      for (int j = 0; j < SPT; ++j) {
        const float yp = (j == 0) ? yprev : yr[(j - 1) & (SPT - 1)];
        const int i = i0t + j;
        dr[j] = (i >= 1 && i < L) ? (yr[j] - yp) + Z.eps * yp : 0.f;
      }
      {
        float g = 0.f;
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
          const float t = fmaf(Z.q_hi, g, dr[j]);
          g = fmaf(Z.q_lo, g, t);
          loc[j] = g;
        }
        const float carry = block_exscan_affine(Z.a32_hi, g, reinterpret_cast<float*>(S.red));  // barriers: A1 reads of B1 done
#pragma unroll
        for (int c = 0; c < SPT / 4; ++c) {
          float4 v;
          float* pv = &v.x;
#pragma unroll
          for (int e = 0; e < 4; ++e) pv[e] = fmaf(Z.qpow[4 * c + e + 1], carry, loc[4 * c + e]);
          *reinterpret_cast<float4*>(&S.B1[sw(i0t + 4 * c)]) = v;
        }
      }
      __syncthreads();
      for (int m = 0; m < SPT; ++m) {
        const int k = tid + NT * m;
        if (k < nout) {
          const int n = k + Lf - 1;
          const float pm = S.B1[sw(n)] - Z.q_lt * S.B1[sw(n - lt)];
          const float fp = Z.q_mltp * (S.B1[sw(k + ltp - 1)] - Z.q_ltp1 * S.B1[sw(k)]);
          ac[m] = fmaf(Z.sc_half_den, fp - pm, ac[m]);
        }
      }
      // ---- step C: anti-causal one-pole A -> B1, rise(+) and fall(-) exponentials
      {
        float a = 0.f;
#pragma unroll
        for (int j = SPT - 1; j >= 0; --j) {
          const float t = fmaf(Z.q_hi, a, dr[j]);
          a = fmaf(Z.q_lo, a, t);
          loc[j] = a;
        }
        const float carry = block_exscan_affine_rev(Z.a32_hi, a, reinterpret_cast<float*>(S.red));  // barriers: G reads done
#pragma unroll
        for (int c = 0; c < SPT / 4; ++c) {
          float4 v;
          float* pv = &v.x;
#pragma unroll
          for (int e = 0; e < 4; ++e) pv[e] = fmaf(Z.qpow[SPT - (4 * c + e)], carry, loc[4 * c + e]);
          *reinterpret_cast<float4*>(&S.B1[sw(i0t + 4 * c)]) = v;
        }
        if (tid == 0) S.B1[sw(Lp)] = 0.f;  // A[L] when L == Lp
      }
      __syncthreads();
      for (int m = 0; m < SPT; ++m) {
        const int k = tid + NT * m;
        if (k < nout) {
          const int n = k + Lf - 1;
          const float pp = Z.q_mlt1 * S.B1[sw(n - lt + 1)] - Z.q1 * S.B1[sw(n + 1)];
          const float fm = Z.q2 * (S.B1[sw(k + 1)] - Z.q_ltp1 * S.B1[sw(k + ltp)]);
          ac[m] = fmaf(Z.sc_half_den, pp - fm, ac[m]);
        }
      }
      if (want_c) finish(0, Lf, ac);
      if (want_z) {
#pragma unroll
        for (int m = 0; m < SPT; ++m) dz[m] += ac[m];
        finish(1, Lf, dz);
      }
    }
  }

  // ---------------------------------------------------------------- outputs
  if (tid == 0) {
    float* o = S.outv;
    o[C_blmean] = blmean; o[C_blsigma] = blsigma; o[C_blslope] = blslope; o[C_bloffset] = bloffset;
    o[C_tailmean] = tailmean; o[C_tailsigma] = tailsigma; o[C_tailslope] = tailslope; o[C_tailoffset] = tailoffset;
    o[C_t0] = t0_us; o[C_t10] = ttx[0]; o[C_t50] = ttx[1]; o[C_t80] = ttx[2]; o[C_t90] = ttx[3]; o[C_t99] = ttx[4];
    o[C_t50_current] = t50cur_us;
    o[C_drift_time] = (ttx[3] - t0_us) * P.unit_per_us;
    o[C_tail_tau] = tail_tau; o[C_tail_mean] = tail_mean; o[C_tail_sigma] = tail_sigma;
    o[C_e_max] = e_max; o[C_e_min] = e_min;
    o[C_e_10410] = mx_f[0]; o[C_e_535] = mx_f[1]; o[C_e_313] = mx_f[2];
    o[C_e_10410_inv] = -mn_f[0]; o[C_e_313_inv] = -mn_f[2];  // trap(-y) = -trap(y)  (dsp_icpc.jl:199-204)
    o[C_t0_inv] = t0inv_us;
    o[C_e_trap] = e_trap; o[C_e_cusp] = e_cz[0]; o[C_e_zac] = e_cz[1];
    o[C_e_trap_max] = mx_opt.v; o[C_e_cusp_max] = mx_cz[0]; o[C_e_zac_max] = mx_cz[1];
    o[C_t_trap_max] = P.t_first + P.dt * (float)(mx_opt.i + P.opt.flen - 1);
    o[C_t_cusp_max] = tmx_cz[0]; o[C_t_zac_max] = tmx_cz[1];
    o[C_qdrift] = qdrift; o[C_lq] = lq;
    o[C_a_sg] = a_cur[0]; o[C_a_60] = a_cur[1]; o[C_a_100] = a_cur[2]; o[C_a_raw] = a_cur[3];
    o[C_inTrace_intersect] = intr_x;
    o[C_inTrace_n] = __int_as_float(intr_n);
    o[C_n_sat_low] = __int_as_float(n_low); o[C_n_sat_high] = __int_as_float(n_high);
    o[C_n_sat_low_cons] = __int_as_float(cons_low); o[C_n_sat_high_cons] = __int_as_float(cons_high);
  }
  __syncthreads();
  if (tid < C_NCOLS) {
    float* dst = reinterpret_cast<float*>(out.col[tid]);
    if (dst) dst[(size_t)blockIdx.x * (size_t)out.stride] = S.outv[tid];
  }
}

// ---------------------------------------------------------------------------
// BASELINE config 2: blmean -> shift -> InvCR -> Trap(10us,4us) -> maximum
// (reference src/dsp_icpc.jl:102-105,119-120,147-148).  Same staging and scans
// as the fused kernel, nothing else: 4L+8 algorithmic bytes per trace.
template <int NT_MAX>
__global__ void __launch_bounds__(NT_MAX)
pz_trap_kernel(const float* __restrict__ wf, const IcpcDev* __restrict__ Pp, float* __restrict__ o_blmean,
               float* __restrict__ o_e10410) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const IcpcDev& P = *Pp;
  const int L = P.L, NT = blockDim.x, tid = threadIdx.x;
  const int Lp = NT * SPT;
  float* B = reinterpret_cast<float*>(smem_raw);             // [Lp+32]
  double* red = reinterpret_cast<double*>(B + Lp + 32);        // [MAX_WAVES*2]
  const float* w = wf + (size_t)blockIdx.x * (size_t)L;
  if ((L & 3) == 0) {
    const float4* w4 = reinterpret_cast<const float4*>(w);
#pragma unroll
    for (int m = 0; m < SPT / 4; ++m) {
      int i4 = tid + NT * m;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (4 * i4 < L) v = w4[i4];
      *reinterpret_cast<float4*>(&B[sw(4 * i4)]) = v;
    }
  } else {
    for (int m = 0; m < SPT; ++m) {
      int i = tid + NT * m;
      B[sw(i)] = (i < L) ? w[i] : 0.f;
    }
  }
  if (tid < 32) B[Lp + tid] = 0.f;
  __syncthreads();
  float xr[SPT], loc[SPT];
  const int i0t = SPT * tid;
#pragma unroll
  for (int c = 0; c < SPT / 4; ++c) {
    float4 v = *reinterpret_cast<const float4*>(&B[sw(i0t + 4 * c)]);
    xr[4 * c] = v.x; xr[4 * c + 1] = v.y; xr[4 * c + 2] = v.z; xr[4 * c + 3] = v.w;
  }
  const int nv = max(0, min(SPT, L - i0t));
  double s1 = 0;
  if ((i0t <= P.bl.until) && (i0t + SPT - 1 >= P.bl.from)) {
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      int i = i0t + j;
      if (i >= P.bl.from && i <= P.bl.until) s1 += (double)xr[j];
    }
  }
  {
    double v[1] = {s1};
    block_reduce<1>(v, red, OpSum());
    s1 = v[0];
  }
  const float blmean = (float)(s1 * P.bl.inv_n);
  float run = 0.f;
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    xr[j] = (j < nv) ? xr[j] - blmean : 0.f;
    run += xr[j];
    loc[j] = run;
  }
  const double s_off = block_exscan_f64((double)run, red, nullptr);
  const float coff = (float)(P.pz_c64 * s_off);
  run = 0.f;
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    float y = (j < nv) ? (xr[j] + coff) + P.pz_c * loc[j] : 0.f;
    loc[j] = run;  // exclusive prefix of y
    run += y;
  }
  double tot;
  const double t_off = block_exscan_f64((double)run, red, &tot);
  // the barriers inside the scans order these writes after every read of the raw trace
#pragma unroll
  for (int c = 0; c < SPT / 4; ++c) {
    float4 vt;
    vt.x = (float)(t_off + (double)loc[4 * c]);
    vt.y = (float)(t_off + (double)loc[4 * c + 1]);
    vt.z = (float)(t_off + (double)loc[4 * c + 2]);
    vt.w = (float)(t_off + (double)loc[4 * c + 3]);
    *reinterpret_cast<float4*>(&B[sw(i0t + 4 * c)]) = vt;
  }
  if (tid == NT - 1) B[sw(Lp)] = (float)tot;
  __syncthreads();
  const TrapDev tr = P.fixed[0];
  const int nout = L - tr.flen + 1;
  float mx = -INFINITY;
  for (int m = 0; m < SPT; ++m) {
    const int k = tid + NT * m;
    if (k < nout) mx = fmaxf(mx, trap_at(B, k, tr));
  }
  {
    float v[1] = {mx};
    block_reduce<1>(v, reinterpret_cast<float*>(red), OpMax());
    mx = v[0];
  }
  if (tid == 0) {
    o_blmean[blockIdx.x] = blmean;
    o_e10410[blockIdx.x] = mx;
  }
}

hipError_t launch_pz_trap(const float* wf, int64_t n, int NT, const IcpcDev* dP, float* blmean, float* e10410, hipStream_t st) {
  size_t smem = ((size_t)NT * SPT + 32) * 4 + MAX_WAVES * 2 * 8;
  dim3 grid((unsigned)n), block((unsigned)NT);
  if (NT <= 256) hipLaunchKernelGGL(pz_trap_kernel<256>, grid, block, smem, st, wf, dP, blmean, e10410);
  else if (NT <= 512) hipLaunchKernelGGL(pz_trap_kernel<512>, grid, block, smem, st, wf, dP, blmean, e10410);
  else {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pz_trap_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pz_trap_kernel<1024>, grid, block, smem, st, wf, dP, blmean, e10410);
  }
  return hipGetLastError();
}

size_t icpc_smem_bytes(int NT) {
  size_t Lp = (size_t)NT * SPT;
  return (Lp + Lp + 32) * 4 + (size_t)NMASK * NT * 4 + MAX_WAVES * 8 * 8 + C_NCOLS * 4 + 64 * 4;
}

hipError_t launch_icpc(const float* wf, int64_t n, int NT, const IcpcDev* dP, const IcpcOutDev& out, hipStream_t st) {
  size_t smem = icpc_smem_bytes(NT);
  dim3 grid((unsigned)n), block((unsigned)NT);
  hipError_t e;
  if (NT <= 256) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&icpc_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(icpc_kernel<256>, grid, block, smem, st, wf, dP, out);
  } else if (NT <= 512) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&icpc_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(icpc_kernel<512>, grid, block, smem, st, wf, dP, out);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&icpc_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(icpc_kernel<1024>, grid, block, smem, st, wf, dP, out);
  }
  return hipGetLastError();
}

}  // namespace ldsp
