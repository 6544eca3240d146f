// sipm_kernel.hip — fused dsp_sipm for gfx950 (reference src/dsp_sipm.jl:47-158).
//
// One workgroup per trace; the trace is read from HBM once and every column is
// produced from two LDS arrays:
//   A: x  -> cumsum(I) -> P = InvCR(I) -> cumsum(P)
//   B: g = SG'(x) -> I = cumsum(g) (-I serves the discharge scans) -> trapezoid output
// MAD thresholds by radix select on monotone float keys, triggers by ballot bit-masks
// (IntersectMaximum: ragged outputs as fixed-capacity slabs + counts).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <vector>

#include "host_math.hpp"
#include "ldsp_ctx.hpp"
#include "trace_blocks.hpp"

namespace ldsp {
namespace sipm {

using tb::Scratch;

struct SipmDev {
  int32_t L, np;
  float t_first, dt, inv_upus;
  double t_first64, dt64;   // the time axis as given (trigger positions are composed in double)
  int32_t trunc_from, trunc_until;
  float sg_c[LDSP_MAX_SG_PTS];  // correlation taps
  float sg_cs[LDSP_MAX_SG_PTS]; // their suffix sums, cs[m] = sum_{i >= m} c[i]: the taps of the integrated filter (k_sipm_s4)
  int32_t sg_mintot, sg_maxtot;
  float sg_min_thr, sg_max_thr, sg_nsigma, sg_min_dc, sg_max_dc, sg_nsigma_dc;
  float pz_c;
  ldsp_trap trap;
  int32_t trap_mintot, trap_maxtot;
  float trap_min_thr, trap_max_thr, trap_nsigma, trap_min_dc, trap_max_dc, trap_nsigma_dc;
  int32_t dbg_stop;  // profiling aid: return after stage k (tools/gpu_time_sipm.py)
  int32_t in_u16;    // the traces are uint16 ADC counts (ldsp_sipm_run_u16): converted as they are loaded
  long long* dbg_stamps;   // diagnostic build (-DLDSP_STAMPS, tools/dev_build_sipm.sh): s_memtime per wave at every SSTAMP
};

// SSTAMP(): the wave's next time stamp (slot = a per-wave counter in LDS) -> dbg_stamps[(block*16 + wave)*64 + slot]; compiled
// in only with -DLDSP_STAMPS (tools/stamp_map_sipm.py reads the timeline)
#ifdef LDSP_STAMPS
__device__ __forceinline__ void s4_stamp(long long* st, bool init = false) {
  __shared__ int wcnt[16];
  if (init) { if (threadIdx.x < 16) wcnt[threadIdx.x] = 0; __syncthreads(); }
  if (st && (threadIdx.x & 63) == 0 && blockIdx.x < 1024) {
    const int w = threadIdx.x >> 6, k = wcnt[w];
    wcnt[w] = k + 1;
    if (k < 64) st[((size_t)blockIdx.x * 16 + w) * 64 + k] = (long long)__builtin_amdgcn_s_memtime();
  }
}
#define SSTAMP_INIT(P) ::ldsp::sipm::s4_stamp((P).dbg_stamps, true)
#define SSTAMP(P) ::ldsp::sipm::s4_stamp((P).dbg_stamps)
#else
#define SSTAMP_INIT(P) do { } while (0)
#define SSTAMP(P) do { } while (0)
#endif

enum { S_t_max, S_t_min, S_t_max_lar, S_t_min_lar, S_e_max, S_e_min, S_e_max_lar, S_e_min_lar,
       S_blmean, S_blsigma, S_blslope, S_bloffset, S_wfmean, S_wfsigma, S_wfslope, S_wfoffset,
       S_threshold, S_threshold_DC, S_threshold_trap, S_threshold_DC_trap, S_NCOLS };

struct SipmOutDev {
  float* col[S_NCOLS];
  ldsp_trig_out trig[4];  // SG, DC, trap, DC_trap
};

__device__ __forceinline__ int pad4(int n) { return ((n + 3) & ~3) + 64; }

#include "intersect_maximum_block.inc"

__global__ void __launch_bounds__(1024) k_sipm(const float* __restrict__ wf, SipmDev P, SipmOutDev out) {
  extern __shared__ __align__(16) unsigned char raw[];
  const int L = P.L, tid = threadIdx.x, NT = blockDim.x;
  float* A = reinterpret_cast<float*>(raw);
  float* B = A + pad4(L);
  uint32_t* bm = reinterpret_cast<uint32_t*>(B + pad4(L));
  const int nwmax = ((L + 31) >> 5) + 2;
  // (offset arithmetic on integers, pointer derived from `raw`: keeps the LDS address space — see functor_kernels.hip)
  const size_t sc_off = ((size_t)(reinterpret_cast<unsigned char*>(bm + nwmax) - raw) + 15) & ~(size_t)15;
  Scratch& sc = *reinterpret_cast<Scratch*>(raw + sc_off);
  const size_t b = blockIdx.x;
  auto put = [&](int c, float v) { if (tid == 0 && out.col[c]) out.col[c][b] = v; };

  // shift_waveform(wvfs, 0.0)  (dsp_sipm.jl:88) — values unchanged
  if (P.in_u16) tb::load_trace_u16(reinterpret_cast<const uint16_t*>(wf) + b * (size_t)L, A, L);
  else tb::load_trace(wf + b * (size_t)L, A, L);
  for (int i = L + tid; i < pad4(L); i += NT) A[i] = 0.f;
  __syncthreads();
  {  // extremestats on the full trace and on TruncateFilter(t0_hpge_window)   :91-95
    float vmin, vmax; int imin, imax;
    tb::extreme_stats(A, 0, L - 1, sc, &vmin, &imin, &vmax, &imax);
    put(S_e_min, vmin); put(S_e_max, vmax);
    put(S_t_min, (P.t_first + P.dt * (float)imin) * P.inv_upus); put(S_t_max, (P.t_first + P.dt * (float)imax) * P.inv_upus);
    tb::extreme_stats(A, P.trunc_from, P.trunc_until, sc, &vmin, &imin, &vmax, &imax);
    put(S_e_min_lar, vmin); put(S_e_max_lar, vmax);
    put(S_t_min_lar, (P.t_first + P.dt * (float)imin) * P.inv_upus); put(S_t_max_lar, (P.t_first + P.dt * (float)imax) * P.inv_upus);
  }
  if (P.dbg_stop == 1) return;
  // SavitzkyGolayFilter(wl, degree, 1): g -> B   :99-100   (valid mode, trailing time axis, A1)
  const int np = P.np, ng = L - np + 1;
  const float tg = P.t_first + P.dt * (float)(np - 1);
  const double tg64 = fma(P.dt64, (double)(np - 1), P.t_first64);   // (trigger positions: double time axis)
  for (int k = tid; k < pad4(L); k += NT) {
    float g = 0.f;
    if (k < ng) for (int i = 0; i < np; ++i) g = fmaf(P.sg_c[i], A[k + i], g);
    B[k] = g;
  }
  __syncthreads();
  if (P.dbg_stop == 2) return;
  // SG triggers :103-105
  // (A = x is dead from here to the InvCR stage: it serves as the radix select's candidate buffer)
  uint32_t* cand = reinterpret_cast<uint32_t*>(A);
  const float thr_sg = tb::mad_threshold(B, ng, P.sg_min_thr, P.sg_max_thr, 1.f, sc, cand, pad4(L));
  put(S_threshold, thr_sg);
  if (P.dbg_stop == 3) return;
  {
    const float th = P.sg_nsigma * thr_sg;
    tb::build_mask(B, ng, 1.f, th, bm);
    __syncthreads();
    if (tid == 0) sc.f[1] = 0.f;  // minimum.(inters.x; init = 0)
    __syncthreads();
    const ldsp_trig_out& o = out.trig[0];
    const size_t off = b * (size_t)o.cap;
    const int tot = intersect_maximum_block(B, ng, 1.f, th, P.sg_mintot, P.sg_maxtot, tg64, P.dt64, bm, sc, o.cap,
                                            o.x ? o.x + off : nullptr, o.x_high ? o.x_high + off : nullptr,
                                            o.x_tot ? o.x_tot + off : nullptr, o.max ? o.max + off : nullptr, &sc.f[1]);
    if (tid == 0 && o.count) o.count[b] = tot;
  }
  __syncthreads();
  if (P.dbg_stop == 4) return;
  const float minx = fminf(sc.f[1], 0.f);
  // IntegratorFilter(gain = 1) on g   :108-109, telescoped as in k_sipm_s4 (sipm_s4.inc: no running sum of rounded values):
  //   I[k] = F(k) - F(-1),  F(k) = sum_{m=1}^{np-1} cs[m] (x[k+m] - x[0]),  cs[m] = sum_{i >= m} c[i];  the samples are read again into A
  if (P.in_u16) tb::load_trace_u16(reinterpret_cast<const uint16_t*>(wf) + b * (size_t)L, A, L);
  else tb::load_trace(wf + b * (size_t)L, A, L);
  for (int i = L + tid; i < pad4(L); i += NT) A[i] = 0.f;
  __syncthreads();
  {
    const float x0 = A[0];
    float f0 = 0.f;
    for (int m = 1; m < np; ++m) f0 = fmaf(P.sg_cs[m], A[m - 1] - x0, f0);
    for (int k = tid; k < pad4(L); k += NT) {
      float acc = -f0;
      if (k < ng) for (int m = 1; m < np; ++m) acc = fmaf(P.sg_cs[m], A[k + m] - x0, acc);
      B[k] = (k < ng) ? acc : 0.f;
    }
  }
  __syncthreads();
  {  // signalstats on the integrated trace :112-115 (the init = 0 quirk: SURVEY a2)
    const float time_min = tg, d3 = 3.f * P.dt;
    const float stop = (minx < time_min + d3) ? time_min + d3 : minx;
    const int from = (int)nearbyintf((time_min - tg) / P.dt), until = (int)nearbyintf((stop - tg) / P.dt);
    float m = NAN, sg = NAN, sl = NAN, of = NAN;
    if (0 <= from && from <= until && until <= ng - 1) tb::window_stats<false>(B, from, until, tg, P.dt, sc, &m, &sg, &sl, &of);
    put(S_blmean, m); put(S_blsigma, sg); put(S_blslope, sl); put(S_bloffset, of);
    tb::window_stats<false>(B, 0, ng - 1, tg, P.dt, sc, &m, &sg, &sl, &of);
    put(S_wfmean, m); put(S_wfsigma, sg); put(S_wfslope, sl); put(S_wfoffset, of);
  }
  if (P.dbg_stop == 5) return;
  // discharge detection on the flipped integrated trace (-I): SG and trap bounds, both with
  // the SG IntersectMaximum functor   :118-120, :137-138
  for (int v = 0; v < 2; ++v) {
    const float lo = v ? P.trap_min_dc : P.sg_min_dc, hi = v ? P.trap_max_dc : P.sg_max_dc;
    const float ns = v ? P.trap_nsigma_dc : P.sg_nsigma_dc;
    const float thr = tb::mad_threshold(B, ng, lo, hi, -1.f, sc, cand, pad4(L));
    put(v ? S_threshold_DC_trap : S_threshold_DC, thr);
    const float th = ns * thr;
    tb::build_mask(B, ng, -1.f, th, bm);
    __syncthreads();
    const ldsp_trig_out& o = out.trig[v ? 3 : 1];
    const size_t off = b * (size_t)o.cap;
    const int tot = intersect_maximum_block(B, ng, -1.f, th, P.sg_mintot, P.sg_maxtot, tg64, P.dt64, bm, sc, o.cap,
                                            o.x ? o.x + off : nullptr, o.x_high ? o.x_high + off : nullptr,
                                            o.x_tot ? o.x_tot + off : nullptr, o.max ? o.max + off : nullptr, nullptr);
    if (tid == 0 && o.count) o.count[b] = tot;
    __syncthreads();
  }
  if (P.dbg_stop == 6) return;
  // InvCRFilter(pz_tau) on I, then TrapezoidalChargeFilter(rt, ft)   :124-129
  const int flen = P.trap.navg + P.trap.ngap + P.trap.navg2, nt = ng - flen + 1;
  const float tt = tg + P.dt * (float)(flen - 1);
  const double tt64 = fma(P.dt64, (double)(flen - 1), tg64);
  if (P.trap.navg == P.trap.navg2) {
    // equal legs (what TrapezoidalChargeFilter(rt, ft) builds): difference-first as in k_sipm_s4 (sipm_s4.inc) — the pole-zero corrected
    // trace Pz = I + c cumsum(I) is never formed:  tr[k] = (1/n) sum_{j<n} E[k+j],  E[k] = Pz[k+o2] - Pz[k] = I[k+o2] - I[k] + c sum_{t=k+1}^{k+o2} I[t]
    const int o2 = P.trap.navg + P.trap.ngap, nn = P.trap.navg;
    const float inv = 1.f / (float)nn;
    for (int k = tid; k < pad4(L); k += NT) {
      float v = 0.f;
      if (k + o2 < ng) {
        float w = 0.f;
        for (int t = 1; t <= o2; ++t) w += B[k + t];
        v = fmaf(P.pz_c, w, B[k + o2] - B[k]);
      }
      A[k] = v;   // E
    }
    __syncthreads();
    for (int k = tid; k < pad4(L); k += NT) {
      float v = 0.f;
      if (k < nt) {
        for (int j = 0; j < nn; ++j) v += A[k + j];
        v *= inv;
      }
      B[k] = v;
    }
  } else {
    for (int i = tid; i < pad4(L); i += NT) A[i] = (i < ng) ? B[i] : 0.f;
    __syncthreads();
    tb::prefix_sum_inplace(A, ng, sc);
    __syncthreads();
    for (int i = tid; i < pad4(L); i += NT) A[i] = (i < ng) ? B[i] + P.pz_c * A[i] : 0.f;  // P
    __syncthreads();
    {
      const float i1 = 1.f / (float)P.trap.navg, i2 = 1.f / (float)P.trap.navg2;
      // SiPM shaping times are a few samples: sum the windows directly (a difference of float
      // prefix sums would carry ulp(cumsum) / navg of error into the MAD threshold)
      const bool direct = P.trap.navg <= 16 && P.trap.navg2 <= 16;
      if (!direct) {
        tb::prefix_sum_inplace(A, ng, sc);  // S = cumsum(P)
        __syncthreads();
      }
      for (int k = tid; k < pad4(L); k += NT) {
        float v = 0.f;
        if (k < nt) {
          if (direct) {
            float a = 0.f, bb = 0.f;
            for (int j = 0; j < P.trap.navg2; ++j) a += A[k + P.trap.navg + P.trap.ngap + j];
            for (int j = 0; j < P.trap.navg; ++j) bb += A[k + j];
            v = a * i2 - bb * i1;
          } else {
            const float s0 = (k > 0) ? A[k - 1] : 0.f;
            v = (A[k + flen - 1] - A[k + P.trap.navg + P.trap.ngap - 1]) * i2 - (A[k + P.trap.navg - 1] - s0) * i1;
          }
        }
        B[k] = v;
      }
    }
  }
  __syncthreads();
  if (P.dbg_stop == 7) return;
  // trap triggers :132-134
  const float thr_t = tb::mad_threshold(B, nt, P.trap_min_thr, P.trap_max_thr, 1.f, sc, cand, pad4(L));  // A = P is dead
  put(S_threshold_trap, thr_t);
  {
    const float th = P.trap_nsigma * thr_t;
    tb::build_mask(B, nt, 1.f, th, bm);
    __syncthreads();
    const ldsp_trig_out& o = out.trig[2];
    const size_t off = b * (size_t)o.cap;
    const int tot = intersect_maximum_block(B, nt, 1.f, th, P.trap_mintot, P.trap_maxtot, tt64, P.dt64, bm, sc, o.cap,
                                            o.x ? o.x + off : nullptr, o.x_high ? o.x_high + off : nullptr,
                                            o.x_tot ? o.x_tot + off : nullptr, o.max ? o.max + off : nullptr, nullptr);
    if (tid == 0 && o.count) o.count[b] = tot;
  }
}

#include "sipm_s4.inc"

template <int NT, int R, bool FULL>
static hipError_t launch_s4(const float* wf, int64_t n, const SipmDev* d, const SipmOutDev& od, hipStream_t st) {
  const size_t bytes = S4Lds<NT, R>::bytes();
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sipm_s4<NT, R, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_sipm_s4<NT, R, FULL>), dim3((unsigned)n), dim3(NT), bytes, st, wf, d, od);
  return hipGetLastError();
}

}  // namespace sipm
}  // namespace ldsp

using namespace ldsp;

static int sipm_run_impl(ldsp_ctx* c, const float* wf, int64_t n, const ldsp_sipm_params* p, const ldsp_sipm_out* out, int in_u16);
extern "C" int ldsp_sipm_run(ldsp_ctx* c, const float* wf, int64_t n, const ldsp_sipm_params* p, const ldsp_sipm_out* out) {
  return sipm_run_impl(c, wf, n, p, out, 0);
}
extern "C" int ldsp_sipm_run_u16(ldsp_ctx* c, const uint16_t* wf, int64_t n, const ldsp_sipm_params* p, const ldsp_sipm_out* out) {
  return sipm_run_impl(c, reinterpret_cast<const float*>(wf), n, p, out, 1);
}
static int sipm_run_impl(ldsp_ctx* c, const float* wf, int64_t n, const ldsp_sipm_params* p, const ldsp_sipm_out* out, int in_u16) {
  if (!p || !out) return ldsp_fail(LDSP_ERR_INVALID_ARG, "ldsp_sipm_run: NULL argument");
  int rc = ldsp_check_batch(c, wf, n, p->L, "ldsp_sipm_run");
  if (rc || n == 0) return rc;
  const int L = p->L;
  if (!(p->dt > 0) || !(p->unit_per_us > 0)) return ldsp_fail(LDSP_ERR_INVALID_ARG, "dt and unit_per_us must be positive");
  if (p->sg_npts < 1 || p->sg_npts > LDSP_MAX_SG_PTS || (p->sg_npts & 1) == 0 || p->sg_npts <= p->sg_degree || p->sg_npts > L)
    return ldsp_fail(LDSP_ERR_UNSUPPORTED, "Savitzky-Golay window of %d points (degree %d) unsupported", p->sg_npts, p->sg_degree);
  const int ng = L - p->sg_npts + 1;
  const int flen = p->trap.navg + p->trap.ngap + p->trap.navg2;
  if (p->trap.navg < 1 || p->trap.navg2 < 1 || p->trap.ngap < 0 || flen > ng) return ldsp_fail(LDSP_ERR_WINDOW, "trapezoid does not fit");
  if (!(0 <= p->trunc_from && p->trunc_from <= p->trunc_until && p->trunc_until <= L - 1)) return ldsp_fail(LDSP_ERR_WINDOW, "t0_hpge_window outside the trace");
  if (p->sg_mintot < 1 || p->sg_maxtot < 1 || p->trap_mintot < 1 || p->trap_maxtot < 1) return ldsp_fail(LDSP_ERR_INVALID_ARG, "tot values must be >= 1 sample");
  sipm::SipmDev d;
  memset(&d, 0, sizeof d);
  d.L = L; d.np = p->sg_npts;
  d.t_first = (float)p->t_first; d.dt = (float)p->dt; d.inv_upus = (float)(1.0 / p->unit_per_us);
  d.t_first64 = p->t_first; d.dt64 = p->dt;
  d.trunc_from = p->trunc_from; d.trunc_until = p->trunc_until;
  std::vector<double> cc;
  if (!hm::sg_corr_coeffs(p->sg_npts, p->sg_degree, 1, cc)) return ldsp_fail(LDSP_ERR_INVALID_ARG, "Savitzky-Golay coefficients");
  for (int i = 0; i < p->sg_npts; ++i) d.sg_c[i] = (float)cc[i];
  { double suf = 0.0; for (int i = p->sg_npts - 1; i >= 0; --i) { suf += cc[i]; d.sg_cs[i] = (float)suf; } }
  d.sg_mintot = p->sg_mintot; d.sg_maxtot = p->sg_maxtot;
  d.sg_min_thr = (float)p->sg_min_thr; d.sg_max_thr = (float)p->sg_max_thr; d.sg_nsigma = (float)p->sg_nsigma;
  d.sg_min_dc = (float)p->sg_min_dc_thr; d.sg_max_dc = (float)p->sg_max_dc_thr; d.sg_nsigma_dc = (float)p->sg_nsigma_dc;
  d.pz_c = (float)p->pz_c; d.trap = p->trap;
  d.trap_mintot = p->trap_mintot; d.trap_maxtot = p->trap_maxtot;
  d.trap_min_thr = (float)p->trap_min_thr; d.trap_max_thr = (float)p->trap_max_thr; d.trap_nsigma = (float)p->trap_nsigma;
  d.dbg_stop = c->dbg_stop;
  d.in_u16 = in_u16;
  d.dbg_stamps = c->dbg_stamps;
  d.trap_min_dc = (float)p->trap_min_dc_thr; d.trap_max_dc = (float)p->trap_max_dc_thr; d.trap_nsigma_dc = (float)p->trap_nsigma_dc;
  sipm::SipmOutDev od;
  static_assert(sizeof(ldsp_sipm_out) == sizeof(void*) * sipm::S_NCOLS + 4 * sizeof(ldsp_trig_out), "ldsp_sipm_out layout");
  memcpy(od.col, out, sizeof(void*) * sipm::S_NCOLS);
  od.trig[0] = out->trig; od.trig[1] = out->trig_DC; od.trig[2] = out->trig_trap; od.trig[3] = out->trig_DC_trap;
  for (auto& t : od.trig) {
    if (t.cap < 0) return ldsp_fail(LDSP_ERR_INVALID_ARG, "ldsp_trig_out.cap must be >= 0");
    if (t.cap == 0) t.cap = LDSP_MAX_TRIG;
  }
  if (c->timing) HIP_TRY(hipEventRecord(c->ev0, c->stream));
  // register-resident kernel (one LDS copy of the trace, two traces per CU) when the trace fills a tile and the
  // filters are the usual short ones; otherwise the generic two-array kernel
  const bool s4_ok = !c->sipm_generic && p->sg_npts <= sipm::S4_SG_MAX && p->trap.navg <= sipm::S4_LEG_MAX && p->trap.navg2 <= sipm::S4_LEG_MAX;
  hipError_t e = hipErrorInvalidValue;
  bool launched = false;
  if (s4_ok) {
    // smallest tile that holds the trace: 32 samples per thread, 64 .. 512 threads
    // the parameter block travels through device memory (scalar loads where a value is needed instead of ~60 SGPRs held for
    // the whole kernel); uploaded when it differs from the last call's
    if (!c->d_sipm) HIP_TRY(hipMalloc(&c->d_sipm, sizeof(sipm::SipmDev)));
    if (c->sipm_last.size() != sizeof d || memcmp(c->sipm_last.data(), &d, sizeof d)) {
      HIP_TRY(hipMemcpyAsync(c->d_sipm, &d, sizeof d, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));   // d dies at return
      c->sipm_last.assign(reinterpret_cast<const unsigned char*>(&d), reinterpret_cast<const unsigned char*>(&d) + sizeof d);
    }
    const sipm::SipmDev* dd = reinterpret_cast<const sipm::SipmDev*>(c->d_sipm);
    // FULL: the trace fills the tile AND the filters shorten it by at most one lane-strided row (N samples), so that only the
    // last registers of a thread can lie behind the end of a filtered signal (s4_valid)
    const bool tail_ok = (p->sg_npts - 1) + (flen - 1) <= 64;
    // a trace that does not fill the 32-samples-per-thread tile takes the smallest number of rows that holds it (round 4: 5 .. 8 rows of
    // 4 N samples — 12 000 samples run 6 rows of 2048 instead of 8: a quarter less of everything)
#define LDSP_S4R(N, RR) sipm::launch_s4<N, RR, false>(wf, n, dd, od, c->stream)
#define LDSP_S4(N) (L == 32 * N && tail_ok ? sipm::launch_s4<N, 8, true>(wf, n, dd, od, c->stream) \
                    : (L <= 20 * N ? LDSP_S4R(N, 5) : L <= 24 * N ? LDSP_S4R(N, 6) : L <= 28 * N ? LDSP_S4R(N, 7) : LDSP_S4R(N, 8)))
    launched = true;
    if (L <= 2048) e = LDSP_S4(64);
    else if (L <= 4096) e = LDSP_S4(128);
    else if (L <= 8192) e = LDSP_S4(256);
    else if (L <= 16384) e = LDSP_S4(512);
    else launched = false;
    if (launched) c->last_kernel = "sipm::k_sipm_s4";
#undef LDSP_S4
#undef LDSP_S4R
  }
  if (!launched) {
    const size_t p4 = (size_t)(((L + 3) & ~3) + 64);
    const size_t bytes = 2 * p4 * 4 + (size_t)(((L + 31) >> 5) + 2) * 4 + 16 + sizeof(tb::Scratch);
    if (bytes > 160 * 1024) return ldsp_fail(LDSP_ERR_UNSUPPORTED, "dsp_sipm keeps two arrays of the trace in LDS: L <= ~19000 (got %d)", L);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&sipm::k_sipm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    const int nt = L <= 4096 ? 256 : (L <= 8192 ? 512 : 1024);
    hipLaunchKernelGGL(sipm::k_sipm, dim3((unsigned)n), dim3(nt), bytes, c->stream, wf, d, od);
    c->last_kernel = "sipm::k_sipm";
    e = hipGetLastError();
  }
  if (e != hipSuccess) return ldsp_fail(LDSP_ERR_HIP, "launch: %s", hipGetErrorString(e));
  if (c->timing) { HIP_TRY(hipEventRecord(c->ev1, c->stream)); c->n_launches = 1; c->n_stages = 1; }
  return LDSP_OK;
}
