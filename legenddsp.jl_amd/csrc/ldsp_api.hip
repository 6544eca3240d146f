// ldsp_api.hip — the C ABI of include/ldsp.h: context, host-side lowering of the
// parameter blocks to device constants, launches.  No torch types, no CPU
// compute fallback: every entry point either launches HIP kernels or fails.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ldsp.h"
#include "host_math.hpp"
#include "icpc_dev.hpp"
#include "ldsp_ctx.hpp"

namespace ldsp {
hipError_t launch_icpc(const float* wf, int64_t n, int NT, int R, bool full, const IcpcDev* dP, float* aux, const IcpcOutDev& out,
                       const float* ext_bl, float ext_bl_scale, bool direct, bool cz_shared, bool fuse_ok, int stop_after_main, int cz_pad_floats, hipStream_t st, hipEvent_t mid,
                       int* stages);
hipError_t launch_icpc_lean3(const float* wf, int64_t n, int NT, int sg_slots, bool cz_shared, bool full, const IcpcDev* dP, const IcpcOutDev& out,
                             const float* ext_bl, float ext_bl_scale, int Lf, bool skip_cz, hipStream_t st);
size_t icpc_lean3_smem_bytes(int NT, int Lf);
extern int g_dbg_lds_pad;
hipError_t launch_pz_trap_lean(const float* wf, int64_t n, int NT, bool u16, const IcpcDev* dP, float* blmean, float* e10410, hipStream_t st);
hipError_t launch_pz_trap(const float* wf, int64_t n, int NT, bool full, const IcpcDev* dP, float* blmean, float* e10410, hipStream_t st);
size_t icpc_smem_bytes(int NT);
hipError_t launch_trap_grid(const float* wf, int64_t n, int NT, bool full, const TrapGridDev* dP, float* out, hipStream_t st);
hipError_t launch_sg_grid(const float* wf, int64_t n, int NT, bool full, const SgGridDev* dP, float* amax, float* energy, float* t50, float* blm,
                          float* bls, hipStream_t st);
hipError_t launch_fir_grid(const float* wf, int64_t n, int NT, bool full, const FirGridDev* dP, float* out, hipStream_t st);
}  // namespace ldsp

using namespace ldsp;

static thread_local std::string g_err;
int ldsp_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define fail ldsp_fail

int ldsp_check_batch_impl(ldsp_ctx* c, const void* x, int64_t n, int32_t L, const char* who) {
  if (!c) return fail(LDSP_ERR_INVALID_ARG, "%s: ctx is NULL", who);
  if (n < 0 || n > 0x7fffffffLL) return fail(LDSP_ERR_INVALID_ARG, "%s: n = %lld out of range", who, (long long)n);
  if (L < 1 || L > LDSP_MAX_L) return fail(LDSP_ERR_UNSUPPORTED, "%s: trace length %d outside [1, %d]", who, L, LDSP_MAX_L);
  if (n > 0 && !x) return fail(LDSP_ERR_INVALID_ARG, "%s: waveform pointer is NULL", who);
  // an error another library left behind on this thread (e.g. a failed pointer-attribute probe of the host program) must
  // not be reported as the failure of the launch below, which reads hipGetLastError() after it
  (void)hipGetLastError();
  return LDSP_OK;
}

extern "C" {

int ldsp_abi_version(void) { return LDSP_ABI_VERSION; }

int64_t ldsp_abi_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(ldsp_icpc_params);
    case 1: return sizeof(ldsp_icpc_out);
    case 2: return sizeof(ldsp_sipm_params);
    case 3: return sizeof(ldsp_sipm_out);
    case 4: return sizeof(ldsp_trig_out);
    case 5: return sizeof(ldsp_icpc_opts);
    default: return -1;
  }
}

const char* ldsp_last_error_string(void) { return g_err.c_str(); }

int ldsp_ctx_destroy(ldsp_ctx* c);
static int ctx_build(ldsp_ctx* c) {
  HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  HIP_TRY(hipMalloc(&c->d_icpc, sizeof(IcpcDev)));
  HIP_TRY(hipMalloc(&c->d_hc, sizeof(float) * LDSP_MAX_FIR_TAPS));
  HIP_TRY(hipMalloc(&c->d_hz, sizeof(float) * LDSP_MAX_FIR_TAPS));
  HIP_TRY(hipMalloc(&c->d_icpc_b, sizeof(IcpcDev)));
  HIP_TRY(hipMalloc(&c->d_hc_b, sizeof(float) * LDSP_MAX_FIR_TAPS));
  HIP_TRY(hipMalloc(&c->d_hz_b, sizeof(float) * LDSP_MAX_FIR_TAPS));
  HIP_TRY(hipMalloc(&c->d_coef, sizeof(float) * LDSP_MAX_FIR_TAPS));
  HIP_TRY(hipEventCreate(&c->ev0));
  HIP_TRY(hipEventCreate(&c->ev1));
  HIP_TRY(hipEventCreate(&c->evm));
  HIP_TRY(hipEventCreateWithFlags(&c->evs, hipEventDisableTiming));
  return LDSP_OK;
}

int ldsp_ctx_create(int device, ldsp_ctx** out) {
  if (!out) return fail(LDSP_ERR_INVALID_ARG, "ldsp_ctx_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(LDSP_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
  ldsp_device_guard guard(device);
  ldsp_ctx* c = new (std::nothrow) ldsp_ctx();
  if (!c) return fail(LDSP_ERR_NOMEM, "out of host memory");
  c->device = device;
  const int rc = ctx_build(c);
  if (rc) {   // a half-built context is released (the error string of the failing call is kept)
    const std::string keep = g_err;
    (void)ldsp_ctx_destroy(c);
    g_err = keep;
    return rc;
  }
  *out = c;
  return LDSP_OK;
}

int ldsp_ctx_destroy(ldsp_ctx* c) {
  if (!c) return LDSP_OK;
  ldsp_device_guard guard(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  (void)hipFree(c->d_icpc); (void)hipFree(c->d_hc); (void)hipFree(c->d_hz); (void)hipFree(c->d_icpc_b); (void)hipFree(c->d_hc_b); (void)hipFree(c->d_hz_b); (void)hipFree(c->d_aux); (void)hipFree(c->d_coef); (void)hipFree(c->d_grid); (void)hipFree(c->d_fir_grid); (void)hipFree(c->d_sg_grid); (void)hipFree(c->d_sipm);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->evm) (void)hipEventDestroy(c->evm);
  if (c->evs) (void)hipEventDestroy(c->evs);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return LDSP_OK;
}

// The context's workspaces (filter tables, slabs, the two-kernel scratch) are reused by every launch: work queued on the
// previous stream may still be reading them when the first launch on the new stream overwrites them.  A change of stream
// therefore orders the new stream behind everything queued on the old one (event + stream wait, no host synchronisation).
// The previous stream is NOT touched: every run call has left an event behind on the stream it used (ldsp_run_guard), and the new
// stream waits for the latest one — the previous stream may already have been destroyed by its owner.
static int switch_stream(ldsp_ctx* c, hipStream_t next) {
  if (next == c->stream) return LDSP_OK;
  ldsp_device_guard guard(c->device);
  c->stream = next;   // adopted whatever happens below
  if (c->evs_recorded) HIP_TRY(hipStreamWaitEvent(next, c->evs, 0));
  return LDSP_OK;
}

int ldsp_ctx_set_stream(ldsp_ctx* c, void* s) {
  if (!c) return fail(LDSP_ERR_INVALID_ARG, "ctx is NULL");
  return switch_stream(c, reinterpret_cast<hipStream_t>(s));
}

int ldsp_ctx_use_own_stream(ldsp_ctx* c) {
  if (!c) return fail(LDSP_ERR_INVALID_ARG, "ctx is NULL");
  return switch_stream(c, c->own_stream);
}

int ldsp_ctx_synchronize(ldsp_ctx* c) {
  if (!c) return fail(LDSP_ERR_INVALID_ARG, "ctx is NULL");
  HIP_TRY(hipStreamSynchronize(c->stream));
  return LDSP_OK;
}

int ldsp_ctx_set_option(ldsp_ctx* c, const char* key, int64_t value) {
  if (!c || !key) return fail(LDSP_ERR_INVALID_ARG, "ctx/key is NULL");
  if (!strcmp(key, "cusp_direct")) { c->cusp_direct = value != 0; return LDSP_OK; }
  if (!strcmp(key, "two_kernel")) { c->two_kernel = value != 0; return LDSP_OK; }
  if (!strcmp(key, "sipm_generic")) { c->sipm_generic = value != 0; return LDSP_OK; }
  if (!strcmp(key, "dbg_stop")) { c->dbg_stop = (int)value; c->icpc_valid = c->icpc_valid_b = false; return LDSP_OK; }
  if (!strcmp(key, "dbg_lds_pad")) { ldsp::g_dbg_lds_pad = (int)value; return LDSP_OK; }
  if (!strcmp(key, "icpc_generic")) { c->icpc_generic = value != 0; return LDSP_OK; }
  if (!strcmp(key, "icpc_r2")) { c->icpc_r2 = value != 0; c->icpc_valid = c->icpc_valid_b = false; return LDSP_OK; }
  if (!strcmp(key, "dbg_stamps")) { c->dbg_stamps = reinterpret_cast<long long*>((uintptr_t)value); c->icpc_valid = c->icpc_valid_b = false; return LDSP_OK; }
  if (!strcmp(key, "multi_serial")) { c->multi_serial = value != 0; return LDSP_OK; }
  if (!strcmp(key, "fir_grid_per_point")) { c->fir_grid_per_point = value != 0; return LDSP_OK; }
  return fail(LDSP_ERR_INVALID_ARG, "unknown option '%s'", key);
}

int ldsp_ctx_enable_timing(ldsp_ctx* c, int on) {
  if (!c) return fail(LDSP_ERR_INVALID_ARG, "ctx is NULL");
  c->timing = on != 0;
  return LDSP_OK;
}

int ldsp_ctx_last_kernel_ms(ldsp_ctx* c, float* ms) {
  if (!c || !ms) return fail(LDSP_ERR_INVALID_ARG, "ctx/ms is NULL");
  if (!c->timing || c->n_launches == 0) return fail(LDSP_ERR_INVALID_ARG, "timing not enabled or nothing launched");
  HIP_TRY(hipEventSynchronize(c->ev1));
  float t = 0;
  HIP_TRY(hipEventElapsedTime(&t, c->ev0, c->ev1));
  *ms = t / (float)c->n_launches;
  return LDSP_OK;
}

const char* ldsp_ctx_last_kernel_name(ldsp_ctx* c) { return c ? c->last_kernel : ""; }

int ldsp_ctx_last_stage_ms(ldsp_ctx* c, int stage, float* ms) {
  if (!c || !ms) return fail(LDSP_ERR_INVALID_ARG, "ctx/ms is NULL");
  if (!c->timing || c->n_launches == 0) return fail(LDSP_ERR_INVALID_ARG, "timing not enabled or nothing launched");
  if (stage < 0 || stage >= c->n_stages) return fail(LDSP_ERR_INVALID_ARG, "the last call had %d stage(s)", c->n_stages);
  HIP_TRY(hipEventSynchronize(c->ev1));
  hipEvent_t a = stage == 0 ? c->ev0 : c->evm, b = (stage == 0 && c->n_stages > 1) ? c->evm : c->ev1;
  HIP_TRY(hipEventElapsedTime(ms, a, b));
  return LDSP_OK;
}

int ldsp_cusp_coeffs(const ldsp_cuspzac* p, double* h) {
  if (!p || !h || !hm::cusp_shape_ok(*p)) return fail(LDSP_ERR_INVALID_ARG, "bad CUSP parameters");
  std::vector<double> v;
  hm::cuspzac_taps(*p, false, v);
  memcpy(h, v.data(), sizeof(double) * v.size());
  return LDSP_OK;
}
int ldsp_zac_coeffs(const ldsp_cuspzac* p, double* h) {
  if (!p || !h || !hm::cusp_shape_ok(*p)) return fail(LDSP_ERR_INVALID_ARG, "bad ZAC parameters");
  std::vector<double> v;
  hm::cuspzac_taps(*p, true, v);
  memcpy(h, v.data(), sizeof(double) * v.size());
  return LDSP_OK;
}
int ldsp_sg_coeffs(int32_t npts, int32_t degree, int32_t derivative, double* h) {
  std::vector<double> c;
  if (!h || !hm::sg_corr_coeffs(npts, degree, derivative, c)) return fail(LDSP_ERR_INVALID_ARG, "bad Savitzky-Golay parameters");
  for (int i = 0; i < npts; ++i) h[npts - 1 - i] = c[i];  // true-convolution order
  return LDSP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// lowering ldsp_icpc_params -> IcpcDev

static int round_up(int a, int b) { return (a + b - 1) / b * b; }

static bool check_window(int from, int until, int n) { return 0 <= from && from <= until && until <= n - 1; }

static WinDev make_win(int from, int until) {
  WinDev w;
  w.from = from; w.until = until;
  double n = (double)(until - from + 1);
  w.ic = 0.5 * ((double)from + (double)until);
  w.inv_n = 1.0 / n;
  w.var_i = (n * n - 1.0) / 12.0;
  return w;
}
static TrapDev make_trap(const ldsp_trap& t) {
  TrapDev d;
  d.n1 = t.navg; d.g = t.ngap; d.n2 = t.navg2; d.flen = t.navg + t.ngap + t.navg2;
  d.inv1 = (float)(1.0 / t.navg); d.inv2 = (float)(1.0 / t.navg2);
  d.rr = (float)((double)t.navg / (double)t.navg2); d.navg = (float)t.navg;
  return d;
}
static bool trap_ok(const ldsp_trap& t, int L) {   // (64-bit sum: the fields are whatever the caller's struct held)
  return t.navg >= 1 && t.navg2 >= 1 && t.ngap >= 0 && (int64_t)t.navg + (int64_t)t.ngap + (int64_t)t.navg2 <= (int64_t)L;
}

static bool make_est(const ldsp_dni& e, EstDev& d) {
  if (e.npts < 1 || e.npts > LDSP_MAX_EST_PTS || e.degree < 0 || e.degree > LDSP_MAX_EST_DEG || e.degree >= e.npts) return false;
  std::vector<double> B;
  double c, s;
  if (!hm::lsq_basis(e.npts, e.degree, B, c, s)) return false;
  d.npts = e.npts; d.deg = e.degree; d.c = (float)c; d.s_inv = (float)(1.0 / s);
  memset(d.B, 0, sizeof d.B);
  for (int i = 0; i < e.npts; ++i)
    for (int j = 0; j <= e.degree; ++j) d.B[i * (LDSP_MAX_EST_DEG + 1) + j] = (float)B[(size_t)i * (e.degree + 1) + j];
  return true;
}

static bool make_cuspzac(const ldsp_cuspzac& p, bool zac, CuspZacDev& d) {
  hm::CuspShape g = hm::cusp_geometry(p);
  memset(&d, 0, sizeof d);
  d.Lf = g.Lf; d.lt = g.lt; d.flat = g.flat; d.f1 = g.f1; d.ltp = g.ltp;
  d.is_zac = zac;
  const double q = std::exp(-1.0 / p.sigma), sc = p.beta / (double)p.length;
  for (int e = 0; e <= 4; ++e) d.qp1[e] = (float)std::exp(-(double)e / p.sigma);
  for (int j = 0; j <= 64; ++j) d.qp4[j] = (float)std::exp(-4.0 * j / p.sigma);
  for (int j = 0; j <= 64; ++j) d.qpw[j] = (float)std::exp(-256.0 * j / p.sigma);
  d.eps = (float)(-std::expm1(-1.0 / p.tau));
  d.sc = (float)sc;
  d.sc_half_den = (float)(sc * 0.5 / g.den);
  d.q_lt = (float)std::exp(-(double)g.lt / p.sigma);
  d.q_mlt1 = (float)std::exp((double)(g.lt - 1) / p.sigma);
  d.q_ltp1 = (float)std::exp(-(double)(g.ltp - 1) / p.sigma);
  d.q_mltp = (float)std::exp((double)g.ltp / p.sigma);
  d.q1 = (float)q; d.q2 = (float)(q * q);
  std::vector<double> cusp, par;
  hm::cusp_and_par(p, cusp, par);
  double apar = 0, acusp = 0;
  for (int j = 0; j < g.Lf; ++j) { apar += par[j]; acusp += cusp[j]; }
  const double rho = zac ? acusp / apar : 0.0;
  d.w_last = (float)(sc * (cusp[g.Lf - 1] - rho * par[g.Lf - 1]));
  d.rho_sc = (float)(rho * sc);
  {
    std::vector<double> taps;
    hm::cuspzac_taps(p, zac, taps);
    double hs = 0;
    for (double t : taps) hs += t;
    d.hsum = hs;
  }
  if (!zac) return true;
  // Second difference of the parabola kernel restricted to taps 0..Lf-2 (the last
  // tap multiplies y[k] directly): constant 2 on two ranges plus a few edge taps.
  // u[n] = sum_j D2[j] d[n-j],  d[i] = Dp[i]-Dp[i-1]  =>  u[n] = sum_e coef_e Dp[n-shift_e].
  std::vector<double> ext(g.Lf + 3, 0.0), D2(g.Lf + 3, 0.0);
  for (int j = 0; j <= g.Lf - 2; ++j) ext[j] = par[j];
  for (int j = 0; j < g.Lf + 2; ++j) D2[j] = ext[j] - (j >= 1 ? 2 * ext[j - 1] : 0.0) + (j >= 2 ? ext[j - 2] : 0.0);
  // coefficient of Dp[n-s]: D2[s] - D2[s-1]
  std::vector<std::pair<int, double>> terms;
  for (int sft = 0; sft <= g.Lf + 2; ++sft) {
    double c = (sft < g.Lf + 2 ? D2[sft] : 0.0) - (sft >= 1 ? D2[sft - 1] : 0.0);
    if (std::fabs(c) > 1e-9) terms.push_back({sft, c});
  }
  // summation by parts: sum_e c_e Dp[n-s_e] = sum_e R_e (Dp[n-s_e] - Dp[n-s_{e+1}]),  R_e = c_0 + .. + c_e  (R_last = 0)
  if (terms.size() < 2 || terms.size() > 13) return false;
  double run = 0, total = 0;
  for (auto& t : terms) total += t.second;
  if (std::fabs(total) > 1e-6) return false;
  d.zu_n = 0;
  d.zc_n = (int)terms.size();
  {
    double rr = 0;
    for (size_t e = 0; e < terms.size(); ++e) {
      d.zc_s[e] = terms[e].first;
      rr += terms[e].second;
      if (e + 1 < terms.size()) d.zc_r[e] = (float)(std::fabs(rr) <= 1e-9 ? 0.0 : rr);
    }
  }
  {   // the chain with the last tap folded in
    std::vector<std::pair<int, double>> tf = terms;
    const double pl = par[g.Lf - 1];
    const std::pair<int, double> extra[3] = {{g.Lf - 1, pl}, {g.Lf, -2.0 * pl}, {g.Lf + 1, pl}};
    for (auto& x : extra) {
      bool found = false;
      for (auto& t : tf) if (t.first == x.first) { t.second += x.second; found = true; }
      if (!found) tf.push_back(x);
    }
    std::sort(tf.begin(), tf.end());
    std::vector<std::pair<int, double>> tg;
    for (auto& t : tf) if (std::fabs(t.second) > 1e-9) tg.push_back(t);
    d.zf_n = 0;
    if (tg.size() >= 2 && tg.size() <= 13) {
      d.zf_n = (int)tg.size();
      double rr = 0;
      for (size_t e = 0; e < tg.size(); ++e) {
        d.zf_s[e] = tg[e].first;
        rr += tg[e].second;
        if (e + 1 < tg.size()) d.zf_r[e] = (float)(std::fabs(rr) <= 1e-9 ? 0.0 : rr);
      }
    }
  }
  for (size_t e = 0; e + 1 < terms.size(); ++e) {
    run += terms[e].second;
    if (std::fabs(run) <= 1e-9) continue;
    d.zu_shift[d.zu_n] = terms[e].first; d.zu_shift_b[d.zu_n] = terms[e + 1].first; d.zu_coef[d.zu_n] = (float)run;
    ++d.zu_n;
  }
  return true;
}

static int lower_icpc_dev(const ldsp_icpc_params& p, int cusp_direct, int r2, IcpcDev& d, std::vector<float>& hc, std::vector<float>& hz) {
  memset(&d, 0, sizeof d);
  const int L = p.L;
  if (L < 64 || L > LDSP_MAX_L) return fail(LDSP_ERR_UNSUPPORTED, "trace length %d outside [64, %d]", L, LDSP_MAX_L);
  if (!(p.dt > 0) || !(p.unit_per_us > 0)) return fail(LDSP_ERR_INVALID_ARG, "dt and unit_per_us must be positive");
  d.L = L;
  // launch geometry: NT threads x R float4 rows per thread, NT*4*R >= L
  if (L <= 1024) { d.NT = 64; d.R = 4; }
  else if (L <= 2048) { d.NT = 128; d.R = 4; }
  else if (L <= 4096) { d.NT = 256; d.R = 4; }
  else if (L <= 8192) { d.NT = 512; d.R = 4; if (r2) { d.NT = 1024; d.R = 2; } }
  else if (L <= 16384) { d.NT = 1024; d.R = 4; }
  else return fail(LDSP_ERR_UNSUPPORTED, "dsp_icpc kernel keeps the trace in LDS: L <= 16384 (got %d)", L);
  d.t_first = (float)p.t_first; d.dt = (float)p.dt;
  d.unit_per_us = (float)p.unit_per_us; d.inv_unit_per_us = (float)(1.0 / p.unit_per_us);
  d.sat_low = (float)p.sat_low; d.sat_high = (float)p.sat_high;
  if (!check_window(p.bl_from, p.bl_until, L)) return fail(LDSP_ERR_WINDOW, "bl_window [%d,%d] outside trace", p.bl_from, p.bl_until);
  if (!check_window(p.tail_from, p.tail_until, L)) return fail(LDSP_ERR_WINDOW, "tail_window [%d,%d] outside trace", p.tail_from, p.tail_until);
  d.bl = make_win(p.bl_from, p.bl_until);
  d.tail = make_win(p.tail_from, p.tail_until);
  d.pz_c = (float)p.pz_c; d.pz_c64 = p.pz_c;
  const ldsp_trap* traps[6] = {&p.t0_trap, &p.t0inv_trap, &p.trap_fixed[0], &p.trap_fixed[1], &p.trap_fixed[2], &p.trap_opt};
  for (auto t : traps)
    if (!trap_ok(*t, L)) return fail(LDSP_ERR_WINDOW, "trapezoid (%d,%d,%d) does not fit a trace of %d samples", t->navg, t->ngap, t->navg2, L);
  d.t0 = make_trap(p.t0_trap); d.t0inv = make_trap(p.t0inv_trap);
  for (int i = 0; i < 3; ++i) d.fixed[i] = make_trap(p.trap_fixed[i]);
  d.opt = make_trap(p.trap_opt);
  d.t0inv_same = !memcmp(&p.t0_trap, &p.t0inv_trap, sizeof(ldsp_trap));
  if (p.t0_mintot < 1 || p.tx_mintot < 1 || p.intrace_mintot < 1) return fail(LDSP_ERR_INVALID_ARG, "mintot values must be >= 1 sample");
  d.t0_mintot = p.t0_mintot; d.tx_mintot = p.tx_mintot; d.intrace_mintot = p.intrace_mintot;
  d.t0_thr = (float)p.t0_threshold; d.intrace_nsigma = (float)p.intrace_nsigma;
  if (!make_est(p.int_est, d.int_est) || !make_est(p.sig_est, d.sig_est))
    return fail(LDSP_ERR_UNSUPPORTED, "PolynomialDNI window/degree outside the built limits (%d pts, degree %d)", LDSP_MAX_EST_PTS, LDSP_MAX_EST_DEG);
  d.qdrift_d1 = (float)(p.qdrift_d1 / p.dt); d.qdrift_d2 = (float)(p.qdrift_d2 / p.dt);
  d.lq_d1 = (float)(p.lq_d1 / p.dt); d.lq_d2 = (float)(p.lq_d2 / p.dt);
  d.trap_pickoff = (float)(p.trap_pickoff / p.dt);
  d.cusp_pickoff = (float)(p.cusp_pickoff / p.dt);
  d.zac_pickoff = (float)(p.zac_pickoff / p.dt);
  for (int f = 0; f < 3; ++f) {
    const int np = p.sg_npts[f];
    if (np < 1 || np > LDSP_MAX_SG_PTS || (np & 1) == 0 || np <= p.sg_degree || np > L)
      return fail(LDSP_ERR_UNSUPPORTED, "Savitzky-Golay window of %d points (degree %d) unsupported", np, p.sg_degree);
    std::vector<double> cc;
    if (!hm::sg_corr_coeffs(np, p.sg_degree, 1, cc)) return fail(LDSP_ERR_INVALID_ARG, "Savitzky-Golay coefficients");
    d.sg_npts[f] = np;
    for (int i = 0; i < np; ++i) d.sg_c[f][i] = (float)cc[i];
    const double tf = p.t_first + (np - 1) * p.dt;  // trailing alignment (A1)
    d.cur_from[f] = (int)std::nearbyint((p.cur_left - tf) / p.dt);
    d.cur_until[f] = (int)std::nearbyint((p.cur_right - tf) / p.dt);
    if (!check_window(d.cur_from[f], d.cur_until[f], L - np + 1)) return fail(LDSP_ERR_WINDOW, "current_window outside the SG output axis");
  }
  d.sg_same_02 = d.sg_npts[0] == d.sg_npts[2];
  d.cur_from[3] = (int)std::nearbyint((p.cur_left - p.t_first) / p.dt);
  d.cur_until[3] = (int)std::nearbyint((p.cur_right - p.t_first) / p.dt);
  if (!check_window(d.cur_from[3], d.cur_until[3], L)) return fail(LDSP_ERR_WINDOW, "current_window outside trace");
  {
    const int np = d.sg_npts[0];
    const double tf = p.t_first + (np - 1) * p.dt;
    int from = (int)std::nearbyint(((p.bl_left + tf) - tf) / p.dt);  // dsp_routines.jl:75
    int until = (int)std::nearbyint((p.bl_right - tf) / p.dt);
    if (!check_window(from, until, L - np + 1)) return fail(LDSP_ERR_WINDOW, "bl_window outside the SG output axis");
    d.sgbl = make_win(from, until);
  }
  if (!hm::cusp_shape_ok(p.cusp) || !hm::cusp_shape_ok(p.zac) || p.cusp.length > L || p.zac.length > L ||
      p.cusp.length > LDSP_MAX_FIR_TAPS || p.zac.length > LDSP_MAX_FIR_TAPS)
    return fail(LDSP_ERR_WINDOW, "CUSP/ZAC filter does not fit the trace");
  bool cz_ok = make_cuspzac(p.cusp, false, d.cusp) && make_cuspzac(p.zac, true, d.zac);
  d.cz_shared = p.cusp.sigma == p.zac.sigma && p.cusp.flat == p.zac.flat && p.cusp.length == p.zac.length &&
                p.cusp.tau == p.zac.tau && p.cusp.beta == p.zac.beta;
  d.cusp_mode = (cusp_direct || !cz_ok) ? 0 : 1;  // unexpected kernel structure -> direct-form comparator
  {
    const int wlo = std::min(std::min(d.cur_from[1], d.cur_from[2]), d.cur_from[3]), whi = std::max(std::max(d.cur_until[1], d.cur_until[2]), d.cur_until[3]);
    const int ilo = std::max(std::max(d.cur_from[1], d.cur_from[2]), d.cur_from[3]), ihi = std::min(std::min(d.cur_until[1], d.cur_until[2]), d.cur_until[3]);
    const int win[6][2] = {{d.bl.from, d.bl.until}, {d.tail.from, d.tail.until}, {d.sgbl.from, d.sgbl.until}, {d.cur_from[0], d.cur_until[0]}, {wlo, whi}, {ilo, ihi}};
    for (int w = 0; w < 6; ++w)
      for (int v = 0; v < 16; ++v) {
        uint32_t m = 0;
        for (int r = 0; r < 4; ++r) {
          const int a = 4 * (64 * v + d.NT * r), b = a + 255;
          if (b < win[w][0] || a > win[w][1]) m |= 1u << r;
          else if (a >= win[w][0] && b <= win[w][1]) m |= 16u << r;
        }
        d.rowcls[w][v] = m;
      }
  }
  std::vector<double> h;
  hm::cuspzac_taps(p.cusp, false, h);
  hc.assign(h.begin(), h.end());
  hm::cuspzac_taps(p.zac, true, h);
  hz.assign(h.begin(), h.end());
  return LDSP_OK;
}

// The lean kernels (icpc_lean3.hip: the fused chain; icpc_lean.hip: config 2's sub-chain) cover the standard geometry: the trace
// fills the tile (L = 16 NT, NT <= 512), CUSP and ZAC in closed form (sharing their geometry: one pass; optimised separately:
// one pass each inside the same launch), the inverted t0 uses the same trapezoid, tx_mintot <= 2 samples, Savitzky-Golay windows
// of at most 25 taps (the optimised one) / 13 taps (the two fixed ones), the chain of ZAC shifts with the parabola's last tap folded in exists (icpc_dev.hpp: zf_*), and the
// eps * T term of the filters' last tap, which icpc_lean3 drops, is far below the columns' resolution: |w_last| * eps * rail * L
// < 1e-2 (for a 16-bit rail; the bound scales with the rail: 1.5e-7 of full scale) on a trace that sits at the rail, a hundredth of that on a real one (dsp_icpc sets the filters' tau to 1e7 us,
// src/dsp_icpc.jl:98: 1.4e-3 for a 16-bit rail and 8192 samples).  ldsp_icpc_run and ldsp_icpc_pz_trap_run decide alike.
// `full_tile`: only traces that fill the tile (config 2's lean kernel); the fused kernel also takes shorter traces of any length
// (8000-, 7300-, 8190-, 7001-sample traces run it on the next tile up; when the length is no multiple of four samples the rows are
// 4-byte aligned and the quad that holds the end of a trace is read sample by sample).
static bool icpc_lean_applies(const ldsp_ctx* c, bool full_tile = false, bool skip_cz = false) {
  const IcpcDev& H = c->icpc_host;
  // (Savitzky-Golay: the optimised window up to 25 taps — 350 ns at 16 ns, the end of the reference's scan grid, test/test_dsp_icpc.jl:134-138 —,
  // the two fixed windows, 60 and 100 ns, up to 13; a third filter equal to the first is not evaluated again)
  const int sg_fixed = std::max(H.sg_npts[1], H.sg_same_02 ? 0 : H.sg_npts[2]);
  if (c->icpc_generic || c->two_kernel || !(c->dbg_stop == 0 || c->dbg_stop >= 100) || H.R != 4 || H.L > 16 * H.NT || H.NT > 512 ||
      H.L <= 8 * H.NT || (full_tile && H.L != 16 * H.NT) ||   // (more than half the tile: its first two rows are in the trace)
      (!skip_cz && H.cusp_mode != 1) || !H.t0inv_same || H.tx_mintot > 2 || H.sg_npts[0] > 25 || sg_fixed > 13)
    return false;
  // (a call without the CUSP / ZAC stage — main_only, the windowed traces of dsp_icpc_compressed — leaves that stage's terms out: the form
  // of its filters, the dropped eps * T term, the folded ZAC chain)
  if (skip_cz) return icpc_lean3_smem_bytes(H.NT, 8) <= 80640;
  // (the rails bound the signal; rails left at zero — saturation not configured — are taken as a 16-bit range, not as "no signal")
  const double amp = std::max(std::max(std::fabs((double)H.sat_high), std::fabs((double)H.sat_low)), 65535.0);
  const double rail = amp * (double)H.L;
  const double drop = std::max(std::fabs((double)H.cusp.w_last) * H.cusp.eps, std::fabs((double)H.zac.w_last) * H.zac.eps) * rail;
  if (H.cz_shared && H.zac.zf_n <= 0) return false;
  // The bound scales with the rail: 1e-2 at a 16-bit range, i.e. 1.5e-7 of full scale — a quarter of a float32 ulp there.  Presummed
  // traces (dsp_icpc_compressed: the upper rail times the presum rate, src/dsp_icpc.jl:339) carry the same physics at rate x the
  // amplitude and rate x the sampling step; an absolute bound sent them to the generic kernel for a term the same size relative to
  // their columns' float32 resolution.
  return drop < 1e-2 * (amp / 65535.0) && icpc_lean3_smem_bytes(H.NT, std::max(H.cusp.Lf, H.zac.Lf)) <= 80640;   // (<= 53 760: three workgroups per CU)
}

static int prepare_icpc(ldsp_ctx* c, const ldsp_icpc_params* p, int in_u16 = 0) {
  auto current_matches = [&]() {
    return c->icpc_valid && c->icpc_mode_built == c->cusp_direct && c->icpc_u16_built == in_u16 && !memcmp(&c->icpc_last, p, sizeof *p);
  };
  if (current_matches()) return LDSP_OK;
  // the previous block becomes the current one (and is kept if a new one has to be built: the kernels of earlier calls read theirs in
  // stream order, and a block is only ever written by stream-ordered copies)
  std::swap(c->icpc_last, c->icpc_last_b); std::swap(c->icpc_valid, c->icpc_valid_b); std::swap(c->icpc_mode_built, c->icpc_mode_built_b);
  std::swap(c->icpc_u16_built, c->icpc_u16_built_b); std::swap(c->icpc_host, c->icpc_host_b);
  std::swap(c->d_icpc, c->d_icpc_b); std::swap(c->d_hc, c->d_hc_b); std::swap(c->d_hz, c->d_hz_b);
  if (current_matches()) return LDSP_OK;
  c->icpc_valid = false;
  std::vector<float> hc, hz;
  IcpcDev d;
  int rc = lower_icpc_dev(*p, c->cusp_direct, c->icpc_r2, d, hc, hz);
  if (rc) return rc;
  d.h_cusp = c->d_hc; d.h_zac = c->d_hz;
  d.dbg_stop = c->dbg_stop;
  d.dbg_stamps = c->dbg_stamps;
  d.in_u16 = in_u16;
  c->icpc_host = d;
  HIP_TRY(hipMemcpyAsync(c->d_icpc, &c->icpc_host, sizeof(IcpcDev), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_hc, hc.data(), sizeof(float) * hc.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_hz, hz.data(), sizeof(float) * hz.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // staging vectors die at return
  c->icpc_last = *p; c->icpc_valid = true; c->icpc_mode_built = c->cusp_direct; c->icpc_u16_built = in_u16;
  return LDSP_OK;
}

extern "C" {

int ldsp_icpc_run(ldsp_ctx* c, const float* wf, int64_t n, const ldsp_icpc_params* p, const ldsp_icpc_out* out) {
  return ldsp_icpc_run_opts(c, wf, n, p, nullptr, out);
}

// the whole host-side lowering of a parameter block (window / filter checks, filter constants and taps, launch geometry) without
// a device: what ldsp_icpc_run would reject, it rejects with the same code and message
int ldsp_icpc_check_params(const ldsp_icpc_params* p) {
  if (!p) return fail(LDSP_ERR_INVALID_ARG, "ldsp_icpc_check_params: NULL argument");
  std::vector<float> hc, hz;
  IcpcDev d;
  for (int direct = 0; direct < 2; ++direct) {
    const int rc = lower_icpc_dev(*p, direct, 0, d, hc, hz);
    if (rc) return rc;
  }
  return LDSP_OK;
}

int ldsp_icpc_run_opts(ldsp_ctx* c, const float* wf, int64_t n, const ldsp_icpc_params* p, const ldsp_icpc_opts* opts,
                       const ldsp_icpc_out* out) {
  if (!c || !p || !out) return fail(LDSP_ERR_INVALID_ARG, "ldsp_icpc_run: NULL argument");
  const bool main_only = opts && opts->main_only != 0;
  const float* ext_bl = opts ? opts->ext_baseline : nullptr;
  const float ext_bl_scale = opts ? (float)opts->ext_baseline_scale : 1.f;
  const int in_u16 = (opts && opts->in_u16) ? 1 : 0;
  if (n < 0 || n > 0x7fffffffLL) return fail(LDSP_ERR_INVALID_ARG, "n = %lld out of range", (long long)n);
  if (n == 0) return LDSP_OK;
  if (!wf) return fail(LDSP_ERR_INVALID_ARG, "waveform pointer is NULL");
  ldsp_run_guard guard(c);
  int rc = prepare_icpc(c, p, in_u16);
  if (rc) return rc;
  IcpcOutDev od;
  static_assert(sizeof(ldsp_icpc_out) == sizeof(void*) * LDSP_ICPC_NCOLS + sizeof(int64_t), "ldsp_icpc_out layout");
  memcpy(od.col, out, sizeof(void*) * LDSP_ICPC_NCOLS);
  od.stride = out->stride > 0 ? out->stride : 1;
  if (main_only)   // include/ldsp.h: the six CUSP / ZAC columns are not written by such a call, whichever kernel runs it
    for (int col : {C_e_cusp, C_e_zac, C_e_cusp_max, C_e_zac_max, C_t_cusp_max, C_t_zac_max}) od.col[col] = nullptr;
  if (n > c->aux_cap) {  // grow-only workspace; steady-state calls do not allocate
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_aux);
    c->d_aux = nullptr; c->aux_cap = 0;
    HIP_TRY(hipMalloc(&c->d_aux, sizeof(float) * 4 * (size_t)n));
    c->aux_cap = n;
  }
  if (c->timing) HIP_TRY(hipEventRecord(c->ev0, c->stream));
  int stages = 1;
  // the lean kernel (icpc_lean3.hip) covers the standard geometry; everything else — and option "icpc_generic" — runs icpc_kernel
  const IcpcDev& H = c->icpc_host;
  const int sg_max = std::max(H.sg_npts[0], std::max(H.sg_npts[1], H.sg_npts[2]));
  // (main_only — the windowed traces of dsp_icpc_compressed — runs the same kernel with its CUSP / ZAC stage skipped)
  if (icpc_lean_applies(c, false, main_only)) {
    HIP_TRY(launch_icpc_lean3(wf, n, H.NT, sg_max, main_only || H.cz_shared != 0, H.L == 16 * H.NT, c->d_icpc, od, ext_bl, ext_bl_scale,
                              main_only ? 8 : std::max(H.cusp.Lf, H.zac.Lf), main_only, c->stream));
    c->last_kernel = "lean3::icpc_lean3_kernel";
    if (c->timing) { HIP_TRY(hipEventRecord(c->ev1, c->stream)); c->n_launches = 1; c->n_stages = 1; }
    return LDSP_OK;
  }
  HIP_TRY(launch_icpc(wf, n, c->icpc_host.NT, c->icpc_host.R, c->icpc_host.L == 4 * c->icpc_host.R * c->icpc_host.NT, c->d_icpc, c->d_aux, od, ext_bl, ext_bl_scale,
                      c->icpc_host.cusp_mode == 0, c->icpc_host.cz_shared != 0, !c->two_kernel && !main_only,
                      main_only || (c->dbg_stop > 0 && c->dbg_stop < 10),
                      ((std::max(c->icpc_host.cusp.Lf, c->icpc_host.zac.Lf) + 2 + 7) & ~3), c->stream,
                      c->timing ? c->evm : nullptr, &stages));
  c->last_kernel = "icpc_kernel";
  if (c->timing) { HIP_TRY(hipEventRecord(c->ev1, c->stream)); c->n_launches = 1; c->n_stages = stages; }
  return LDSP_OK;
}

static int pz_trap_run_impl(ldsp_ctx* c, const float* wf, int64_t n, const ldsp_icpc_params* p, float* blmean, float* e_10410, bool in_u16) {
  if (!c || !p || !blmean || !e_10410) return fail(LDSP_ERR_INVALID_ARG, "ldsp_icpc_pz_trap_run: NULL argument");
  if (n < 0 || n > 0x7fffffffLL) return fail(LDSP_ERR_INVALID_ARG, "n = %lld out of range", (long long)n);
  if (n == 0) return LDSP_OK;
  if (!wf) return fail(LDSP_ERR_INVALID_ARG, "waveform pointer is NULL");
  ldsp_run_guard guard(c);
  int rc = prepare_icpc(c, p, in_u16);   // (the kernels read in_u16 from the parameter block)
  if (rc) return rc;
  if (c->timing) HIP_TRY(hipEventRecord(c->ev0, c->stream));
  const int nt_pz = c->icpc_host.R == 2 ? 512 : c->icpc_host.NT;   // pz_trap_kernel keeps 16 samples per thread
  if (icpc_lean_applies(c, true)) {
    HIP_TRY(launch_pz_trap_lean(wf, n, c->icpc_host.NT, in_u16, c->d_icpc, blmean, e_10410, c->stream));
    c->last_kernel = "lean::pz_trap_lean_kernel";
  } else {
    HIP_TRY(launch_pz_trap(wf, n, nt_pz, c->icpc_host.L == 16 * nt_pz, c->d_icpc, blmean, e_10410, c->stream));
    c->last_kernel = "pz_trap_kernel";
  }
  if (c->timing) { HIP_TRY(hipEventRecord(c->ev1, c->stream)); c->n_launches = 1; c->n_stages = 1; }
  return LDSP_OK;
}
int ldsp_icpc_pz_trap_run(ldsp_ctx* c, const float* wf, int64_t n, const ldsp_icpc_params* p, float* blmean, float* e_10410) {
  return pz_trap_run_impl(c, wf, n, p, blmean, e_10410, false);
}
int ldsp_icpc_pz_trap_run_u16(ldsp_ctx* c, const uint16_t* wf, int64_t n, const ldsp_icpc_params* p, float* blmean, float* e_10410) {
  return pz_trap_run_impl(c, reinterpret_cast<const float*>(wf), n, p, blmean, e_10410, true);
}

int ldsp_trap_grid_run(ldsp_ctx* c, const float* wf, int64_t n, const ldsp_trapgrid_params* p, int32_t G, const ldsp_trap* traps,
                       const double* offsets, float* out) {
  if (!c || !p || !traps || !out) return fail(LDSP_ERR_INVALID_ARG, "ldsp_trap_grid_run: NULL argument");
  int rc = ldsp_check_batch(c, wf, n, p->L, "ldsp_trap_grid_run");
  if (rc || n == 0) return rc;
  const int L = p->L;
  if (G < 1 || G > LDSP_MAX_GRID) return fail(LDSP_ERR_UNSUPPORTED, "grid of %d points (1..%d supported)", G, LDSP_MAX_GRID);
  if (!(p->dt > 0)) return fail(LDSP_ERR_INVALID_ARG, "dt must be positive");
  if (!check_window(p->bl_from, p->bl_until, L)) return fail(LDSP_ERR_WINDOW, "bl_window [%d,%d] outside trace", p->bl_from, p->bl_until);
  if (p->pick_mode != 0 && p->pick_mode != 1) return fail(LDSP_ERR_INVALID_ARG, "pick_mode must be 0 or 1");
  if (p->pick_mode == 1 && (!offsets || p->tx_mintot < 1)) return fail(LDSP_ERR_INVALID_ARG, "mode 1 needs offsets and tx_mintot >= 1");
  TrapGridDev d;   // host staging (~4 KB), copied before the call returns
  memset(&d, 0, sizeof d);
  d.L = L; d.G = G; d.pick_mode = p->pick_mode; d.tx_mintot = p->tx_mintot;
  d.NT = 64;
  while (d.NT * 16 < L) d.NT *= 2;
  if (d.NT > 1024) return fail(LDSP_ERR_UNSUPPORTED, "trace length %d: at most 16384 samples here", L);
  d.t_first = (float)p->t_first; d.dt = (float)p->dt;
  d.bl = make_win(p->bl_from, p->bl_until);
  d.pz_c = (float)p->pz_c; d.pz_c64 = p->pz_c;
  if (!make_est(p->sig_est, d.est)) return fail(LDSP_ERR_UNSUPPORTED, "PolynomialDNI(%d, %d points) unsupported", p->sig_est.degree, p->sig_est.npts);
  const double pos = (p->pick_time - p->t_first) / p->dt, fl = std::floor(pos);
  d.pick_ip = (int)fl; d.pick_fp = (float)(pos - fl);
  for (int g = 0; g < G; ++g) {
    if (!trap_ok(traps[g], L) || traps[g].navg + traps[g].ngap + traps[g].navg2 + p->sig_est.npts > L + 1)
      return fail(LDSP_ERR_WINDOW, "grid trapezoid %d (%d,%d,%d) does not fit a trace of %d samples", g, traps[g].navg, traps[g].ngap, traps[g].navg2, L);
    d.trap[g] = make_trap(traps[g]);
    d.offs[g] = offsets ? (float)(offsets[g] / p->dt) : 0.f;
  }
  if (!c->d_grid) HIP_TRY(hipMalloc(&c->d_grid, sizeof(TrapGridDev) > sizeof(FirGridDev) ? sizeof(TrapGridDev) : sizeof(FirGridDev)));
  HIP_TRY(hipMemcpyAsync(c->d_grid, &d, sizeof d, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));   // the staging block is reused by the next call
  if (c->timing) HIP_TRY(hipEventRecord(c->ev0, c->stream));
  HIP_TRY(launch_trap_grid(wf, n, d.NT, L == 16 * d.NT, reinterpret_cast<const TrapGridDev*>(c->d_grid), out, c->stream));
  c->last_kernel = "trap_grid_kernel";
  if (c->timing) { HIP_TRY(hipEventRecord(c->ev1, c->stream)); c->n_launches = 1; c->n_stages = 1; }
  return LDSP_OK;
}

int ldsp_fir_grid_run(ldsp_ctx* c, const float* wf, int64_t n, const ldsp_trapgrid_params* p, int32_t G, int32_t Lf, const double* taps,
                      const double* offsets, float* out) {
  if (!c || !p || !taps || !out) return fail(LDSP_ERR_INVALID_ARG, "ldsp_fir_grid_run: NULL argument");
  int rc = ldsp_check_batch(c, wf, n, p->L, "ldsp_fir_grid_run");
  if (rc || n == 0) return rc;
  const int L = p->L;
  if (G < 1 || G > LDSP_MAX_GRID) return fail(LDSP_ERR_UNSUPPORTED, "grid of %d points (1..%d supported)", G, LDSP_MAX_GRID);
  if (Lf < 1 || Lf > LDSP_MAX_FIR_TAPS || Lf + p->sig_est.npts > L + 1) return fail(LDSP_ERR_WINDOW, "FIR of %d taps does not fit a trace of %d samples", Lf, L);
  if (!(p->dt > 0)) return fail(LDSP_ERR_INVALID_ARG, "dt must be positive");
  if (!check_window(p->bl_from, p->bl_until, L)) return fail(LDSP_ERR_WINDOW, "bl_window [%d,%d] outside trace", p->bl_from, p->bl_until);
  if (p->pick_mode != 0 && p->pick_mode != 1) return fail(LDSP_ERR_INVALID_ARG, "pick_mode must be 0 or 1");
  if (p->pick_mode == 1 && (!offsets || p->tx_mintot < 1)) return fail(LDSP_ERR_INVALID_ARG, "mode 1 needs offsets and tx_mintot >= 1");
  FirGridDev d;
  memset(&d, 0, sizeof d);
  d.L = L; d.G = G; d.Lf = Lf; d.pick_mode = p->pick_mode; d.tx_mintot = p->tx_mintot;
  d.NT = 64;
  while (d.NT * 16 < L) d.NT *= 2;
  if (d.NT > 1024) return fail(LDSP_ERR_UNSUPPORTED, "trace length %d: at most 16384 samples here", L);
  d.t_first = (float)p->t_first; d.dt = (float)p->dt;
  d.bl = make_win(p->bl_from, p->bl_until);
  d.pz_c = (float)p->pz_c; d.pz_c64 = p->pz_c;
  if (!make_est(p->sig_est, d.est)) return fail(LDSP_ERR_UNSUPPORTED, "PolynomialDNI(%d, %d points) unsupported", p->sig_est.degree, p->sig_est.npts);
  const double pos = (p->pick_time - p->t_first) / p->dt, fl = std::floor(pos);
  d.pick_ip = (int)fl; d.pick_fp = (float)(pos - fl);
  for (int g = 0; g < G; ++g) d.offs[g] = offsets ? (float)(offsets[g] / p->dt) : 0.f;
  d.same_offs = 1;
  for (int g = 1; g < G; ++g) d.same_offs &= (d.offs[g] == d.offs[0]);
  if (c->fir_grid_per_point) d.same_offs = 0;
  const size_t ntap = (size_t)G * (size_t)Lf;
  if (ntap > c->fir_grid_cap) {   // grow-only tap buffer
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_fir_grid); c->d_fir_grid = nullptr; c->fir_grid_cap = 0;
    HIP_TRY(hipMalloc(&c->d_fir_grid, ntap * sizeof(float)));
    c->fir_grid_cap = ntap;
  }
  std::vector<float> rev(ntap);   // reversed: out[k] = sum_j h[j] y[k+Lf-1-j] = sum_j c[j] y[k+j], c[j] = h[Lf-1-j]
  for (int g = 0; g < G; ++g)
    for (int j = 0; j < Lf; ++j) rev[(size_t)g * Lf + j] = (float)taps[(size_t)g * Lf + (Lf - 1 - j)];
  d.taps = c->d_fir_grid;
  if (!c->d_grid) HIP_TRY(hipMalloc(&c->d_grid, sizeof(TrapGridDev) > sizeof(FirGridDev) ? sizeof(TrapGridDev) : sizeof(FirGridDev)));
  HIP_TRY(hipMemcpyAsync(c->d_fir_grid, rev.data(), ntap * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_grid, &d, sizeof d, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));   // the staging vectors die at return
  if (c->timing) HIP_TRY(hipEventRecord(c->ev0, c->stream));
  HIP_TRY(launch_fir_grid(wf, n, d.NT, L == 16 * d.NT, reinterpret_cast<const FirGridDev*>(c->d_grid), out, c->stream));
  c->last_kernel = "fir_grid_kernel";
  if (c->timing) { HIP_TRY(hipEventRecord(c->ev1, c->stream)); c->n_launches = 1; c->n_stages = 1; }
  return LDSP_OK;
}

int ldsp_sg_grid_run(ldsp_ctx* c, const float* wf, int64_t n, const ldsp_trapgrid_params* p, const ldsp_trap* trap, double trap_offset,
                     double unit_per_us, int32_t W, const int32_t* npts, int32_t degree, const int32_t* from, const int32_t* until, float* amax,
                     float* energy, float* t50_us, float* blmean, float* blslope) {
  if (!c || !p || !trap || (W > 0 && (!npts || !from || !until))) return fail(LDSP_ERR_INVALID_ARG, "ldsp_sg_grid_run: NULL argument");
  int rc = ldsp_check_batch(c, wf, n, p->L, "ldsp_sg_grid_run");
  if (rc || n == 0) return rc;
  const int L = p->L;
  if (W < 0 || W > 32) return fail(LDSP_ERR_UNSUPPORTED, "grid of %d window lengths (0..32 supported)", W);
  if (!(p->dt > 0) || !(unit_per_us > 0)) return fail(LDSP_ERR_INVALID_ARG, "dt and unit_per_us must be positive");
  if (p->pick_mode != 1 || p->tx_mintot < 1) return fail(LDSP_ERR_INVALID_ARG, "ldsp_sg_grid_run picks off at t50: pick_mode 1, tx_mintot >= 1");
  if (!check_window(p->bl_from, p->bl_until, L)) return fail(LDSP_ERR_WINDOW, "bl_window [%d,%d] outside trace", p->bl_from, p->bl_until);
  if (!trap_ok(*trap, L)) return fail(LDSP_ERR_WINDOW, "trapezoid does not fit");
  static_assert(sizeof(SgGridDev) < 64 * 1024, "parameter block");
  std::vector<unsigned char> buf(sizeof(SgGridDev), 0);
  SgGridDev& d = *reinterpret_cast<SgGridDev*>(buf.data());
  d.L = L; d.W = W; d.tx_mintot = p->tx_mintot;
  d.NT = 64;
  while (d.NT * 16 < L) d.NT *= 2;
  if (d.NT > 1024) return fail(LDSP_ERR_UNSUPPORTED, "trace length %d: at most 16384 samples here", L);
  d.t_first = (float)p->t_first; d.dt = (float)p->dt; d.inv_unit_per_us = (float)(1.0 / unit_per_us);
  d.bl = make_win(p->bl_from, p->bl_until);
  d.pz_c = (float)p->pz_c; d.pz_c64 = p->pz_c;
  if (!make_est(p->sig_est, d.est)) return fail(LDSP_ERR_UNSUPPORTED, "PolynomialDNI(%d, %d points) unsupported", p->sig_est.degree, p->sig_est.npts);
  d.trap = make_trap(*trap);
  d.trap_off = (float)(trap_offset / p->dt);
  for (int g = 0; g < W; ++g) {
    const int np = npts[g];
    if (np < 1 || np > LDSP_MAX_SG_PTS || (np & 1) == 0 || np <= degree || np > L)
      return fail(LDSP_ERR_UNSUPPORTED, "Savitzky-Golay window of %d points (degree %d) unsupported", np, degree);
    if (!check_window(from[g], until[g], L - np + 1)) return fail(LDSP_ERR_WINDOW, "current window [%d,%d] outside the SG output axis", from[g], until[g]);
    std::vector<double> cc;
    if (!hm::sg_corr_coeffs(np, degree, 1, cc)) return fail(LDSP_ERR_INVALID_ARG, "Savitzky-Golay coefficients");
    d.np[g] = np; d.from[g] = from[g]; d.until[g] = until[g];
    for (int i = 0; i < np; ++i) d.c[g][i] = (float)cc[i];
  }
  if (!c->d_sg_grid) HIP_TRY(hipMalloc(&c->d_sg_grid, sizeof(SgGridDev)));
  HIP_TRY(hipMemcpyAsync(c->d_sg_grid, &d, sizeof d, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->timing) HIP_TRY(hipEventRecord(c->ev0, c->stream));
  HIP_TRY(launch_sg_grid(wf, n, d.NT, L == 16 * d.NT, reinterpret_cast<const SgGridDev*>(c->d_sg_grid), amax, energy, t50_us, blmean, blslope, c->stream));
  c->last_kernel = "sg_grid_kernel";
  if (c->timing) { HIP_TRY(hipEventRecord(c->ev1, c->stream)); c->n_launches = 1; c->n_stages = 1; }
  return LDSP_OK;
}

}  // extern "C"
