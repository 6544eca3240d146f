// icpc_dev.hpp — device-resident parameter block of the fused dsp_icpc kernel.
// Built on the host from ldsp_icpc_params (ldsp_api.cpp: lower_icpc_dev) and
// uploaded once per run call; read through scalar loads in the kernel.
#pragma once
#include <stdint.h>
#include "../../include/ldsp.h"

namespace ldsp {

constexpr int SPT_HOST = 32;  // samples per thread of the trace kernels (== ldsp::SPT)

struct TrapDev {
  int32_t n1, g, n2, flen;  // navg, ngap, navg2, total length
  float inv1, inv2;         // 1/navg, 1/navg2
  float rr, navg;           // inv2/inv1 and 1/inv1: the sweeps evaluate the trapezoid unscaled
};

// LSQ polynomial estimator (PolynomialDNI): yhat(u) = sum_i y[i0+i] * sum_j B[i][j] u^j,
// u = (p - i0 - c) * s_inv  (centred, scaled local coordinate)
struct EstDev {
  int32_t npts, deg;
  float c, s_inv;
  float B[LDSP_MAX_EST_PTS * (LDSP_MAX_EST_DEG + 1)];
};

// window statistics in index space: xi = i - ic, sum xi = 0, var_xi = (n^2-1)/12
struct WinDev {
  int32_t from, until;
  double ic, inv_n, var_i;
};

// Closed-form CUSP / ZAC: all constants in float,
// derived in double on the host.
struct CuspZacDev {
  int32_t Lf, lt, flat, f1, ltp;  // taps, rise length, flat, first fall tap, fall length Lf-f1
  int32_t is_zac;
  // q = exp(-1/sigma).  Every decay factor a kernel multiplies by is an exactly
  // rounded power from these tables (a float q re-multiplied 1000x drifts by 3e-5):
  float qp1[5];                    // q^e,    e = 0..4  (inside a 4-sample chunk)
  float qp4[65];                   // q^(4j), j = 0..64 (across the lanes of a wave; qp4[64] = one wave-row)
  float qpw[65];                   // q^(256j), j = 0..64 (across wave-rows)
  float eps;                       // 1 - exp(-1/tau)
  float sc_half_den;               // beta/Lf / (2 sinh(lt/sigma))
  float sc;                        // beta/Lf
  float q_lt, q_mlt1;              // q^lt, q^-(lt-1)
  float q_ltp1, q_mltp;            // q^(ltp-1), q^-ltp
  float q1, q2;                    // q, q^2
  float w_last;                    // sc * shape[Lf-1] (the tap that multiplies y[k] directly)
  float rho_sc;                    // ZAC: sc * acusp/apar (multiplies the parabola part)
  // ZAC parabola part = double prefix sum of u[n] = sum_e zu_coef[e] * (Dp[n - zu_shift[e]] - Dp[n - zu_shift_b[e]]).
  // The second difference of the parabola taps weights Dp at nine shifts with coefficients of +-lt whose sum is zero; taken
  // tap by tap, each product carries the rounding of lt * |Dp| (Dp keeps the level of the first sample for the whole trace)
  // and the double prefix sum integrates that noise twice.  Summed by parts instead (Abel): differences of Dp between
  // neighbouring shifts, which are small wherever the trace is flat, times the running sum of the coefficients.
  int32_t zu_n;
  int32_t zu_shift[12], zu_shift_b[12];
  float zu_coef[12];
  // the same sum as ONE chain over all shifts s_0 < s_1 < ... (links with a zero running sum included): a kernel that keeps
  // Dp[n - s_e] in registers reads every shift once — u[n] = sum_e zc_r[e] * (Dp[n - zc_s[e]] - Dp[n - zc_s[e+1]])
  int32_t zc_n;          // number of shifts (links = zc_n - 1)
  int32_t zc_s[13];
  float zc_r[12];
  // the same chain with the parabola's LAST tap folded in (icpc_lean3.hip, CUSP and ZAC sharing their geometry): that tap multiplies
  // y[k] = Dp[k] + const directly; as a term of the double prefix sum it is par[Lf-1] * (Dp[n-Lf+1] - 2 Dp[n-Lf] + Dp[n-Lf-1]),
  // three more taps at shifts the chain already has or next to them.  The constant goes to the result: wl_fold * (y[0] - pivot)
  int32_t zf_n;
  int32_t zf_s[13];
  float zf_r[12];
  // sum of the direct-form taps: the filter's response to a constant level.  The closed form runs on y - c (c = the level at
  // the left edge of the pick-off window) and adds c * hsum back at the end, see cz_body / icpc_lean.hip phase 7
  double hsum;
};

// trapezoid grid scan (ldsp_trap_grid_run): one parameter block in device memory
struct TrapGridDev {
  int32_t L, NT, G, pick_mode, tx_mintot;
  float t_first, dt;
  WinDev bl;
  float pz_c;
  double pz_c64;
  EstDev est;
  int32_t pick_ip;   // mode 0: pick-off position in samples, split int + frac
  float pick_fp;
  TrapDev trap[64];
  float offs[64];    // mode 1: offsets in samples
};

// FIR grid scan (ldsp_fir_grid_run)
struct FirGridDev {
  int32_t L, NT, G, Lf, pick_mode, tx_mintot;
  float t_first, dt;
  WinDev bl;
  float pz_c;
  double pz_c64;
  EstDev est;
  int32_t pick_ip;
  float pick_fp;
  float offs[64];
  const float* taps;   // device, [G][Lf], taps REVERSED (correlation form: out[k] = sum_j c[j] y[k+j])
  int32_t same_offs;   // every grid point picks off at the same position (always so for the CUSP / ZAC scans)
};

// SG window-length grid scan (ldsp_sg_grid_run)
struct SgGridDev {
  int32_t L, NT, W, tx_mintot;
  float t_first, dt, inv_unit_per_us;
  WinDev bl;
  float pz_c;
  double pz_c64;
  EstDev est;
  TrapDev trap;
  float trap_off;            // rt + ft/2 in samples
  int32_t np[32], from[32], until[32];
  float c[32][LDSP_MAX_SG_PTS];   // correlation taps per grid point
};

struct IcpcDev {
  int32_t L, NT, R;   // trace length; threads and float4 rows per thread of the launch
  float t_first, dt, unit_per_us, inv_unit_per_us;
  float sat_low, sat_high;
  WinDev bl, tail, sgbl;
  float pz_c;
  double pz_c64;
  TrapDev t0, t0inv, fixed[3], opt;
  int32_t t0inv_same;  // t0inv trapezoid == t0 trapezoid (reuse by linearity)
  int32_t t0_mintot, tx_mintot, intrace_mintot;
  float t0_thr, intrace_nsigma;
  EstDev int_est, sig_est;
  float qdrift_d1, qdrift_d2, lq_d1, lq_d2;       // in samples
  float trap_pickoff, cusp_pickoff, zac_pickoff;  // in samples
  // Savitzky-Golay: correlation-form coefficients c[i] (out[k] = sum_i c[i] y[k+i])
  int32_t sg_npts[3];
  int32_t sg_same_02;  // a_100 filter == a_sg filter
  float sg_c[3][LDSP_MAX_SG_PTS];
  int32_t cur_from[4], cur_until[4];  // current window on the axes of sg[0..2] and of the plain derivative
  CuspZacDev cusp, zac;
  int32_t cz_shared;  // cusp and zac share sigma/flat/length/tau: one set of recursions
  int32_t cusp_mode;  // 0 = direct-form FIR (comparator), 1 = closed-form recursions
  int32_t dbg_stop;   // profiling aid: return after phase N (0 = run everything)
  int32_t in_u16;     // the traces are uint16 ADC counts (ldsp_icpc_opts.in_u16): converted to float as they are loaded
  const float* h_cusp; // device, true-convolution taps (mode 0)
  const float* h_zac;
  // lean kernel: how window w (0 bl, 1 tail, 2 SG baseline, 3 current window of SG filter 0, 4 union of the other current
  // windows, 5 their intersection) meets the four 256-sample rows of wave v in the S4 view: bit r = row r lies outside the window,
  // bit 4+r = wholly inside (else it holds an edge)
  uint32_t rowcls[6][16];
  long long* dbg_stamps;   // diagnostic builds (LDSP_STAMPS) only: [LDSP_STAMP_BLOCKS][16 waves][LDSP_STAMP_SLOTS] s_memtime stamps
};
#define LDSP_STAMP_BLOCKS 2048
#define LDSP_STAMP_SLOTS 32

struct IcpcOutDev {
  void* col[LDSP_ICPC_NCOLS];
  int64_t stride;  // elements between consecutive traces (1 = SoA columns, 48 = one [n][48] table)
};

// column indices = order of ldsp_icpc_out
enum IcpcCol {
  C_blmean, C_blsigma, C_blslope, C_bloffset, C_tailmean, C_tailsigma, C_tailslope, C_tailoffset,
  C_t0, C_t10, C_t50, C_t80, C_t90, C_t99, C_t50_current, C_drift_time,
  C_tail_tau, C_tail_mean, C_tail_sigma, C_e_max, C_e_min,
  C_e_10410, C_e_535, C_e_313, C_e_10410_inv, C_e_313_inv, C_t0_inv,
  C_e_trap, C_e_cusp, C_e_zac, C_e_trap_max, C_e_cusp_max, C_e_zac_max,
  C_t_trap_max, C_t_cusp_max, C_t_zac_max, C_qdrift, C_lq,
  C_a_sg, C_a_60, C_a_100, C_a_raw, C_inTrace_intersect, C_inTrace_n,
  C_n_sat_low, C_n_sat_high, C_n_sat_low_cons, C_n_sat_high_cons, C_NCOLS
};
static_assert(C_NCOLS == LDSP_ICPC_NCOLS, "column count");

}  // namespace ldsp
