/*
 * ldsp.h — C ABI of libldsp_hip.so: the MI355X (gfx950) implementation of the
 * per-waveform filter-chain hot path of legend-exp/LegendDSP.jl (dsp_icpc /
 * dsp_sipm and the filter functors / feature extractors they are built from).
 *
 * The reference has no FFI of its own (pure Julia).  Its operator boundary is
 * the RadiationDetectorDSP filter-functor protocol (fltinstance / rdfilt! /
 * flt_output_length / flt_output_time_axis) plus callable extractor structs,
 * driven by Julia broadcast over an ArrayOfRDWaveforms.  Every entry point
 * below names the reference interface it stands in for (file:line relative to
 * the reference checkout).  A Julia `ccall` / Python `ctypes` binding for each
 * is shown in INTEGRATION.md.
 *
 * Conventions
 *  - All waveform pointers are DEVICE pointers to row-major [n][L] float32
 *    (one trace per row, contiguous — the memory of an ArrayOfSimilarVectors).
 *  - Caller owns input and output buffers; no aliasing between x and y.
 *  - All time windows / filter lengths are passed in SAMPLE units, already
 *    lowered on the host with Julia's round-half-to-even (`round(Int, t/dt)`),
 *    window indices are 0-based inclusive.
 *  - Time axis of a trace is the range t_first + i*dt (reference: the
 *    `time::StepRangeLen` field of RDWaveform), in "time-axis units" (ns for
 *    LEGEND data).  FIR filters are valid-mode with the output stamped at the
 *    time of the LAST input sample under the kernel (assumption A1).
 *  - Every call is asynchronous on the context's stream and returns an int:
 *    0 on success, negative ldsp_status otherwise; never aborts.  The text of
 *    the last error of the calling thread: ldsp_last_error_string().
 */
#ifndef LDSP_H
#define LDSP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDSP_ABI_VERSION 4

typedef enum {
  LDSP_OK = 0,
  LDSP_ERR_INVALID_ARG = -1, /* null pointer, negative size, bad degree ...              */
  LDSP_ERR_WINDOW = -2,      /* window outside the trace: the reference's @assert         */
                             /* (src/tailstats.jl:23-25, src/extremestats.jl:26-28)       */
  LDSP_ERR_HIP = -3,         /* a HIP runtime call failed                                 */
  LDSP_ERR_UNSUPPORTED = -4, /* trace length / filter length outside the built kernels    */
  LDSP_ERR_NOMEM = -5
} ldsp_status;

typedef struct ldsp_ctx ldsp_ctx; /* opaque: device id, stream, parameter staging buffers */

/* ---- context ----------------------------------------------------------- */
int ldsp_abi_version(void);
int ldsp_ctx_create(int device, ldsp_ctx** out);
int ldsp_ctx_destroy(ldsp_ctx* ctx);
/* Launch on an existing hipStream_t (e.g. torch's current stream).  NULL is the
 * device's default (null) stream, exactly as in the HIP API.
 * Lifetime: the stream belongs to the caller and must stay valid while the
 * context launches on it.  A change of stream orders the new stream behind the
 * work queued on the previous one (the context's workspaces are shared by all
 * launches): by an event when the previous stream can still record one, else
 * — the previous stream was destroyed, or is being captured — by a device
 * synchronisation; the new stream is adopted either way, so a context can
 * always leave a stream its owner has already released. */
int ldsp_ctx_set_stream(ldsp_ctx* ctx, void* hip_stream);
/* Back to the context's own non-blocking stream (the initial state). */
int ldsp_ctx_use_own_stream(ldsp_ctx* ctx);
int ldsp_ctx_synchronize(ldsp_ctx* ctx);
const char* ldsp_last_error_string(void);
/* Options: "cusp_direct" = 1 evaluates CUSP/ZAC as direct-form FIR (slow
 * comparator for the closed-form recursions), 0 (default) = recursions.
 * "two_kernel" = 1 runs the CUSP/ZAC stage as a second launch (icpc_cz_kernel)
 * instead of fused into the dsp_icpc launch (the default: icpc_lean3_kernel,
 * three traces per CU, for the standard geometry; the generic icpc_kernel, two
 * traces per CU, whenever CUSP and ZAC share their geometry and its LDS budget
 * allows).
 * "icpc_generic" = 1 runs dsp_icpc on the generic icpc_kernel whatever the
 * geometry (it keeps the eps * T term of the CUSP / ZAC last tap, see
 * ldsp_icpc_run); "sipm_generic" = 1 likewise for dsp_sipm (k_sipm).
 * "dbg_stop" = k stops the kernels after phase k (profiling aid; outputs are
 * then incomplete). */
int ldsp_ctx_set_option(ldsp_ctx* ctx, const char* key, int64_t value);
/* sizeof() of the ABI structs as compiled: 0 icpc_params, 1 icpc_out,
 * 2 sipm_params, 3 sipm_out, 4 trig_out, 5 icpc_opts (binding self-check). */
int64_t ldsp_abi_sizeof(int which);
/* Average duration in ms of the launches issued by the last ldsp_*_run call,
 * measured with hipEvents recorded on the context stream (timing must have
 * been enabled; synchronises on the closing event). */
int ldsp_ctx_enable_timing(ldsp_ctx* ctx, int on);
int ldsp_ctx_last_kernel_ms(ldsp_ctx* ctx, float* ms);
/* Per-kernel split of the last ldsp_icpc_run when it ran as two launches: stage 0 =
 * icpc_kernel, stage 1 = icpc_cz_kernel (CUSP/ZAC).  A fused run and all other calls have
 * one stage (asking for stage 1 is an LDSP_ERR_INVALID_ARG). */
int ldsp_ctx_last_stage_ms(ldsp_ctx* ctx, int stage, float* ms);
/* Name (inside namespace ldsp::, without template arguments) of the dominant kernel the last
 * ldsp_*_run call launched, e.g. "lean3::icpc_lean3_kernel" — the name rocprofv3 reports; a static
 * string, "" before the first launch. */
const char* ldsp_ctx_last_kernel_name(ldsp_ctx* ctx);

/* ---- limits of the built kernels --------------------------------------- */
#define LDSP_MAX_L 32768        /* samples per trace                          */
#define LDSP_MAX_EST_PTS 64     /* PolynomialDNI window points                 */
#define LDSP_MAX_EST_DEG 5
#define LDSP_MAX_SG_PTS 65      /* Savitzky-Golay taps                         */
#define LDSP_MAX_FIR_TAPS 8192  /* generic valid-mode FIR (CUSP/ZAC/SG)        */
#define LDSP_MAX_TRIG 64        /* IntersectMaximum: default slab capacity (ldsp_trig_out.cap = 0) */
#define LDSP_MAX_MULTI 128      /* MultiIntersect: thresholds per trace        */

/* ---- lowered parameter blocks ------------------------------------------ */

/* TrapezoidalChargeFilter(avgtime, gaptime, avgtime2) in samples.
 * reference call sites: src/dsp_icpc.jl:147-161, src/dsp_routines.jl:12. */
typedef struct {
  int32_t navg, ngap, navg2;
} ldsp_trap;

/* CUSPChargeFilter / ZACChargeFilter(sigma, toplen, tau, length, beta)
 * (src/dsp_icpc.jl:167,174) lowered to samples. */
typedef struct {
  double sigma;   /* rt / dt (not rounded)            */
  int32_t flat;   /* round(ft / dt)                   */
  int32_t length; /* round(flt_length / dt) = #taps   */
  double tau;     /* tau / dt                         */
  double beta;    /* scale; reference passes length/dt */
} ldsp_cuspzac;

/* SignalEstimator(PolynomialDNI(degree, length)) (src/dsp_icpc.jl:157,
 * src/dsp_routines.jl:56): npts = round(length/dt). */
typedef struct {
  int32_t npts, degree;
} ldsp_dni;

/* Lowered DSPConfig + pars_filter + tau for dsp_icpc
 * (src/dsp_icpc.jl:62-230, src/types.jl:32-93). */
typedef struct {
  int32_t L;
  int32_t _pad0;
  double t_first; /* first(time) of the input traces, time-axis units (ns) */
  double dt;      /* step(time)                                            */
  double unit_per_us; /* time-axis units per microsecond (1000 for ns)     */

  /* saturation(wvfs, 0, 2^bit_depth - bit_depth)          dsp_icpc.jl:93-95 */
  double sat_low, sat_high;
  /* signalstats(wvfs, bl_window)                           dsp_icpc.jl:102   */
  int32_t bl_from, bl_until;
  /* tailstats / signalstats on tail_window                 dsp_icpc.jl:115,123 */
  int32_t tail_from, tail_until;
  /* InvCRFilter(tau): y = x + pz_c*cumsum(x), pz_c = dt/tau   dsp_icpc.jl:119 */
  double pz_c;
  /* get_t0(wvfs, t0_threshold; flt_pars, mintot)           dsp_icpc.jl:126   */
  ldsp_trap t0_trap;
  int32_t t0_mintot;
  double t0_threshold;
  /* get_t0 on the inverted trace uses the DEFAULT flt_pars dsp_icpc.jl:207   */
  ldsp_trap t0inv_trap;
  int32_t tx_mintot;      /* get_threshold(...; mintot)     dsp_icpc.jl:132-136 */
  /* get_qdrift(wvfs, t0, qdrift_int_length) / lq           dsp_icpc.jl:141,144 */
  ldsp_dni int_est;
  double qdrift_d1, qdrift_d2; /* first/last of the range, in time-axis units */
  double lq_d1, lq_d2;
  /* fixed energy trapezoids 10/4, 5/3, 3/1 us             dsp_icpc.jl:147-154 */
  ldsp_trap trap_fixed[3];
  /* optimised trapezoid + pick-off t50 + rt + ft/2        dsp_icpc.jl:160-164 */
  ldsp_trap trap_opt;
  double trap_pickoff; /* rt + ft/2 in time-axis units */
  ldsp_dni sig_est;
  /* CUSP / ZAC + pick-off t50 + L/2                       dsp_icpc.jl:167-178 */
  ldsp_cuspzac cusp, zac;
  double cusp_pickoff, zac_pickoff; /* flt_length/2 in time-axis units */
  /* SavitzkyGolayFilter(wl, degree, 1) for a_sg, a_60, a_100  dsp_icpc.jl:181-185 */
  int32_t sg_npts[3];
  int32_t sg_degree;
  /* current_window in time-axis units (indices depend on each filter's axis) */
  double cur_left, cur_right;
  /* get_intracePileUp(sgderiv, nsigma, bl_window; mintot) dsp_icpc.jl:189   */
  double intrace_nsigma;
  int32_t intrace_mintot;
  int32_t _pad1;
  double bl_left, bl_right; /* bl_window in time-axis units (dsp_routines.jl:75) */
} ldsp_icpc_params;

/* Output table of dsp_icpc (src/dsp_icpc.jl:210-229), struct-of-arrays of
 * DEVICE pointers, each [n].  Passthrough columns (blfc, timestamp,
 * eventID_fadc, e_fc) and qc_label == -1 never touch the device and are added
 * by the host wrapper.  Times t0..t99, t50_current, t0_inv are in us, as
 * uconvert(u"us", .) yields; drift_time, t_*_max, inTrace_intersect, tail_tau
 * stay in time-axis units; slopes are per time-axis unit. */
typedef struct {
  float *blmean, *blsigma, *blslope, *bloffset;
  float *tailmean, *tailsigma, *tailslope, *tailoffset;
  float *t0, *t10, *t50, *t80, *t90, *t99, *t50_current, *drift_time;
  float *tail_tau, *tail_mean, *tail_sigma;
  float *e_max, *e_min;
  float *e_10410, *e_535, *e_313, *e_10410_inv, *e_313_inv, *t0_inv;
  float *e_trap, *e_cusp, *e_zac;
  float *e_trap_max, *e_cusp_max, *e_zac_max;
  float *t_trap_max, *t_cusp_max, *t_zac_max;
  float *qdrift, *lq;
  float *a_sg, *a_60, *a_100, *a_raw;
  float* inTrace_intersect;
  int32_t* inTrace_n;
  int32_t *n_sat_low, *n_sat_high, *n_sat_low_cons, *n_sat_high_cons;
  /* elements between consecutive traces in every column: 1 (or 0) = separate
   * contiguous columns; LDSP_ICPC_NCOLS = the columns interleave into one
   * row-major [n][48] table (one contiguous block for the multi-GPU gather). */
  int64_t stride;
} ldsp_icpc_out;
#define LDSP_ICPC_NCOLS 48 /* computed 4-byte columns above */

/* Lowered config for dsp_sipm (src/dsp_sipm.jl:47-158). */
typedef struct {
  int32_t L;
  int32_t _pad0;
  double t_first, dt, unit_per_us;
  int32_t trunc_from, trunc_until; /* TruncateFilter(t0_hpge_window)   :94   */
  int32_t sg_npts, sg_degree;      /* SavitzkyGolayFilter(wl, deg, 1)  :99   */
  /* SG pipeline */
  int32_t sg_mintot, sg_maxtot;    /* IntersectMaximum(min_tot,max_tot) :103 */
  double sg_min_thr, sg_max_thr, sg_nsigma;          /* :104-105 */
  double sg_min_dc_thr, sg_max_dc_thr, sg_nsigma_dc; /* :119-120 */
  /* trap pipeline */
  double pz_c;                     /* InvCRFilter(pz_tau): dt/tau      :124  */
  ldsp_trap trap;                  /* TrapezoidalChargeFilter(rt, ft)  :128  */
  int32_t trap_mintot, trap_maxtot;/* IntersectMaximum                 :132  */
  int32_t _pad1;
  double trap_min_thr, trap_max_thr, trap_nsigma;          /* :133-134 */
  double trap_min_dc_thr, trap_max_dc_thr, trap_nsigma_dc; /* :137-138 */
} ldsp_sipm_params;

/* Ragged column = slab [n][cap] + count[n]; the host wrapper compacts to (offsets, values) =
 * VectorOfVectors.  The reference returns EVERY up-crossing (src/intersect_maximum.jl:49-56): count
 * holds the true multiplicity even when it exceeds `cap`, and a caller that finds count > cap runs the
 * affected traces again with slabs of that capacity (two-pass count-then-fill; the Python host does so,
 * `extractors.resolve_overflow`) — nothing is dropped silently.  cap = 0 means LDSP_MAX_TRIG. */
typedef struct {
  int32_t* count; /* [n]                    */
  /* Positions are Float64 like the reference's (src/dsp_sipm.jl:87-88 converts the time axis to Float64; the ragged columns
   * :149-156 are Vector{Float64}): t_first + dt * (index + fraction) composed in double — a float32 time at 2.6e5 ns has an
   * ulp of 0.03 ns.  `max` is a sample value / parabola maximum of the float32 signal. */
  double *x, *x_high, *x_tot;      /* each [n][cap], may be NULL */
  float* max;                      /* [n][cap], may be NULL */
  int32_t cap;    /* slab capacity per trace (row stride); 0 = LDSP_MAX_TRIG */
  int32_t _pad;
} ldsp_trig_out;

typedef struct {
  float *t_max, *t_min, *t_max_lar, *t_min_lar; /* us */
  float *e_max, *e_min, *e_max_lar, *e_min_lar;
  float *blmean, *blsigma, *blslope, *bloffset;
  float *wfmean, *wfsigma, *wfslope, *wfoffset;
  float *threshold, *threshold_DC, *threshold_trap, *threshold_DC_trap;
  ldsp_trig_out trig, trig_DC, trig_trap, trig_DC_trap;
} ldsp_sipm_out;

/* ---- fused routines ------------------------------------------------------ */

/* dsp_icpc(data, config, tau, pars_filter)        src/dsp_icpc.jl:62-230
 * Kernel choice and the one amplitude assumption behind it: the single-launch kernel (closed-form CUSP / ZAC) leaves out the
 * eps * T term of those filters' last tap (eps = 1 - exp(-dt / tau_cusp): dsp_icpc sets that tau to 1e7 us "to switch off CR",
 * src/dsp_icpc.jl:98).  The host admits it only where |w_last| * eps * L < 1e-2 / 65535 (1.5e-7 of full scale A, A = max(|sat_low|, |sat_high|, 65535)):
 * the SATURATION RAILS OF THE PARAMETER BLOCK (a 16-bit range when they are left at zero) are taken as the bound of the samples.
 * float32 input is not clamped to them — traces far outside the configured rails get the dropped term's error scaled by their
 * amplitude / A; option "icpc_generic" (ldsp_ctx_set_option) runs the kernel that keeps the term.  ldsp_ctx_last_kernel_name
 * tells which kernel a call ran. */
int ldsp_icpc_run(ldsp_ctx* ctx, const float* wf, int64_t n,
                  const ldsp_icpc_params* p, const ldsp_icpc_out* out);

/* Per-call variations of the chain, passed explicitly (the context carries no per-launch state: like the reference's
 * immutable functors, a call depends on its arguments only).  Used by dsp_icpc_compressed for the windowed traces
 * (src/dsp_icpc.jl:352-353: shift_waveform.(wvfs_wdw, -bl_stats.mean ./ presum_rate)):
 *   ext_baseline != NULL: subtract ext_baseline_scale * ext_baseline[i] (device pointer, [n]) instead of the trace's own
 *                         signalstats(bl_window).mean; blmean then reports that value.
 *   main_only != 0:       run without the CUSP/ZAC stage: e_cusp, e_zac, e_cusp_max, e_zac_max, t_cusp_max, t_zac_max are
 *                         not written (the windowed traces are shorter than those filters).
 *   in_u16 != 0:          `wf` points to uint16 ADC counts ([n][L], the element type of production waveforms) instead of
 *                         float32: the kernel converts them as it loads them (the reference's shift_waveform promotes the
 *                         samples to float the same way, src/dsp_icpc.jl:105); no separate cast pass, half the bytes read.
 * opts == NULL is ldsp_icpc_run. */
typedef struct {
  const float* ext_baseline;
  double ext_baseline_scale;
  int32_t main_only;
  int32_t in_u16;
} ldsp_icpc_opts;
int ldsp_icpc_run_opts(ldsp_ctx* ctx, const float* wf, int64_t n, const ldsp_icpc_params* p,
                       const ldsp_icpc_opts* opts, const ldsp_icpc_out* out);
/* Host-only: runs the complete lowering of a parameter block (the checks of the reference's window / filter
 * arguments, filter constants and taps, launch geometry) and returns what ldsp_icpc_run would return for it before
 * touching the device.  Needs no context and no GPU (config validation on a login node; sanitizer runs of the host code). */
int ldsp_icpc_check_params(const ldsp_icpc_params* p);

/* ---- "next" row 1: trapezoid filter-optimisation grid scans -----------
 * dsp_trap_rt_optimization (src/dsp_filter_optimization.jl:102-133: pick-off at a fixed
 * time, enc_pickoff_trap) and dsp_trap_ft_optimization (:241-274: pick-off at
 * t50 + rt + ft/2, t50 = first crossing of half the maximum of the pole-zero corrected
 * trace).  One read of each trace: baseline subtraction (signalstats mean), InvCRFilter,
 * prefix sum, then for every grid point g the SignalEstimator of the trapezoid output
 * at the pick-off — G results per trace from G x npts trapezoid samples. */
#define LDSP_MAX_GRID 64
typedef struct ldsp_trapgrid_params {
  int32_t L, _pad0;
  double t_first, dt;          /* shared time axis of the traces                        */
  int32_t bl_from, bl_until;   /* signalstats window (0-based samples, inclusive)       */
  double pz_c;                 /* dt / tau of the InvCRFilter                           */
  ldsp_dni sig_est;            /* SignalEstimator(PolynomialDNI(degree, npts))          */
  int32_t pick_mode;           /* 0: pick-off at `pick_time`; 1: at t50 + offsets[g]    */
  int32_t tx_mintot;           /* Intersect(mintot) in samples (mode 1)                 */
  double pick_time;            /* mode 0, time-axis units                               */
} ldsp_trapgrid_params;
/* traps[G] in samples; offsets[G] (time-axis units, mode 1; may be NULL in mode 0); out [G][n] float (device). */
int ldsp_trap_grid_run(ldsp_ctx* ctx, const float* wf, int64_t n, const ldsp_trapgrid_params* p, int32_t G,
                       const ldsp_trap* traps, const double* offsets, float* out);

/* The same scan for an arbitrary FIR filter per grid point — dsp_cusp_rt_optimization / dsp_zac_rt_optimization
 * (src/dsp_filter_optimization.jl:145-181, 193-229) and the *_ft_optimization pair (:286-324, 336-374): taps[g] =
 * the Lf coefficients of grid point g (HOST pointer, [G][Lf] doubles, e.g. from ldsp_cusp_coeffs / ldsp_zac_coeffs;
 * valid-mode, trailing time axis, as ldsp_rdfilt_fir).  Only the npts outputs under the SignalEstimator window are
 * evaluated (direct form, npts x Lf multiply-adds per grid point and trace).  p: as for ldsp_trap_grid_run. */
int ldsp_fir_grid_run(ldsp_ctx* ctx, const float* wf, int64_t n, const ldsp_trapgrid_params* p, int32_t G, int32_t Lf,
                      const double* taps, const double* offsets, float* out);

/* dsp_sg_optimization (src/dsp_filter_optimization.jl:393-441): baseline statistics, pole-zero, t50, the energy
 * e = SignalEstimator(Trapezoid(rt, ft) output, t50 + rt + ft/2), and for every window length of the grid the
 * interpolated maximum (get_wvf_maximum) of the Savitzky-Golay derivative inside the current window.
 * p->pick_mode must be 1; trap_offset = rt + ft/2 (time units).  npts[W], from[W], until[W]: SG points and the
 * current window on each filter's own output axis (0-based samples).  Outputs (device, any may be NULL):
 * amax [W][n], energy / t50_us / blmean / blslope [n]; A/E = amax / energy is the caller's division.
 * W = 0 gives dsp_qc_flt_optimization without a classifier (:31-63): energy, blmean, blslope, t50. */
int ldsp_sg_grid_run(ldsp_ctx* ctx, const float* wf, int64_t n, const ldsp_trapgrid_params* p, const ldsp_trap* trap,
                     double trap_offset, double unit_per_us, int32_t W, const int32_t* npts, int32_t degree, const int32_t* from,
                     const int32_t* until, float* amax, float* energy, float* t50_us, float* blmean, float* blslope);

/* BASELINE config 2: the e_10410 column path only — signalstats(bl) -> shift
 * -> InvCRFilter -> TrapezoidalChargeFilter(10us,4us) -> maximum
 * (src/dsp_icpc.jl:102-105,119-120,147-148).  Writes blmean[n], e_10410[n]. */
int ldsp_icpc_pz_trap_run(ldsp_ctx* ctx, const float* wf, int64_t n,
                          const ldsp_icpc_params* p, float* blmean,
                          float* e_10410);
/* The same on uint16 ADC counts ([n][L]), converted while loading: 2L instead of 4L bytes per trace for this HBM-bound
 * sub-chain; results bit-identical to the float32 entry (every uint16 is exact in float32). */
int ldsp_icpc_pz_trap_run_u16(ldsp_ctx* ctx, const uint16_t* wf, int64_t n,
                              const ldsp_icpc_params* p, float* blmean,
                              float* e_10410);

/* dsp_sipm(data, config, pars_optimization)       src/dsp_sipm.jl:47-158 */
int ldsp_sipm_run(ldsp_ctx* ctx, const float* wf, int64_t n,
                  const ldsp_sipm_params* p, const ldsp_sipm_out* out);
/* The same on uint16 ADC counts ([n][L]): converted to float as the kernel loads them (src/dsp_sipm.jl:87-88 promotes
 * the samples the same way), no separate cast pass. */
int ldsp_sipm_run_u16(ldsp_ctx* ctx, const uint16_t* wf, int64_t n,
                      const ldsp_sipm_params* p, const ldsp_sipm_out* out);

/* ---- filter functors: rdfilt!(y, fltinstance(flt, si), x) ---------------- */
/* Output length of each filter = what flt_output_length(fi) returns. */

/* InvCRFilter(tau): y[i] = x[i] + c*sum_{j<=i} x[j], c = dt/tau
 * (RadiationDetectorDSP; call sites src/dsp_icpc.jl:119, src/dsp_sipm.jl:124) */
int ldsp_rdfilt_invcr(ldsp_ctx*, const float* x, int64_t n, int32_t L, double c, float* y);
/* IntegratorFilter(gain): y = gain*cumsum(x)   (src/dsp_routines.jl:53) */
int ldsp_rdfilt_integrator(ldsp_ctx*, const float* x, int64_t n, int32_t L, double gain, float* y);
/* TrapezoidalChargeFilter: Lout = L-(navg+ngap+navg2)+1 */
int ldsp_rdfilt_trap(ldsp_ctx*, const float* x, int64_t n, int32_t L, ldsp_trap t, float* y);
/* Generic valid-mode FIR y[k] = sum_j h[j]*x[k+ntaps-1-j] (true convolution):
 * CUSPChargeFilter, ZACChargeFilter, SavitzkyGolayFilter.  h is a HOST
 * pointer (coefficients from ldsp_*_coeffs). Lout = L-ntaps+1. */
int ldsp_rdfilt_fir(ldsp_ctx*, const float* x, int64_t n, int32_t L, const double* h, int32_t ntaps, float* y);
/* DerivativeFilter(gain)           src/derivative.jl:47-55 */
int ldsp_rdfilt_derivative(ldsp_ctx*, const float* x, int64_t n, int32_t L, double gain, float* y);
/* HaarAveragingFilter(ds): Lout = ceil(L/ds)   src/haar_filter.jl:26-39 */
int ldsp_rdfilt_haar(ldsp_ctx*, const float* x, int64_t n, int32_t L, int32_t ds, float* y);
/* MovingWindowFilter(length)       src/moving_window_multi.jl:99-116 */
int ldsp_rdfilt_moving_window(ldsp_ctx*, const float* x, int64_t n, int32_t L, int32_t len, float* y);
/* MovingWindowMultiFilter(length)  src/moving_window_multi.jl:118-129 */
int ldsp_rdfilt_moving_window_multi(ldsp_ctx*, const float* x, int64_t n, int32_t L, int32_t len, float* y);
/* shift_waveform / multiply_waveform / reverse_waveform / TruncateFilter in one:
 * y[i] = scale*x[src(i)] + shift for i in [0, until-from], src(i) = from+i, or
 * until-i when reverse != 0  (src/dsp_icpc.jl:105,199, src/dsp_routines.jl:79,
 * src/dsp_sipm.jl:94).  shift_per_trace may be NULL, else shift += it[trace]. */
int ldsp_rdfilt_affine(ldsp_ctx*, const float* x, int64_t n, int32_t L, int32_t from, int32_t until,
                       double scale, double shift, const float* shift_per_trace, int32_t reverse, float* y);

/* QC classifier front end: get_qc_classifier / get_qc_classifier_compressed (src/dsp_ml_routines.jl:9-24, 45-60;
 * with a DSPConfig :26-34, :62-70): optional signalstats(bl).mean + shift_waveform, HaarAveragingFilter(2) applied
 * `levels` times (5, compressed: 2), division by max(|min|, |max|) (0 -> 1).  features: device [n][Lout],
 * Lout = ldsp_qc_features_len(L, levels) — the memory of flatview(VectorOfSimilarArrays(signal)) that the
 * reference hands to f_evaluate_qc (src/ml.jl:6-22: LIBSVM svmpredict).  bl_from < 0: no baseline subtraction
 * (the trace is already shifted, src/dsp_icpc.jl:105-108).  norm: optional device [n], the divisor. */
int32_t ldsp_qc_features_len(int32_t L, int32_t levels);
int ldsp_qc_features(ldsp_ctx*, const float* x, int64_t n, int32_t L, int32_t levels, int32_t bl_from, int32_t bl_until,
                     float* features, float* norm);

/* Host-side coefficient builders (double, length = p->length or npts). */
int ldsp_cusp_coeffs(const ldsp_cuspzac* p, double* h);
int ldsp_zac_coeffs(const ldsp_cuspzac* p, double* h);
int ldsp_sg_coeffs(int32_t npts, int32_t degree, int32_t derivative, double* h);

/* ---- feature extractors -------------------------------------------------- */
/* signalstats(wf, start, stop) -> (mean, sigma, slope, offset)
 * (RadiationDetectorDSP; call sites src/dsp_icpc.jl:102,123). */
int ldsp_signalstats(ldsp_ctx*, const float* x, int64_t n, int32_t L, int32_t from, int32_t until,
                     double t_first, double dt, float* mean, float* sigma, float* slope, float* offset);
/* tailstats -> (mean, sigma, tau)            src/tailstats.jl:13-72 */
int ldsp_tailstats(ldsp_ctx*, const float* x, int64_t n, int32_t L, int32_t from, int32_t until,
                   double t_first, double dt, float* mean, float* sigma, float* tau);
/* extremestats -> (min, max, tmin, tmax)     src/extremestats.jl:14-40 */
int ldsp_extremestats(ldsp_ctx*, const float* x, int64_t n, int32_t L, int32_t from, int32_t until,
                      double t_first, double dt, float* vmin, float* vmax, float* tmin, float* tmax);
/* thresholdstats / thresholdstats_mad        src/thresholdstats.jl:14-71 */
int ldsp_thresholdstats(ldsp_ctx*, const float* x, int64_t n, int32_t L, double lo, double hi, float* sigma);
int ldsp_thresholdstats_mad(ldsp_ctx*, const float* x, int64_t n, int32_t L, double lo, double hi, float* mad);
/* saturation -> (low, high, max_cons_low, max_cons_high)  src/saturation.jl:12-65 */
int ldsp_saturation(ldsp_ctx*, const float* x, int64_t n, int32_t L, int32_t from, int32_t until,
                    double low, double high, int32_t* n_low, int32_t* n_high, int32_t* cons_low, int32_t* cons_high);
/* get_wvf_maximum                            src/interpolation.jl:21-46 */
int ldsp_get_wvf_maximum(ldsp_ctx*, const float* x, int64_t n, int32_t L, int32_t from, int32_t until, float* vmax);
/* Intersect(mintot)(wf, thr) -> (x, multiplicity); thr per trace [n]
 * (RadiationDetectorDSP; call sites src/dsp_routines.jl:18,35,74). x = NaN if none. */
int ldsp_intersect(ldsp_ctx*, const float* x, int64_t n, int32_t L, double t_first, double dt,
                   const float* thr, int32_t min_n, float* xout, int32_t* mult);
/* IntersectMaximum(mintot, maxtot)(wf, thr)  src/intersect_maximum.jl:18-119 */
int ldsp_intersect_maximum(ldsp_ctx*, const float* x, int64_t n, int32_t L, double t_first, double dt,
                           const float* thr, int32_t min_n, int32_t max_n, const ldsp_trig_out* out);
/* MultiIntersect(ratios, mintot, n, d, rate)(wf) -> x[K]   src/multi_intersect.jl:26-104
 * ratios: HOST pointer [K]; xout: device [n][K]. status[n] != 0 where the
 * reference's boundary @assert (src/multi_intersect.jl:75-78) would fire, or where the
 * fit window of a threshold in between leaves the trace (the reference reads out of
 * bounds there under @inbounds, :88-92). */
int ldsp_multi_intersect(ldsp_ctx*, const float* x, int64_t n, int32_t L, double t_first, double dt,
                         const double* ratios, int32_t K, int32_t min_n, int32_t half_n, int32_t degree,
                         int32_t rate, float* xout, int32_t* status);
/* SignalEstimator(PolynomialDNI(degree, npts))(wf, t); t per trace [n] in time-axis units */
int ldsp_signal_estimator(ldsp_ctx*, const float* x, int64_t n, int32_t L, double t_first, double dt,
                          const float* t, ldsp_dni est, float* out);

#ifdef __cplusplus
}
#endif
#endif /* LDSP_H */
