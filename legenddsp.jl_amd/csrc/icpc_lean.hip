// icpc_lean.hip — the fused dsp_icpc kernel for the standard geometry, written against the measured issue model of
// gfx950: a wave issues one instruction every ~8-9 cycles whatever its kind (tools/micro/valu_rate3.hip), four waves per
// SIMD are resident (two 79 KB traces per CU), so the run time of this chain is set by the NUMBER of instructions a wave
// executes.  Everything here serves that count:
//   * packed FP32 (v_pk_add/mul/fma_f32 on aligned register pairs): the two halves of an S4 quad, or the two rows of an
//     LS row pair (ds_read2st64_b32 delivers exactly such a pair), are one instruction instead of two;
//   * independent DPP reductions run interleaved (no s_nop between dependent steps);
//   * wave-uniform classification of every window against the wave's own 256-sample rows: no per-lane exec masks in rows
//     that lie wholly inside or outside a window;
//   * per-wave partials in plain LDS arrays instead of atomics on pre-initialised slots (one barrier less);
//   * the scalar finishing work (statistics, parabolas, interpolations) is spread over different waves, in float32.
// It computes the same 48 columns as icpc_kernel<NT, 4, true, true> (reference src/dsp_icpc.jl:62-230) and is used when
// the trace fills the tile (L = 16 NT), CUSP and ZAC share their geometry (closed form), the three Savitzky-Golay windows
// have at most M taps, the inverted t0 uses the same trapezoid and tx_mintot <= 2 samples; every other configuration runs
// icpc_kernel (option "icpc_generic" forces it: the comparator of tests/test_icpc_gpu.py).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include <stddef.h>
#include "icpc_dev.hpp"
#include "ldsp_device.hpp"
#include "qdrift.hpp"

// STAMP(id): phase boundary.  Always leaves a comment in the assembly (tools read the phases off it, it costs nothing); in
// diagnostic builds (LDSP_STAMPS) every wave of the first blocks also writes s_memtime there (tools/stamp_map.py).
#ifdef LDSP_STAMPS
#define STAMP(id) do { asm volatile("; LDSP_PHASE " #id); if ((threadIdx.x & 63) == 0 && blockIdx.x < LDSP_STAMP_BLOCKS && P.dbg_stamps) \
    P.dbg_stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * LDSP_STAMP_SLOTS + (id)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(id) asm volatile("; LDSP_PHASE " #id)
#endif
// diagnostic builds only (tools/prof_phases_lean.sh): option dbg_stop = 100 + k ends the kernel after stamp k, so that the
// PMC instruction counters can be read phase by phase
#ifdef LDSP_DSTOP
#define DSTOP(id) do { if (P.dbg_stop == 100 + (id)) return; } while (0)
#else
#define DSTOP(id) do { } while (0)
#endif

namespace ldsp {
int g_dbg_lds_pad = 0;   // option "dbg_lds_pad": extra dynamic LDS per workgroup (occupancy experiments, tools/occ_probe.py)
namespace lean {

typedef __attribute__((address_space(3))) float lds_float;   // explicit LDS pointers: survive being pinned to a VGPR (ds_ instructions, not flat_)
typedef __attribute__((address_space(3))) float __attribute__((ext_vector_type(4))) lds_f4;
// the NW per-wave partials of one quantity, summed / folded in wave order: for eight waves two 16-byte reads instead of eight 4-byte ones
template <int NW, typename F>
__device__ __forceinline__ float fold_partials(const lds_float* p, float init, F f) {
  float acc = init;
  if constexpr (NW == 8) {
    const auto a = *(const lds_f4*)p, b = *(const lds_f4*)(p + 4);
    acc = f(f(f(f(f(f(f(f(acc, a.x), a.y), a.z), a.w), b.x), b.y), b.z), b.w);
  } else {
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) acc = f(acc, p[ww]);
  }
  return acc;
}
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int R = 4, SP = 16;
constexpr int EST_TBL = LDSP_MAX_EST_PTS * (LDSP_MAX_EST_DEG + 1);
enum { M_T0, M_T0INV, M_INTR, M_FB, M_SG50 = 8, NMASKROWS = 9 };   // M_FB..M_FB+4: the general scan of the five y thresholds (rare)
enum { W_TAIL = 0, W_SGB = 3, W_PZ = 5, W_CZ = 8, NWSUM = 10 };   // rows of the per-wave window partial sums

__device__ __forceinline__ f2 mk2(float a, float b) { f2 v; v.x = a; v.y = b; return v; }
__device__ __forceinline__ f2 splat(float a) { return mk2(a, a); }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float hsum(f2 v) { return v.x + v.y; }
// keeps a value where it is computed: hipcc otherwise sinks the arithmetic of an accumulator to its next use (the following
// phase) and the loads feeding it stay live across a barrier — in registers that are not there
__device__ __forceinline__ void pin(f2& v) { asm volatile("" : "+v"(v)); }
// arg-max over the wave, value first: the wave maximum of v (DPP chain, lane 63), then the smallest index among the lanes
// that hold it (findmax: first occurrence).  Both results are valid in lane 63.
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
  const float own = v;
  LDSP_DPP_GROUP1("v_max_f32_dpp", v);
  const float vm = readlane_f(v, 63);
  uint32_t key = (own == vm) ? (uint32_t)i : 0x7fffffffu;
  LDSP_DPP_GROUP1("v_min_u32_dpp", key);
  i = (int)key;
}

struct Pos {  // fractional sample position ip + fp
  int ip;
  float fp;
};
__device__ __forceinline__ Pos pos_norm(Pos p) {
  const float f = floorf(p.fp);
  p.ip += (int)f;
  p.fp -= f;
  return p;
}
__device__ __forceinline__ Pos pos_add(Pos p, float d) {
  const float di = floorf(d);
  p.ip += (int)di;
  p.fp += d - di;
  return pos_norm(p);
}
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
  const float a = T[k + t.flen] - T[k + t.n1 + t.g];
  const float b = T[k + t.n1] - T[k];
  return a * t.inv2 - b * t.inv1;
}
__device__ __forceinline__ float est_weight(const EstDev& E, const float* Bt, int l, float u) {
  return dni_weight(E, Bt, l, u);   // qdrift.hpp: the whole coefficient row in one read
}
// window [i0, i0 + npts) and local coordinate u of the LSQ estimate at position p in a signal of nsig samples (A3)
__device__ __forceinline__ void est_window(const EstDev& E, Pos p, int nsig, int* i0, float* u) {
  if (p.ip < 0) { p.ip = 0; p.fp = 0.f; }
  if (p.ip >= nsig - 1) { p.ip = nsig - 1; p.fp = 0.f; }
  int a = p.ip + (int)ceilf(p.fp - 0.5f * (float)E.npts);
  a = max(0, min(a, nsig - E.npts));
  *i0 = a;
  *u = ((float)(p.ip - a) + p.fp - E.c) * E.s_inv;
}
__device__ __forceinline__ float wave_total(float v) {   // sum over the wave, in every lane (one chain: nops between the steps)
  LDSP_DPP_GROUP1("v_add_f32_dpp", v);
  return readlane_f(v, 63);
}

// (mean, sigma, slope per time unit, offset) of a window from its sums about a pivot — the arithmetic of signalstats
// (restated in oracle/ldsp_oracle.c:orc_signalstats); float32: the sums carry the spread of the window, not its level
__device__ __forceinline__ void win_finish(float s1, float s2, float sx, const WinDev& w, float pivot, float t_first, float dt,
                                           float* mean, float* sigma, float* slope, float* offset) {
  const float inv_n = (float)w.inv_n;
  const float md = s1 * inv_n, m = pivot + md;
  const float var = fmaxf(fmaf(s2, inv_n, -md * md), 0.f);
  const float sl = (sx * inv_n) * __builtin_amdgcn_rcpf((float)w.var_i * dt);   // v_rcp / v_sqrt: 1 ulp, no refinement sequences
  *mean = m;
  *sigma = __builtin_amdgcn_sqrtf(var);
  *slope = sl;
  *offset = m - sl * (t_first + (float)w.ic * dt);
}

struct Slots {   // LDS atomics targets (set to their identities in phase 0)
  unsigned long long vi[8];   // packed (value, index) maxima: optimised trapezoid, 4 current windows
  uint32_t fmx[8];            // float maxima as ordered uints: 3 fixed trapezoids, SG maximum, 2 inverted, cusp, zac
  int isum[4];                // tail_bad, run counts of t0 / inverted t0 / in-trace pile-up
  int imin[10];               // first index: 5 thresholds, t0, inverted t0, sg50, cusp max, zac max
  int imax[2];                // last run end (pile-up)
};
enum { VI_OPT, VI_CUR0, VI_CUR1, VI_CUR2, VI_CUR3 };
enum { FX_F0, FX_F1, FX_F2, FX_G, FX_F0I, FX_F2I, FX_CUSP, FX_ZAC };
enum { IS_TAILBAD, IS_T0, IS_T0INV, IS_INTR };
enum { IM_TX0 = 0, IM_T0 = 5, IM_T0INV = 6, IM_SG50 = 7, IM_CUSP = 8, IM_ZAC = 9 };

template <int NT>
struct Smem {
  static constexpr int NW = NT / 64, Lp = NT * SP, NWORDS = Lp / 32;
  // order in memory: [gap: mask words][B][A][small arrays] — a lane-strided read T[k + shift] of a row past the end of B runs
  // into A (finite samples, masked by the caller) instead of past the allocation
  uint32_t* bm;    // [NMASKROWS][NWORDS] in the gap in front of B (zero-filled as Dp[i < 0] by the CUSP/ZAC stage)
  float* B;        // [Lp + 64]  SG output, then T, then Dp / G / A of the CUSP/ZAC stage
  float* A;        // [Lp]       y (S4 stores, LS reads); u of the ZAC stage later
  float* part;     // [2][R*NW]  wave-row totals of the block scans (alternating buffers)
  // The small arrays lie beyond the 64 KiB an LDS instruction's immediate offset reaches from address 0, and hipcc rebuilds
  // every one of their (uniform) addresses as s_add + v_mov in front of the access.  They are addressed instead from ONE
  // base that is pinned to a VGPR (pin_small): base + constant folds into the instruction's offset field.
  lds_float* wred;     // [5][NW]    phase-1 per-wave partials: s1, s2, sx, max, min
  lds_float* wsum;     // [NWSUM][NW]
  double* dpart;   // [2][R*NW]  double prefix sum of the ZAC parabolas
  Slots* sl;
  lds_float* outv;     // [C_NCOLS]
  lds_float* misc;     // [32]
  float* estB;     // [2][EST_TBL]
  static constexpr int gap_floats(int cz_pad) { return NMASKROWS * NWORDS > cz_pad ? NMASKROWS * NWORDS : cz_pad; }
  static constexpr size_t bytes(int cz_pad) {
    return (size_t)(2 * Lp + 64 + gap_floats(cz_pad)) * 4 + (2 * R * NW + 5 * NW + NWSUM * NW) * 4 + 2 * R * NW * 8 + sizeof(Slots) +
           (C_NCOLS + 32 + 2 * EST_TBL) * 4 + 64;
  }
  __device__ Smem(unsigned char* raw, int cz_pad) {
    bm = reinterpret_cast<uint32_t*>(raw);
    B = reinterpret_cast<float*>(raw) + gap_floats(cz_pad);
    A = B + Lp + 64;
    dpart = reinterpret_cast<double*>(A + Lp);
    part = reinterpret_cast<float*>(dpart + 2 * R * NW);
    float* wred_ = part + 2 * R * NW;
    float* wsum_ = wred_ + 5 * NW;
    sl = reinterpret_cast<Slots*>(wsum_ + NWSUM * NW);
    float* outv_ = reinterpret_cast<float*>(sl + 1);
    estB = outv_ + C_NCOLS + 32;
    lds_float* base = (lds_float*)wred_;
    asm volatile("" : "+v"(base));   // pin_small: one VGPR base for the arrays below
    wred = base;
    wsum = base + (wsum_ - wred_);
    outv = base + (outv_ - wred_);
    misc = outv + C_NCOLS;
  }
};

__host__ __device__ inline int cz_pad_floats(int Lf) { return (Lf + 2 + 7) & ~3; }

// Accumulators of a window's sums in the S4 view, as register pairs: per quad  t = d01 + d23,
//   s1 += t,  sr += r t (row weight),  s2 += d d,  se += e d (element weight 0..3).
// sum d xi  with  xi = 4 (tid + NT r) + e - ic  is then  (4 tid - ic) s1 + 4 NT sr + se.
struct WAcc {
  f2 s1, s2, se, sr;
};
__device__ __forceinline__ void wacc_quad(WAcc& a, f2 d0, f2 d1, int r) {
  const f2 t = d0 + d1;
  a.s1 += t;
  if (r) a.sr = fma2(t, splat((float)r), a.sr);
  a.s2 = fma2(d0, d0, a.s2);
  a.s2 = fma2(d1, d1, a.s2);
  a.se = fma2(d0, mk2(0.f, 1.f), a.se);
  a.se = fma2(d1, mk2(2.f, 3.f), a.se);
}
template <int NT>
__device__ __forceinline__ void wacc_lane(const WAcc& a, int tid, float ic, float* s1, float* s2, float* sx) {
  *s1 = hsum(a.s1);
  *s2 = hsum(a.s2);
  *sx = fmaf((float)(4 * tid) - ic, *s1, fmaf((float)(4 * NT), hsum(a.sr), hsum(a.se)));
}

// ---------------------------------------------------------------------------------------------------------------------
// ---- cross-wave prefix sums of the wave-row totals.  A wave-row is the 256 samples a wave holds of one register row; its
// total part[4*w + r] comes out of the in-wave DPP scan.  ONE wave turns the R*NW totals (time order j = r*NW + w) into
// exclusive prefixes in double and leaves each wave what it needs as floats, laid out [w][r] (one 16-byte read per wave);
// the other waves wait at the barrier that follows instead of running the same 60-instruction double scan eight times.
template <int NW>
__device__ __forceinline__ double row_prefix_f64(const float* part, int lane, double* total) {
  const int jr = lane / NW, jw = lane - jr * NW;   // lane j <-> wave-row (r = jr, w = jw)
  const double pvv = (lane < R * NW) ? (double)part[4 * jw + jr] : 0.0;
  const double pinc = wave_incl_scan_sum_f64(pvv);
  if (total) *total = readlane_d(pinc, R * NW - 1);
  return pinc - pvv;
}
// InvCRFilter y = x + c*cumsum(x) (dsp_icpc.jl:119-120): c * (sum of x before the wave-row) as a float per wave-row
template <int NW>
__device__ __forceinline__ void pz_offsets_scan(const float* part, float* scn, double pz_c64, int wave, int lane) {
  if (wave != 0) return;
  const double ex = row_prefix_f64<NW>(part, lane, nullptr);
  const int jr = lane / NW, jw = lane - jr * NW;
  if (lane < R * NW) scn[4 * jw + jr] = (float)(pz_c64 * ex);
}
template <int NW>
__device__ __forceinline__ void pz_apply(f4 (&x)[R], const float (&inc)[R], const float (&tot)[R], const float* scn, float pz_c, int wave) {
  const f4 co = *reinterpret_cast<const f4*>(&scn[4 * wave]);
  const float cw[R] = {co.x, co.y, co.z, co.w};
  const f2 c2 = splat(pz_c);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float coff = fmaf(pz_c, inc[r] - tot[r], cw[r]);   // c * (sum of x before this lane's quad)
    const float r1 = x[r].x + x[r].y, r2 = r1 + x[r].z, r3 = r2 + x[r].w;   // running sums inside the quad
    const f2 cf = splat(coff);
    x[r].xy = fma2(c2, mk2(x[r].x, r1), x[r].xy + cf);
    x[r].zw = fma2(c2, mk2(r2, r3), x[r].zw + cf);
  }
}
// T = exclusive prefix sum of y: the sum before each wave-row split into hi = float(sum), lo = float(sum - hi); T[L] -> *t_end
template <int NW>
__device__ __forceinline__ void t_offsets_scan(const float* part, float* hilo, float* t_end, int wave, int lane) {
  if (wave != 0) return;
  double total;
  const double ex = row_prefix_f64<NW>(part, lane, &total);
  const int jr = lane / NW, jw = lane - jr * NW;
  const float hi = (float)ex;
  if (lane < R * NW) { hilo[4 * jw + jr] = hi; hilo[R * NW + 4 * jw + jr] = (float)(ex - (double)hi); }
  if (lane == 0) *t_end = (float)total;
}
// T of a quad = hi + (lo + (sum inside the wave-row before the sample)): one rounding at the magnitude of T
template <int NT, typename LdsF>
__device__ __forceinline__ void t_rows_store(const f4 (&y)[R], const float (&tin)[R], const float (&tot)[R], const float* hilo, LdsF* B, int tid, int wave) {
  constexpr int NW = NT / 64;
  const f4 h = *reinterpret_cast<const f4*>(&hilo[4 * wave]), l = *reinterpret_cast<const f4*>(&hilo[R * NW + 4 * wave]);
  const float hw[R] = {h.x, h.y, h.z, h.w}, lw[R] = {l.x, l.y, l.z, l.w};
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float p0 = tin[r] - tot[r];   // sum of the wave-row's samples before this lane's quad (exclusive DPP scan)
    const float p1 = p0 + y[r].x, p2 = p1 + y[r].y, p3 = p2 + y[r].z;
    const f2 h2 = splat(hw[r]), lo2 = splat(lw[r]);
    const f2 ta = h2 + (lo2 + mk2(p0, p1)), tb = h2 + (lo2 + mk2(p2, p3));
    *reinterpret_cast<f4*>(&B[4 * (tid + NT * r)]) = (f4){ta.x, ta.y, tb.x, tb.y};
  }
}


// SEP: CUSP and ZAC have their own geometry (two passes of the closed-form stage); a separate instantiation, so that the usual
// shared-geometry kernel keeps its register allocation
template <int NT, int M, bool SEP>
__global__ void __launch_bounds__(NT, 4)
icpc_lean_kernel(const float* __restrict__ wf, const IcpcDev* __restrict__ Pp, IcpcOutDev out, const float* __restrict__ ext_bl,
                 float ext_bl_scale) {
  using SM = Smem<NT>;
  constexpr int NW = SM::NW, Lp = SM::Lp, NWORDS = SM::NWORDS, L = Lp;
  static_assert(R * NW <= 64, "wave-row partials must fit one wave");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const IcpcDev& P = *Pp;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Lf_max = max(P.cusp.Lf, P.zac.Lf);   // (equal when the two filters share their geometry)
  SM S(smem_raw, cz_pad_floats(Lf_max));
  const float* w = wf + (size_t)blockIdx.x * (size_t)L;
  // first sample of this wave's 256-sample row r, and how a window [from, until] meets that row (wave-uniform)
  auto wrow = [&](int r) { return 4 * (64 * wave + NT * r); };
  // class of row r against window w (host-built table, one scalar load per window): outside / wholly inside / edge
  enum { WN_BL, WN_TAIL, WN_SGBL, WN_CUR0, WN_CURX };
  const uint32_t cls_bl = P.rowcls[WN_BL][wave], cls_tail = P.rowcls[WN_TAIL][wave], cls_sgbl = P.rowcls[WN_SGBL][wave],
                 cls_cur0 = P.rowcls[WN_CUR0][wave], cls_curx = P.rowcls[WN_CURX][wave];
  auto row_out = [&](uint32_t cls, int r) { return ((cls >> r) & 1u) != 0u; };
  auto row_in = [&](uint32_t cls, int r) { return ((cls >> (4 + r)) & 1u) != 0u; };

  // ------------------------------------------------------------------------------------------------ phase 0: load
  STAMP(0); DSTOP(0);
  f4 x[R];
  // uint16 ADC counts (ldsp_icpc_opts.in_u16) are converted as they are loaded: no separate cast pass over HBM
  const uint16_t* w16 = reinterpret_cast<const uint16_t*>(wf) + (size_t)blockIdx.x * (size_t)L;
  auto wv = [&](int i) { return P.in_u16 ? (float)w16[i] : w[i]; };
  if (P.in_u16) {   // (block-uniform)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint2 q = *reinterpret_cast<const uint2*>(w16 + 4 * (tid + NT * r));
      x[r] = (f4){(float)(q.x & 0xffffu), (float)(q.x >> 16), (float)(q.y & 0xffffu), (float)(q.y >> 16)};
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = *reinterpret_cast<const f4*>(w + 4 * (tid + NT * r));
  }
  const float pv_bl = wv(P.bl.from);      // pivot of the baseline sums: the window's first sample
  for (int i = tid; i < 2 * EST_TBL; i += NT)   // LSQ basis tables of the two estimators -> LDS
    S.estB[i] = (i < EST_TBL) ? P.sig_est.B[i] : P.int_est.B[i - EST_TBL];
  if (tid < (int)(sizeof(Slots) / 4)) {
    const int o = tid * 4;
    uint32_t init = 0;
    if (o >= (int)offsetof(Slots, imin) && o < (int)offsetof(Slots, imax)) init = 0x7fffffffu;
    else if (o >= (int)offsetof(Slots, imax)) init = 0xffffffffu;   // -1
    reinterpret_cast<uint32_t*>(S.sl)[tid] = init;
  }
  if (tid < 64) S.B[Lp + tid] = 0.f;

  // ---------------------------------------------------------------------------------- phase 1: raw extremes, baseline sums
  {
    float rmax = vmax3(x[0].x, x[0].y, x[0].z), rmin = vmin3(x[0].x, x[0].y, x[0].z);
    rmax = vmax3(rmax, x[0].w, x[1].x); rmin = vmin3(rmin, x[0].w, x[1].x);
    rmax = vmax3(rmax, x[1].y, x[1].z); rmin = vmin3(rmin, x[1].y, x[1].z);
    rmax = vmax3(rmax, x[1].w, x[2].x); rmin = vmin3(rmin, x[1].w, x[2].x);
    rmax = vmax3(rmax, x[2].y, x[2].z); rmin = vmin3(rmin, x[2].y, x[2].z);
    rmax = vmax3(rmax, x[2].w, x[3].x); rmin = vmin3(rmin, x[2].w, x[3].x);
    rmax = vmax3(rmax, x[3].y, x[3].z); rmin = vmin3(rmin, x[3].y, x[3].z);
    rmax = vmax(rmax, x[3].w); rmin = vmin(rmin, x[3].w);
    WAcc a = {splat(0.f), splat(0.f), splat(0.f), splat(0.f)};
    const f2 pv = splat(pv_bl);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (row_out(cls_bl, r)) continue;
      f2 d0 = x[r].xy - pv, d1 = x[r].zw - pv;
      if (!row_in(cls_bl, r)) {
        const int lo = P.bl.from - 4 * (tid + NT * r), hi = P.bl.until - 4 * (tid + NT * r);   // in-window e in [lo, hi]
        d0.x = (lo <= 0 && hi >= 0) ? d0.x : 0.f; d0.y = (lo <= 1 && hi >= 1) ? d0.y : 0.f;
        d1.x = (lo <= 2 && hi >= 2) ? d1.x : 0.f; d1.y = (lo <= 3 && hi >= 3) ? d1.y : 0.f;
      }
      wacc_quad(a, d0, d1, r);
    }
    float s1, s2, sx;
    wacc_lane<NT>(a, tid, (float)P.bl.ic, &s1, &s2, &sx);
    LDSP_DPP_GROUP5("v_add_f32_dpp", s1, "v_add_f32_dpp", s2, "v_add_f32_dpp", sx, "v_max_f32_dpp", rmax, "v_min_f32_dpp", rmin);
    if (lane == 63) {
      S.wred[0 * NW + wave] = s1; S.wred[1 * NW + wave] = s2; S.wred[2 * NW + wave] = sx;
      S.wred[3 * NW + wave] = rmax; S.wred[4 * NW + wave] = rmin;
    }
  }
  STAMP(1); DSTOP(1);
  __syncthreads();
  // every thread: baseline mean and the raw extremes from the per-wave partials
  float blmean, raw_max, raw_min;
  {
    const float s = fold_partials<NW>(S.wred, 0.f, [](float a, float b) { return a + b; });
    const float mx = fold_partials<NW>(S.wred + 3 * NW, -INFINITY, [](float a, float b) { return vmax(a, b); });
    const float mn = fold_partials<NW>(S.wred + 4 * NW, INFINITY, [](float a, float b) { return vmin(a, b); });
    blmean = fmaf(s, (float)P.bl.inv_n, pv_bl);
    if (ext_bl) blmean = ext_bl[blockIdx.x] * ext_bl_scale;   // windowed traces of dsp_icpc_compressed (dsp_icpc.jl:353)
    raw_max = mx; raw_min = mn;
    if (tid == 64 % NT) {   // a lane of wave 1 (wave 0 for one-wave tiles): sigma, slope, offset
      float s2 = 0.f, sx = 0.f;
      for (int ww = 0; ww < NW; ++ww) { s2 += S.wred[NW + ww]; sx += S.wred[2 * NW + ww]; }
      float m_, blsigma, blslope, bloffset;
      win_finish(s, s2, sx, P.bl, pv_bl, P.t_first, P.dt, &m_, &blsigma, &blslope, &bloffset);
      S.outv[C_blmean] = blmean; S.outv[C_blsigma] = blsigma; S.outv[C_blslope] = blslope; S.outv[C_bloffset] = bloffset;
      S.outv[C_e_max] = raw_max - blmean; S.outv[C_e_min] = raw_min - blmean;
    }
  }
  const float e_max = raw_max - blmean;
  STAMP(2); DSTOP(2);

  // saturation (src/saturation.jl:28-65): only a trace whose extremes reach a rail can have saturated samples
  {
    int n_low = 0, n_high = 0, cons_low = 0, cons_high = 0;
    if (raw_min <= P.sat_low || raw_max >= P.sat_high) {   // block-uniform, rare
#pragma unroll
      for (int r = 0; r < R; ++r) {
        n_low += (x[r].x == P.sat_low) + (x[r].y == P.sat_low) + (x[r].z == P.sat_low) + (x[r].w == P.sat_low);
        n_high += (x[r].x == P.sat_high) + (x[r].y == P.sat_high) + (x[r].z == P.sat_high) + (x[r].w == P.sat_high);
        *reinterpret_cast<f4*>(&S.A[4 * (tid + NT * r)]) = x[r];
      }
      n_low = wave_sum_all_i(n_low); n_high = wave_sum_all_i(n_high);
      if (lane == 0) { atomicAdd(&S.sl->imax[0], n_low); atomicAdd(&S.sl->imax[1], n_high); }   // the slots start at -1
      __syncthreads();
      for (int m = 0; m < SP; ++m) {
        const float v = S.A[tid + NT * m];
        ballot_store(v == P.sat_low, S.bm + M_FB * NWORDS, (NT >> 5) * m + 2 * wave);
        ballot_store(v == P.sat_high, S.bm + (M_FB + 1) * NWORDS, (NT >> 5) * m + 2 * wave);
      }
      __syncthreads();
      n_low = S.sl->imax[0] + 1; n_high = S.sl->imax[1] + 1;   // the slots started at -1
      if (tid < 2) {
        const uint32_t* b = S.bm + (M_FB + tid) * NWORDS;
        int best = 0, run = 0;
        for (int wd = 0; wd < NWORDS; ++wd) {
          const uint32_t v = b[wd];
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
      cons_low = __float_as_int(S.misc[0]); cons_high = __float_as_int(S.misc[1]);
      __syncthreads();
      if (tid < 2) S.sl->imax[tid] = -1;
    }
    if (tid == 0) {
      S.outv[C_n_sat_low] = __int_as_float(n_low); S.outv[C_n_sat_high] = __int_as_float(n_high);
      S.outv[C_n_sat_low_cons] = __int_as_float(cons_low); S.outv[C_n_sat_high_cons] = __int_as_float(cons_high);
    }
  }

  // ------------------------------------------------------- phase 2: shift, tailstats sums, cumulative sum for the pole-zero
  // shift_waveform(-blmean) (dsp_icpc.jl:105); tailstats on the shifted trace (src/tailstats.jl:22-72)
  const float pv_tl = __logf(fmaxf(wv(P.tail.from) - blmean, 1e-30f));   // pivot of the log sums
  float inc[R], tot[R];
  {
    const f2 bm2 = splat(blmean);
    WAcc a = {splat(0.f), splat(0.f), splat(0.f), splat(0.f)};
    float tmin = INFINITY;   // smallest in-window sample: tailstats returns zeros if any is <= 0 (:27-33)
    const f2 pv = splat(pv_tl);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      x[r].xy -= bm2; x[r].zw -= bm2;
      const f2 t = x[r].xy + x[r].zw;
      tot[r] = t.x + t.y;
      if (row_out(cls_tail, r)) continue;
      f2 v0 = x[r].xy, v1 = x[r].zw;
      const bool edge = !row_in(cls_tail, r);
      const int lo = P.tail.from - 4 * (tid + NT * r), hi = P.tail.until - 4 * (tid + NT * r);
      if (edge) {   // samples outside the window must not trip the sign test
        v0.x = (lo <= 0 && hi >= 0) ? v0.x : 1.f; v0.y = (lo <= 1 && hi >= 1) ? v0.y : 1.f;
        v1.x = (lo <= 2 && hi >= 2) ? v1.x : 1.f; v1.y = (lo <= 3 && hi >= 3) ? v1.y : 1.f;
      }
      tmin = vmin3(tmin, v0.x, v0.y); tmin = vmin3(tmin, v1.x, v1.y);
      f2 d0 = mk2(__logf(fmaxf(v0.x, 1e-30f)), __logf(fmaxf(v0.y, 1e-30f))) - pv;
      f2 d1 = mk2(__logf(fmaxf(v1.x, 1e-30f)), __logf(fmaxf(v1.y, 1e-30f))) - pv;
      if (edge) {
        d0.x = (lo <= 0 && hi >= 0) ? d0.x : 0.f; d0.y = (lo <= 1 && hi >= 1) ? d0.y : 0.f;
        d1.x = (lo <= 2 && hi >= 2) ? d1.x : 0.f; d1.y = (lo <= 3 && hi >= 3) ? d1.y : 0.f;
      }
      wacc_quad(a, d0, d1, r);
    }
    float s1, s2, sx;
    wacc_lane<NT>(a, tid, (float)P.tail.ic, &s1, &s2, &sx);
#pragma unroll
    for (int r = 0; r < R; ++r) inc[r] = tot[r];
    LDSP_DPP_GROUP8("v_add_f32_dpp", inc[0], "v_add_f32_dpp", inc[1], "v_add_f32_dpp", inc[2], "v_add_f32_dpp", inc[3], "v_add_f32_dpp", s1, "v_add_f32_dpp", s2, "v_add_f32_dpp", sx, "v_min_f32_dpp", tmin);
    if (lane == 63) {
      *reinterpret_cast<f4*>(&S.part[4 * wave]) = (f4){inc[0], inc[1], inc[2], inc[3]};   // [wave][r]
      S.wsum[(W_TAIL + 0) * NW + wave] = s1; S.wsum[(W_TAIL + 1) * NW + wave] = s2; S.wsum[(W_TAIL + 2) * NW + wave] = sx;
      if (tmin <= 0.f) S.sl->isum[IS_TAILBAD] = 1;   // any wave may set it (same value)
    }
  }
  STAMP(3); DSTOP(3);
  __syncthreads();
  // c * (sum of x before each wave-row), by wave 0 (pz_offsets_scan); the scratch is the ZAC stage's dpart, unused until then
  float* scn = reinterpret_cast<float*>(S.dpart);
  pz_offsets_scan<NW>(S.part, scn, P.pz_c64, wave, lane);
  if (tid == (128 % NT)) {   // a lane of wave 2: tailstats -> (mean, sigma, tau)
    float tail_mean = 0.f, tail_sigma = 0.f, tail_tau = 0.f;
    if (S.sl->isum[IS_TAILBAD] == 0) {
      float s1 = 0.f, s2 = 0.f, sx = 0.f, sl, of;
      for (int ww = 0; ww < NW; ++ww) { s1 += S.wsum[(W_TAIL + 0) * NW + ww]; s2 += S.wsum[(W_TAIL + 1) * NW + ww]; sx += S.wsum[(W_TAIL + 2) * NW + ww]; }
      win_finish(s1, s2, sx, P.tail, pv_tl, P.t_first, P.dt, &tail_mean, &tail_sigma, &sl, &of);
      tail_tau = -__builtin_amdgcn_rcpf(sl);
    }
    S.outv[C_tail_tau] = tail_tau; S.outv[C_tail_mean] = tail_mean; S.outv[C_tail_sigma] = tail_sigma;
  }
  // InvCRFilter: y = x + c*cumsum(x)  (dsp_icpc.jl:119-120); x becomes y
  __syncthreads();
  pz_apply<NW>(x, inc, tot, scn, P.pz_c, wave);
  auto& y = x;
  STAMP(4); DSTOP(4);

  // get_threshold at 10 / 50 / 80 / 90 / 99 % of the pre-PZ maximum (dsp_icpc.jl:132-136).  Intersect reports the FIRST up-crossing that
  // holds for tx_mintot samples; on a pulse that is the first sample at or above the threshold.  Found here from registers: the
  // maximum of every quad, one ballot per row and threshold (rows on which the lowest and the highest threshold agree need
  // two), first set lane by scalar bit scan -> the first quad of this wave reaching each threshold.  The candidate is
  // confirmed after the run scans; a trace that fails the confirmation runs the general bit-mask scan there.
  const float thr_tx[5] = {e_max * 0.1f, e_max * 0.5f, e_max * 0.8f, e_max * 0.9f, e_max * 0.99f};
  if (e_max > 0.f) {
    int qfirst[5] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    bool all_found = false;   // wave-uniform
#pragma unroll
    for (int r = 0; r < R; ++r) {   // ascending: the first hit of a threshold is its earliest row; later rows are skipped
      if (all_found) continue;
      const float qm = vmax(vmax3(y[r].x, y[r].y, y[r].z), y[r].w);
      const unsigned long long b0 = __ballot(qm >= thr_tx[0]);
      if (b0 == 0ull) continue;
      const unsigned long long b4 = __ballot(qm >= thr_tx[4]);
      unsigned long long bq[5] = {b0, b0, b0, b0, b4};
      if (b0 != b4) { bq[1] = __ballot(qm >= thr_tx[1]); bq[2] = __ballot(qm >= thr_tx[2]); bq[3] = __ballot(qm >= thr_tx[3]); }
#pragma unroll
      for (int q = 0; q < 5; ++q)
        if (bq[q] && qfirst[q] == 0x7fffffff) qfirst[q] = NT * r + 64 * wave + (int)__builtin_ctzll(bq[q]);
      all_found = qfirst[4] != 0x7fffffff;   // thresholds ascend: the highest found implies all found
    }
    if (lane < 5) {
      const int qq = lane == 0 ? qfirst[0] : lane == 1 ? qfirst[1] : lane == 2 ? qfirst[2] : lane == 3 ? qfirst[3] : qfirst[4];
      if (qq != 0x7fffffff) atomicMin(&S.sl->imin[IM_TX0 + lane], qq);   // quad index; the sample inside it is found at the confirmation
    }
  }

  // ------------------------------------------------------------------ phase 3: Savitzky-Golay derivatives, current maxima
  // g[k] = sum_i c[i] y[k+i] (valid mode, trailing time axis).  S4 evaluation from the register-resident quad plus a halo
  // of the next M-1 samples; packed: (g0, g1) and (g2, g3) each take one v_pk_fma per tap on (even, odd)-aligned pairs.
  const int ng = L - P.sg_npts[0] + 1;
#pragma unroll
  for (int r = 0; r < R; ++r) *reinterpret_cast<f4*>(&S.A[4 * (tid + NT * r)]) = y[r];
  __syncthreads();   // A = y visible to every wave
  STAMP(5); DSTOP(5);
  {
    constexpr int NP = (M + 3 + 1) / 2;      // register pairs covering the M + 3 samples a quad's four outputs need
    f2 c0[M], c1[M], c2[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      c0[i] = splat(P.sg_c[0][i]);   // zero beyond the filter's own taps (the host zero-fills the block)
      c1[i] = splat(P.sg_c[1][i]);
      c2[i] = splat(P.sg_c[2][i]);
    }
    float gmax = -INFINITY;
    f2 g_s1 = splat(0.f), g_s2 = splat(0.f);   // sums of the SG output over the baseline window (pivot 0: a derivative has no level)
    float bv[4]; int bi[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { bv[f] = -INFINITY; bi[f] = 0x7fffffff; }
    // running arg-max of filter f inside its current window, branch-free: k in [from, until] as ONE unsigned compare, samples
    // outside the window masked to -inf, then a compare and two selects (first occurrence wins: strict >)
    const int cw_from[4] = {P.cur_from[0], P.cur_from[1], P.cur_from[2], P.cur_from[3]};
    const uint32_t cw_len[4] = {(uint32_t)(P.cur_until[0] - P.cur_from[0]), (uint32_t)(P.cur_until[1] - P.cur_from[1]),
                                (uint32_t)(P.cur_until[2] - P.cur_from[2]), (uint32_t)(P.cur_until[3] - P.cur_from[3])};
    auto track = [&](int f, int k, float g) {
      const float gw = ((uint32_t)(k - cw_from[f]) <= cw_len[f]) ? g : -INFINITY;
      const bool gt = gw > bv[f];
      bv[f] = gt ? gw : bv[f];
      bi[f] = gt ? k : bi[f];
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i0 = 4 * (tid + NT * r);
      f2 E[NP + 1], O[NP];
      E[0] = y[r].xy; E[1] = y[r].zw;
#pragma unroll
      for (int j = 2; j < NP + 1; j += 2) {   // halo (runs into the gap / B past the trace: only masked outputs see that)
        const f4 h = *reinterpret_cast<const f4*>(&S.A[i0 + 2 * j]);
        E[j] = h.xy;
        if (j + 1 < NP + 1) E[j + 1] = h.zw;
      }
#pragma unroll
      for (int j = 0; j < NP; ++j) O[j] = __builtin_shufflevector(E[j], E[j + 1], 1, 2);   // (w[2j+1], w[2j+2])
      auto pair_at = [&](int i) { return (i & 1) ? O[i >> 1] : E[i >> 1]; };            // (w[i], w[i+1])
      f2 ga = splat(0.f), gb = splat(0.f);
#pragma unroll
      for (int i = 0; i < M; ++i) { ga = fma2(c0[i], pair_at(i), ga); gb = fma2(c0[i], pair_at(i + 2), gb); }
      if (wrow(r) + 255 >= ng) {   // only the wave-row holding the end of the output axis
        ga.x = (i0 + 0 < ng) ? ga.x : -INFINITY; ga.y = (i0 + 1 < ng) ? ga.y : -INFINITY;
        gb.x = (i0 + 2 < ng) ? gb.x : -INFINITY; gb.y = (i0 + 3 < ng) ? gb.y : -INFINITY;
      }
      gmax = vmax3(vmax3(gmax, ga.x, ga.y), gb.x, gb.y);
      *reinterpret_cast<f4*>(&S.B[i0]) = (f4){ga.x, ga.y, gb.x, gb.y};
      if (!row_out(cls_sgbl, r)) {   // sgbl.until <= ng-1: -inf never enters
        f2 d0 = ga, d1 = gb;
        if (!row_in(cls_sgbl, r)) {
          const int lo = P.sgbl.from - i0, hi = P.sgbl.until - i0;
          d0.x = (lo <= 0 && hi >= 0) ? d0.x : 0.f; d0.y = (lo <= 1 && hi >= 1) ? d0.y : 0.f;
          d1.x = (lo <= 2 && hi >= 2) ? d1.x : 0.f; d1.y = (lo <= 3 && hi >= 3) ? d1.y : 0.f;
        }
        g_s1 += d0 + d1;
        g_s2 = fma2(d0, d0, g_s2); g_s2 = fma2(d1, d1, g_s2);
      }
      const float go[4] = {ga.x, ga.y, gb.x, gb.y};
      if (!row_out(cls_cur0, r)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          track(0, i0 + e, go[e]);
        }
      }
      if (!row_out(cls_curx, r)) {   // SG(60 ns), SG(100 ns), plain derivative: only rows that touch the current window
        f2 h1a = splat(0.f), h1b = splat(0.f), h2a = splat(0.f), h2b = splat(0.f);
#pragma unroll
        for (int i = 0; i < M; ++i) { h1a = fma2(c1[i], pair_at(i), h1a); h1b = fma2(c1[i], pair_at(i + 2), h1b); }
        if (!P.sg_same_02) {
#pragma unroll
          for (int i = 0; i < M; ++i) { h2a = fma2(c2[i], pair_at(i), h2a); h2b = fma2(c2[i], pair_at(i + 2), h2b); }
        }
        const float ypv = (i0 > 0) ? S.A[i0 - 1] : 0.f;
        float g3[4] = {y[r].x - ypv, y[r].y - y[r].x, y[r].z - y[r].y, y[r].w - y[r].z};   // y[k] - y[k-1]
        if (i0 == 0) g3[0] = y[r].y - y[r].x;                                                // y[max(i,1)] - y[max(i-1,0)] at i = 0
        const float g1[4] = {h1a.x, h1a.y, h1b.x, h1b.y}, g2[4] = {h2a.x, h2a.y, h2b.x, h2b.y};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          track(1, i0 + e, g1[e]);
          if (!P.sg_same_02) track(2, i0 + e, g2[e]);
          track(3, i0 + e, g3[e]);
        }
      }
    }
    STAMP(6); DSTOP(6);
    float s1 = hsum(g_s1), s2 = hsum(g_s2);
    LDSP_DPP_GROUP3("v_add_f32_dpp", s1, "v_add_f32_dpp", s2, "v_max_f32_dpp", gmax);
    if (lane == 63) {
      S.wsum[(W_SGB + 0) * NW + wave] = s1; S.wsum[(W_SGB + 1) * NW + wave] = s2;
      atomicMax(&S.sl->fmx[FX_G], ford(gmax));
    }
    // arg-maxima of the current windows: only waves whose samples touch a window hold a candidate (wave-uniform test)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (f == 2 && P.sg_same_02) continue;
      if (__ballot(bi[f] != 0x7fffffff) != 0ull) {
        wave_argmax(bv[f], bi[f]);
        if (lane == 63) atomicMax(&S.sl->vi[VI_CUR0 + f], pack_vi(bv[f], bi[f]));
      }
    }
  }
  __syncthreads();
  STAMP(7); DSTOP(7);
  // filter f at output index k from y in LDS (the few samples the parabolas and crossing interpolations need)
  auto flt_at = [&](int f, int k) -> float {
    if (f < 3) {   // all M reads in flight together (taps beyond the filter's own read the margin and count as zero), summed in tap order
      float a[M];
#pragma unroll
      for (int i = 0; i < M; ++i) a[i] = S.A[k + i];
      asm volatile("" ::: "memory");
      float g = 0.f;
#pragma unroll
      for (int i = 0; i < M; ++i) g = fmaf(P.sg_c[f][i], (i < P.sg_npts[f]) ? a[i] : 0.f, g);
      return g;
    }
    return S.A[max(k, 1)] - S.A[max(k - 1, 0)];
  };
  // in-trace pile-up threshold (dsp_routines.jl:75-77) and t50_current threshold (dsp_icpc.jl:192)
  float thr_intr, thr_sg50;
  {
    const float s1 = fold_partials<NW>(S.wsum + (W_SGB + 0) * NW, 0.f, [](float a, float b) { return a + b; });
    const float s2 = fold_partials<NW>(S.wsum + (W_SGB + 1) * NW, 0.f, [](float a, float b) { return a + b; });
    const float m_ = s1 * (float)P.sgbl.inv_n;
    const float var_ = fmaxf(fmaf(s2, (float)P.sgbl.inv_n, -m_ * m_), 0.f);
    thr_intr = __builtin_amdgcn_sqrtf(var_) * P.intrace_nsigma;
    if (thr_intr == 0.f) thr_intr = 1.f;
    thr_sg50 = ford_inv(S.sl->fmx[FX_G]) * 0.5f;
  }
  // LS pass over the SG output, two rows per step (ds_read2st64): two masks by ballot — the pile-up threshold (a count of runs is
  // needed) and half the maximum (t50_current, dsp_icpc.jl:192-195: the first run of tx_mintot samples, found by the run scans
  // below like t0; reading g[k-1] and g[k+1] here for a direct crossing test cost two more LDS reads per sample).
  {
    uint32_t acci = 0u;   // lanes 0..31: words of the pile-up mask, 32..63: of the half-maximum mask (g = -inf beyond the output: 0)
    const float* gb = &S.B[tid];
#pragma unroll
    for (int m = SP - 2; m >= 0; m -= 2) {
      const f2 g = mk2(gb[NT * m], gb[NT * (m + 1)]);
      const unsigned long long ba = __ballot(g.x >= thr_intr), bb = __ballot(g.y >= thr_intr);
      const unsigned long long bc = __ballot(g.x >= thr_sg50), bd = __ballot(g.y >= thr_sg50);
      put_ballots(acci, ba, bb, 2 * m);
      put_ballots(acci, bc, bd, 32 + 2 * m);
    }
    // lane j: mask j >> 5, row (j & 31) >> 1, half j & 1 of the wave's 64-bit word
    S.bm[((lane >> 5) ? M_SG50 : M_INTR) * NWORDS + (NT >> 5) * ((lane & 31) >> 1) + 2 * wave + (lane & 1)] = acci;
  }
  STAMP(8); DSTOP(8);

  // ------------------------------------------------------------------------------------ phase 4: T = prefix sum of y
  float pv_pz;
  {
    float tin[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const f2 t = y[r].xy + y[r].zw; tot[r] = t.x + t.y; tin[r] = tot[r]; }
    // signalstats of the pole-zero corrected tail (dsp_icpc.jl:122), pivot = its first sample
    WAcc a = {splat(0.f), splat(0.f), splat(0.f), splat(0.f)};
    pv_pz = S.A[P.tail.from];
    const f2 pv = splat(pv_pz);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (row_out(cls_tail, r)) continue;
      f2 d0 = y[r].xy - pv, d1 = y[r].zw - pv;
      if (!row_in(cls_tail, r)) {
        const int lo = P.tail.from - 4 * (tid + NT * r), hi = P.tail.until - 4 * (tid + NT * r);
        d0.x = (lo <= 0 && hi >= 0) ? d0.x : 0.f; d0.y = (lo <= 1 && hi >= 1) ? d0.y : 0.f;
        d1.x = (lo <= 2 && hi >= 2) ? d1.x : 0.f; d1.y = (lo <= 3 && hi >= 3) ? d1.y : 0.f;
      }
      wacc_quad(a, d0, d1, r);
    }
    float s1, s2, sx;
    wacc_lane<NT>(a, tid, (float)P.tail.ic, &s1, &s2, &sx);
    LDSP_DPP_GROUP7("v_add_f32_dpp", tin[0], "v_add_f32_dpp", tin[1], "v_add_f32_dpp", tin[2], "v_add_f32_dpp", tin[3], "v_add_f32_dpp", s1, "v_add_f32_dpp", s2, "v_add_f32_dpp", sx);
    float* pb = S.part + R * NW;   // second buffer
    if (lane == 63) {
      *reinterpret_cast<f4*>(&pb[4 * wave]) = (f4){tin[0], tin[1], tin[2], tin[3]};
      S.wsum[(W_PZ + 0) * NW + wave] = s1; S.wsum[(W_PZ + 1) * NW + wave] = s2; S.wsum[(W_PZ + 2) * NW + wave] = sx;
    }
    __syncthreads();   // also: every LS read of the SG output in B is done
    t_offsets_scan<NW>(pb, scn + R * NW, &S.B[Lp], wave, lane);   // wave 0; T[L] -> B[Lp]
    __syncthreads();
    t_rows_store<NT>(y, tin, tot, scn + R * NW, S.B, tid, wave);
  }
  __syncthreads();
  STAMP(9); DSTOP(9);
  if (tid == (256 % NT)) {   // a lane of wave 4
    float s1 = 0.f, s2 = 0.f, sx = 0.f;
    for (int ww = 0; ww < NW; ++ww) { s1 += S.wsum[(W_PZ + 0) * NW + ww]; s2 += S.wsum[(W_PZ + 1) * NW + ww]; sx += S.wsum[(W_PZ + 2) * NW + ww]; }
    float tailmean, tailsigma, tailslope, tailoffset;
    win_finish(s1, s2, sx, P.tail, pv_pz, P.t_first, P.dt, &tailmean, &tailsigma, &tailslope, &tailoffset);
    S.outv[C_tailmean] = tailmean; S.outv[C_tailsigma] = tailsigma; S.outv[C_tailslope] = tailslope; S.outv[C_tailoffset] = tailoffset;
  }

  // ------------------------------------------------------------------------------------ phase 5: lane-strided sweeps over T
  // A trapezoid is evaluated unscaled, o' = (T[k+flen]-T[k+n1+g])*(inv2/inv1) - (T[k+n1]-T[k]); two rows per step: a pair's
  // reads are one ds_read2st64_b32 each and its arithmetic is packed.  Rows outside an output range read zeros / stale
  // samples and are masked by -inf / +inf.
  {
    const float* tb = &S.B[tid];
    auto rd2 = [&](const float* p, int m) { return mk2(p[NT * m], p[NT * (m + 1)]); };
    // ---- sweep A: the two threshold masks of the t0 trapezoid (get_t0, dsp_routines.jl:9-25; inverted: dsp_icpc.jl:207)
    {
      const TrapDev t0 = P.t0;
      const float *ta = tb + t0.n1, *tbb = tb + t0.n1 + t0.g, *tc = tb + t0.flen;
      const f2 rr = splat(t0.rr);
      const float thr0 = P.t0_thr * t0.navg;
      const int nout = L - t0.flen + 1;
      uint32_t acc = 0u;   // lanes 0..31: words of the t0 mask, 32..63: of the inverted one (rows beyond the output range stay 0)
      // A first leg of <= 3 samples (get_t0's 40 ns) is summed from y itself: on the tail T is 1e7..1e8 and a difference of two
      // of its float values is good to 1..8 counts — the size of the threshold the INVERTED trace is tested against there.
      const int n1s = t0.n1;
      const bool short1 = n1s <= 3;
      const float* ya = &S.A[tid];
#pragma unroll
      for (int m = 0; m < SP; m += 2) {
        if (NT * m >= nout) continue;   // pair beyond the output range
        const f2 b = rd2(tbb, m), c = rd2(tc, m);
        f2 ml;   // minus the first leg's sum
        if (short1) {
          f2 l = rd2(ya, m);
          if (n1s >= 2) l = l + rd2(ya + 1, m);
          if (n1s >= 3) l = l + rd2(ya + 2, m);
          ml = splat(0.f) - l;
        } else {
          ml = rd2(tb, m) - rd2(ta, m);
        }
        f2 o = fma2(c - b, rr, ml);
        if (NT * (m + 2) > nout) {   // the pair that holds the end of the output range (and pairs beyond it)
          o.x = (tid + NT * m < nout) ? o.x : NAN; o.y = (tid + NT * (m + 1) < nout) ? o.y : NAN;   // NaN: both comparisons false
        }
        const unsigned long long p0 = __ballot(o.x >= thr0), p1 = __ballot(o.y >= thr0);
        const unsigned long long n0 = __ballot(o.x <= -thr0), n1 = __ballot(o.y <= -thr0);   // -trap >= thr
        put_ballots(acc, p0, p1, 2 * m);
        put_ballots(acc, n0, n1, 32 + 2 * m);
      }
      static_assert(M_T0INV == M_T0 + 1, "mask order");
      // lane j: mask j >> 5, row (j & 31) >> 1, half j & 1 of the wave's 64-bit word
      S.bm[(M_T0 + (lane >> 5)) * NWORDS + (NT >> 5) * ((lane & 31) >> 1) + 2 * wave + (lane & 1)] = acc;
    }
    STAMP(10); DSTOP(10);
    // ---- sweep B: extrema of the three fixed trapezoids, arg-max of the optimised one (dsp_icpc.jl:147-164, 202-204)
    float mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY, mn0 = INFINITY, mn2 = INFINITY;
    float bo_v = -INFINITY; int bo_i = 0x7fffffff;
    {
      const TrapDev f0 = P.fixed[0], f1 = P.fixed[1], f2_ = P.fixed[2], fo = P.opt;
      const float *f0a = tb + f0.n1, *f0b = tb + f0.n1 + f0.g, *f0c = tb + f0.flen;
      const float *f1a = tb + f1.n1, *f1b = tb + f1.n1 + f1.g, *f1c = tb + f1.flen;
      const float *f2a = tb + f2_.n1, *f2b = tb + f2_.n1 + f2_.g, *f2c = tb + f2_.flen;
      const float *foa = tb + fo.n1, *fob = tb + fo.n1 + fo.g, *foc = tb + fo.flen;
      const f2 rr0 = splat(f0.rr), rr1 = splat(f1.rr), rr2 = splat(f2_.rr), rro = splat(fo.rr);
      const int n0 = L - f0.flen + 1, n1 = L - f1.flen + 1, n2 = L - f2_.flen + 1, no = L - fo.flen + 1;
      const int nall = min(min(n0, n1), min(n2, no));
#pragma unroll
      for (int m = 0; m < SP; m += 2) {
        if (NT * m >= max(max(n0, n1), max(n2, no))) continue;   // pair beyond every output range
        const f2 Tk = rd2(tb, m);
        const f2 a0 = rd2(f0a, m), b0 = rd2(f0b, m), c0_ = rd2(f0c, m);
        const f2 a1 = rd2(f1a, m), b1 = rd2(f1b, m), c1_ = rd2(f1c, m);
        const f2 a2 = rd2(f2a, m), b2 = rd2(f2b, m), c2_ = rd2(f2c, m);
        const f2 ao = rd2(foa, m), bo = rd2(fob, m), co = rd2(foc, m);
        f2 o0 = fma2(c0_ - b0, rr0, Tk - a0), o1 = fma2(c1_ - b1, rr1, Tk - a1), o2 = fma2(c2_ - b2, rr2, Tk - a2), oo = fma2(co - bo, rro, Tk - ao);
        if (NT * (m + 2) <= nall) {   // pair wholly inside every output range (most pairs)
          mx0 = vmax3(mx0, o0.x, o0.y); mn0 = vmin3(mn0, o0.x, o0.y);
          mx1 = vmax3(mx1, o1.x, o1.y);
          mx2 = vmax3(mx2, o2.x, o2.y); mn2 = vmin3(mn2, o2.x, o2.y);
        } else {
          const int k0 = tid + NT * m, k1 = k0 + NT;
          mx0 = vmax3(mx0, k0 < n0 ? o0.x : -INFINITY, k1 < n0 ? o0.y : -INFINITY);
          mn0 = vmin3(mn0, k0 < n0 ? o0.x : INFINITY, k1 < n0 ? o0.y : INFINITY);
          mx1 = vmax3(mx1, k0 < n1 ? o1.x : -INFINITY, k1 < n1 ? o1.y : -INFINITY);
          mx2 = vmax3(mx2, k0 < n2 ? o2.x : -INFINITY, k1 < n2 ? o2.y : -INFINITY);
          mn2 = vmin3(mn2, k0 < n2 ? o2.x : INFINITY, k1 < n2 ? o2.y : INFINITY);
          oo.x = k0 < no ? oo.x : -INFINITY; oo.y = k1 < no ? oo.y : -INFINITY;
        }
        if (oo.x > bo_v) { bo_v = oo.x; bo_i = tid + NT * m; }
        if (oo.y > bo_v) { bo_v = oo.y; bo_i = tid + NT * (m + 1); }
      }
      mx0 *= f0.inv1; mn0 *= f0.inv1; mx1 *= f1.inv1; mx2 *= f2_.inv1; mn2 *= f2_.inv1; bo_v *= fo.inv1;
    }
    STAMP(11); DSTOP(11);
    LDSP_DPP_GROUP5("v_max_f32_dpp", mx0, "v_max_f32_dpp", mx1, "v_max_f32_dpp", mx2, "v_min_f32_dpp", mn0, "v_min_f32_dpp", mn2);
    wave_argmax(bo_v, bo_i);
    const unsigned long long bo = pack_vi(bo_v, bo_i);
    if (lane == 63) {
      atomicMax(&S.sl->fmx[FX_F0], ford(mx0));
      atomicMax(&S.sl->fmx[FX_F1], ford(mx1));
      atomicMax(&S.sl->fmx[FX_F2], ford(mx2));
      atomicMax(&S.sl->fmx[FX_F0I], ford(-mn0));   // max(trap(-y)) = -min(trap(y))
      atomicMax(&S.sl->fmx[FX_F2I], ford(-mn2));
      atomicMax(&S.sl->vi[VI_OPT], bo);
    }
  }
  __syncthreads();
  STAMP(12); DSTOP(12);
  // (Placed ahead of the mask scans: candidates and y are final since phase 3, and its chain of three dependent LDS reads overlaps
  // the scans' instead of standing between their barrier and the crossings.)
  // Confirmation of the five threshold candidates: imin[q] = first QUAD with a sample at or above threshold q.  Lane q of every
  // wave finds the sample inside the quad and checks that it is not sample 0 (a run that starts the trace is no crossing) and
  // that the next tx_mintot - 1 samples stay at or above the threshold; all waves reach the same verdict.  A trace that fails
  // (a noise spike in front of the pulse, thresholds that do not ascend) runs the general scan: bit-masks of y by ballot, run scan
  // on the words (src/intersect_maximum.jl:41-56).
  int p_tx = 0x7fffffff;   // lane q < 5: first confirmed sample of threshold q
  {
    const int q = min(lane, 4);
    const float thrq = e_max * ((q == 0) ? 0.1f : (q == 1) ? 0.5f : (q == 2) ? 0.8f : (q == 3) ? 0.9f : 0.99f);   // = thr_tx[q]
    const int qd = S.sl->imin[IM_TX0 + q];
    bool ok = e_max > 0.f;
    if (ok && qd != 0x7fffffff) {
      const f4 v = *reinterpret_cast<const f4*>(&S.A[4 * qd]);
      const int e = (v.x >= thrq) ? 0 : (v.y >= thrq) ? 1 : (v.z >= thrq) ? 2 : 3;
      p_tx = 4 * qd + e;
      ok = p_tx >= 1 && p_tx + P.tx_mintot <= L;
      for (int j = 1; ok && j < P.tx_mintot; ++j) ok = S.A[p_tx + j] >= thrq;
    }
    if (__ballot(lane < 5 && !ok) != 0ull) {   // block-uniform
      __syncthreads();
      if (tid < 5) S.sl->imin[IM_TX0 + tid] = 0x7fffffff;
      for (int m = 0; m < SP; ++m) {
        const float yv = S.A[tid + NT * m];
#pragma unroll
        for (int qq = 0; qq < 5; ++qq) {
          const unsigned long long b = __ballot(yv >= thr_tx[qq]);
          if (lane == 0) *reinterpret_cast<unsigned long long*>(&S.bm[(M_FB + qq) * NWORDS + (NT >> 5) * m + 2 * wave]) = b;
        }
      }
      __syncthreads();
      for (int j = tid; j < 5 * NWORDS; j += NT) {
        const int qq = j / NWORDS, wd = j % NWORDS;
        int c, f;
        intersect_word(S.bm + (M_FB + qq) * NWORDS, wd, NWORDS, P.tx_mintot, &c, &f);
        if (c) atomicMin(&S.sl->imin[IM_TX0 + qq], f);
      }
      __syncthreads();
      p_tx = S.sl->imin[IM_TX0 + q];
    }
  }
  // Intersect scans on the bit-masks (thread <-> word): t0, inverted t0, in-trace pile-up (lower half of the threads), half maximum
  // of the SG output (upper half).  Every word a thread's scans can need is
  // read first (one wait instead of a dependent LDS round trip per word), and the run tests are loop-free (intersect_pre /
  // intersect_rev_pre); run lengths beyond their windows take the word-by-word forms.
  static_assert(2 * NWORDS == NT, "one t0 / inverted-t0 word per thread");
  if (P.t0_mintot <= 97 && P.intrace_mintot <= 32) {   // block-uniform
    const int q = tid / NWORDS, wd = tid % NWORDS;
    const uint32_t* b0 = S.bm + (M_T0 + q) * NWORDS;
    const bool has_i = tid < NWORDS;   // the lower half also takes an in-trace word, the upper half a word of the half-maximum mask
    const uint32_t* bi = S.bm + (has_i ? M_INTR : M_SG50) * NWORDS;
    uint32_t t[5], u[3];
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] = b0[min(max(wd + k - 1, 0), NWORDS - 1)];
#pragma unroll
    for (int k = 0; k < 3; ++k) u[k] = bi[min(max(wd + k - 1, 0), NWORDS - 1)];
    asm volatile("" ::: "memory");
    if (wd == 0) { t[0] = 0u; u[0] = 0u; }
#pragma unroll
    for (int k = 2; k < 5; ++k) t[k] = (wd + k - 1 < NWORDS) ? t[k] : 0u;
    u[2] = (wd + 1 < NWORDS) ? u[2] : 0u;
    int c, f;
    intersect_pre(t[0], t[1], t[2], t[3], t[4], wd, P.t0_mintot, &c, &f);
    if (c) { atomicAdd(&S.sl->isum[IS_T0 + q], c); atomicMin(&S.sl->imin[IM_T0 + q], f); }
    if (has_i) {
      intersect_rev_pre(u[0], u[1], u[2], wd, ng, P.intrace_mintot, &c, &f);
      if (c) { atomicAdd(&S.sl->isum[IS_INTR], c); atomicMax(&S.sl->imax[0], f); }
    } else {   // (tx_mintot <= 2 in this kernel)
      intersect_pre(u[0], u[1], u[2], 0u, 0u, wd, P.tx_mintot, &c, &f);
      if (c) atomicMin(&S.sl->imin[IM_SG50], f);
    }
  } else {
    for (int wd = tid; wd < NWORDS; wd += NT) {
      int c, f;
      intersect_word(S.bm + M_SG50 * NWORDS, wd, NWORDS, P.tx_mintot, &c, &f);
      if (c) atomicMin(&S.sl->imin[IM_SG50], f);
    }
    for (int j = tid; j < 2 * NWORDS; j += NT) {
      const int q = j / NWORDS, wd = j % NWORDS;
      int c, f;
      intersect_word(S.bm + (M_T0 + q) * NWORDS, wd, NWORDS, P.t0_mintot, &c, &f);
      if (c) { atomicAdd(&S.sl->isum[IS_T0 + q], c); atomicMin(&S.sl->imin[IM_T0 + q], f); }
    }
    for (int wd = tid; wd < NWORDS; wd += NT) {
      int c, f;
      intersect_word_rev(S.bm + M_INTR * NWORDS, wd, NWORDS, ng, P.intrace_mintot, &c, &f);
      if (c) { atomicAdd(&S.sl->isum[IS_INTR], c); atomicMax(&S.sl->imax[0], f); }
    }
  }
  __syncthreads();
  STAMP(13); DSTOP(13);
  // t50_current and the in-trace pile-up position: one wave (5, or the last), lanes 0..3 evaluate the four SG samples
  if (wave == min(5, NW - 1)) {
    const int intr_n = S.sl->isum[IS_INTR];
    const int p = S.sl->imin[IM_SG50], e = S.sl->imax[0];
    const bool has50 = p != 0x7fffffff;
    const int at = (lane == 0) ? p - 1 : (lane == 1) ? p : (lane == 2) ? e + 1 : e;
    float ev = 0.f;
    if (lane < 4 && ((lane < 2) ? has50 : intr_n > 0)) ev = flt_at(0, at);
    const float yl5 = __shfl(ev, 0), yh5 = __shfl(ev, 1), yli = __shfl(ev, 2), yhi = __shfl(ev, 3);
    if (lane == 0) {
      const float tg_first = P.t_first + P.dt * (float)(P.sg_npts[0] - 1);   // trailing alignment (A1)
      float t50cur_us = 0.f, intr_x = NAN;
      if (has50) t50cur_us = (tg_first + P.dt * ((float)(p - 1) + (thr_sg50 - yl5) / (yh5 - yl5))) * P.inv_unit_per_us;
      if (intr_n > 0) {   // reversed index pos' = ng-1-e; r[pos'-1] = g[e+1], r[pos'] = g[e]
        const int pr = ng - 1 - e;
        const float xl = tg_first + P.dt * (float)(pr - 1);
        const float xr_ = (thr_intr - yli) * P.dt / (yhi - yli) + xl;
        intr_x = (tg_first + P.dt * (float)(ng - 1)) - xr_;   // last(time) - x   (dsp_routines.jl:81)
      }
      S.outv[C_t50_current] = t50cur_us; S.outv[C_inTrace_intersect] = intr_x; S.outv[C_inTrace_n] = __int_as_float(intr_n);
    }
  }
  if (tid == (384 % NT)) {   // a lane of wave 6
    float v; int i;
    unpack_vi(S.sl->vi[VI_OPT], &v, &i);
    S.outv[C_e_trap_max] = v; S.outv[C_t_trap_max] = P.t_first + P.dt * (float)(i + P.opt.flen - 1);
    S.outv[C_e_10410] = ford_inv(S.sl->fmx[FX_F0]); S.outv[C_e_535] = ford_inv(S.sl->fmx[FX_F1]); S.outv[C_e_313] = ford_inv(S.sl->fmx[FX_F2]);
    S.outv[C_e_10410_inv] = ford_inv(S.sl->fmx[FX_F0I]); S.outv[C_e_313_inv] = ford_inv(S.sl->fmx[FX_F2I]);   // trap(-y) = -trap(y)  (dsp_icpc.jl:199-204)
  }
  // crossing positions (sample units, split int + frac); NaN -> 0 us (dsp_routines.jl:24,41).  Seven interpolations, one per
  // lane (q < 5: threshold q of y; 5: t0; 6: inverted t0), evaluated by every wave and handed out by readlane.
  Pos ptx1 = {0, 0.f}, ptx2 = {0, 0.f}, pt0 = {0, 0.f};
  constexpr int W_CZWIN = 4 % NW;   // the wave that places the CUSP / ZAC estimator windows
  if (wave <= 2 || wave == W_CZWIN || wave == NW - 1) {   // the waves that use a position (estimators, windows) or store the times
    const int q = min(lane, 6);
    const int p = (q < 5) ? p_tx : S.sl->imin[q];
    const bool has = (q < 5) ? p != 0x7fffffff : S.sl->isum[IS_T0 + q - 5] > 0;
    const float frac = (q == 0) ? 0.1f : (q == 1) ? 0.5f : (q == 2) ? 0.8f : (q == 3) ? 0.9f : 0.99f;
    const float thr = (q < 5) ? e_max * frac : P.t0_thr;
    Pos pp; pp.ip = 0; pp.fp = -P.t_first / P.dt;   // sample position of t = 0
    float us = 0.f;
    if (has) {
      float yl, yh; int base;
      if (q < 5) {
        yl = S.A[p - 1]; yh = S.A[p]; base = p - 1;
      } else {
        yl = trap_at_y(S.B, S.A, p - 1, P.t0); yh = trap_at_y(S.B, S.A, p, P.t0);
        if (q == 6) { yl = -yl; yh = -yh; }
        base = p - 1 + (P.t0.flen - 1);   // trailing alignment (A1): back to input index space
      }
      pp.ip = base; pp.fp = (thr - yl) / (yh - yl);
      us = (P.t_first + P.dt * ((float)base + pp.fp)) * P.inv_unit_per_us;
    } else {
      pp = pos_norm(pp);
    }
    ptx1.ip = __builtin_amdgcn_readlane(pp.ip, 1); ptx1.fp = readlane_f(pp.fp, 1);
    ptx2.ip = __builtin_amdgcn_readlane(pp.ip, 2); ptx2.fp = readlane_f(pp.fp, 2);
    pt0.ip = __builtin_amdgcn_readlane(pp.ip, 5); pt0.fp = readlane_f(pp.fp, 5);
    if (wave == NW - 1) {
      if (lane < 7) S.outv[lane == 0 ? C_t10 : lane == 1 ? C_t50 : lane == 2 ? C_t80 : lane == 3 ? C_t90 : lane == 4 ? C_t99 : lane == 5 ? C_t0 : C_t0_inv] = us;
      const float t90 = readlane_f(us, 3), t0u = readlane_f(us, 5);
      if (lane == 0) S.outv[C_drift_time] = (t90 - t0u) * P.unit_per_us;
    }
  }
  STAMP(14); DSTOP(14);

  // ------------------------------------------------------------------------------------ phase 6: signal estimators
  // e_trap = SignalEstimator(trap_opt output, t50 + rt + ft/2)  (dsp_icpc.jl:163): one wave, lane l = window point l.
  // qdrift / lq = get_qdrift (dsp_routines.jl:51-64): second difference of three estimates of the integrator output
  // I[i] = sum_{j<=i} y[j].  The weights of an estimate sum to one, so the common level I(ref) cancels exactly in
  // (E3 - E2) - (E2 - E1): each window point is taken RELATIVE to the first point of the first window, as a sum of y over
  // at most a few hundred samples (float, exact to ~1e-3) instead of a difference of two rounded prefix sums of 1e8.
  {
    lds_float* eslot = S.misc + 4;
    if (wave == 0) {
      Pos p = pos_add(ptx1, P.trap_pickoff);
      p.ip -= (P.opt.flen - 1);
      const int nsig = L - P.opt.flen + 1;
      float v = NAN;
      if (nsig >= P.sig_est.npts) {
        int i0; float u;
        est_window(P.sig_est, p, nsig, &i0, &u);
        float t = 0.f;
        if (lane < P.sig_est.npts) t = est_weight(P.sig_est, S.estB, lane, u) * trap_at(S.B, i0 + lane, P.opt);
        v = wave_total(t);
      }
      if (lane == 0) eslot[0] = v;
    }
    if (wave == W_CZWIN) {
      // windows of the CUSP / ZAC estimates (t50 + flt_length/2, dsp_icpc.jl:170,177) for the last phase
      const int nout_c = L - P.cusp.Lf + 1, nout_z = L - P.zac.Lf + 1;
      if (lane == 0) { S.misc[12] = 0.f; S.misc[13] = 0.f; }
      if (nout_c >= P.sig_est.npts) {
        Pos pc_ = pos_add(ptx1, P.cusp_pickoff);
        pc_.ip -= (P.cusp.Lf - 1);
        int i0c; float uc;
        est_window(P.sig_est, pc_, nout_c, &i0c, &uc);
        // the level at the left edge of the pick-off window: the CUSP / ZAC stage runs on y - cpiv (see there)
        if (lane == 0) { S.misc[8] = __int_as_float(i0c); S.misc[9] = uc; S.misc[12] = S.A[i0c]; }
      }
      if (nout_z >= P.sig_est.npts) {
        Pos pz2 = pos_add(ptx1, P.zac_pickoff);
        pz2.ip -= (P.zac.Lf - 1);
        int i0z; float uz;
        est_window(P.sig_est, pz2, nout_z, &i0z, &uz);
        if (lane == 0) { S.misc[10] = __int_as_float(i0z); S.misc[11] = uz; S.misc[13] = S.A[i0z]; }
      }
    }
    if (wave == 1 % NW || wave == 2 % NW) {
      const bool lq = (NW > 2) ? wave == 2 : false;
      for (int pass = 0; pass < ((NW > 2) ? 1 : 2); ++pass) {
        const bool is_lq = (NW > 2) ? lq : pass == 1;
        const Pos base = is_lq ? ptx2 : pt0;
        const float d1 = is_lq ? P.lq_d1 : P.qdrift_d1, d2 = is_lq ? P.lq_d2 : P.qdrift_d2;
        const Pos p1 = pos_add(base, d1), p2 = pos_add(base, d2);
        const int ips[3] = {base.ip, p1.ip, p2.ip};
        const float fps[3] = {base.fp, p1.fp, p2.fp};
        // scratch: the rows of the general threshold scan (M_FB..) — the crossings are done, the CUSP/ZAC stage clears the gap later
        float* scr = (5 * NWORDS >= 1024) ? reinterpret_cast<float*>(S.bm + M_FB * NWORDS) + (is_lq ? 512 : 0) : nullptr;
        const float res = qdrift_wave(P.int_est, S.estB + EST_TBL, S.A, L, ips, fps, scr);
        if (lane == 0) eslot[is_lq ? 2 : 1] = res;
      }
    }
    // get_wvf_maximum of the four current signals (src/interpolation.jl:30-46): parabola through the three samples about the
    // maximum if it is strictly interior.  One wave (3, or the last); lane 3f+d+1 evaluates filter f at i_f+d.  The arg-maxima
    // come from the SG phase; the job runs here, next to the estimators of waves 0-2, because ahead of the LS pass the whole
    // workgroup waited for it at the next barrier.
    if (wave == min(3, NW - 1)) {
      const int f = min(lane / 3, 3), d = lane - 3 * f - 1;
      const int fs = (f == 2 && P.sg_same_02) ? 0 : f;
      float v; int i;
      unpack_vi(S.sl->vi[VI_CUR0 + fs], &v, &i);
      const bool interior = i > P.cur_from[fs] && i < P.cur_until[fs];
      float ev = 0.f;
      if (lane < 12 && interior) ev = flt_at(fs, i + d);
      const float em = __shfl(ev, 3 * f), e0 = __shfl(ev, 3 * f + 1), ep = __shfl(ev, 3 * f + 2);
      if (interior) v = extrema3points(em, e0, ep);
      if (lane < 12 && d == -1) S.outv[f == 0 ? C_a_sg : f == 1 ? C_a_60 : f == 2 ? C_a_100 : C_a_raw] = v;
    }
    STAMP(15); DSTOP(15);
    __syncthreads();
    if (tid == 0) {
      S.outv[C_e_trap] = eslot[0];
      S.outv[C_qdrift] = eslot[1];
      S.outv[C_lq] = eslot[2];
    }
  }

  // ------------------------------------------------------------------------------------ phase 7: CUSP / ZAC (dsp_icpc.jl:167-178)
  // Closed form (CUSP / ZAC): with d[i] = y[i] - a*y[i-1],
  //   out[k] = sc * ( sum_{j<=Lf-2} w[j] d[n-j] + w[Lf-1] y[k] ),  n = k+Lf-1,
  // w = sinh flanks + flat top (+ parabolas for ZAC) splits into a causal one-pole G, an anti-causal one-pole A, the prefix sum
  // Dp of d, and a double prefix sum of a sparse combination u of Dp; each is built in the S4 view and read back lane-strided,
  // two rows per step.
  // One pass evaluates the filters that share Z's geometry: both (WC, WZ: the usual case, ZAC = CUSP + parabola corrections), or one
  // of them when the two were optimised separately (pars_filter): then CUSP first, y and T are put back, and ZAC follows with
  // Z = ZZ = P.zac (its own flat top / last tap come out of the same statements with dwl = 0).
  auto cz_pass = [&](auto wc_tag, auto wz_tag, const CuspZacDev& Z, const CuspZacDev& ZZ, const float cpiv) {
    constexpr bool WC = decltype(wc_tag)::value, WZ = decltype(wz_tag)::value;
    const int Lf = Z.Lf;
    const int nout = L - Lf + 1, lt = Z.lt, f1 = Z.f1, ltp = Z.ltp;
    const int pad = cz_pad_floats(Lf);
    for (int i = tid; i < pad; i += NT) S.B[i - pad] = 0.f;   // the gap (dead mask words) becomes Dp[i < 0] = 0
    if (tid < 64) S.B[Lp + tid] = 0.f;
    float yprev[R];   // y just before each of the thread's quads (for d[i] = y[i] - a*y[i-1])
#pragma unroll
    for (int r = 0; r < R; ++r) { const int i0 = 4 * (tid + NT * r); yprev[r] = (i0 > 0) ? S.A[i0 - 1] : 0.f; }
    // Pivot.  The parts of the closed form (flat top, sinh flanks, parabolas) each answer a constant level c under the filter
    // with a multiple of c * Lf that cancels between them only in exact arithmetic: a trace whose baseline sits 2000 counts
    // off zero (pile-up in the baseline window) loses 1e-4 of its ZAC energy to float rounding.  So the stage runs on
    // y' = y - cpiv, cpiv = the level at the left edge of the pick-off window (any constant is exact mathematically:
    // out = out' + cpiv * hsum, hsum = the sum of the direct-form taps), which makes every trace look like a clean one.
    // y' is never materialised:  Dp' = Dp - eps*cpiv*i,  d' = d - eps*cpiv,  the taps on y[k] fold cpiv into their fma.
    const float mec = -Z.eps * cpiv;
    // ---- Dp[i] = y[i] - y[0] + eps*T[i] -> B, in place of T (each thread converts its own quads)
    {
      const float y0 = S.A[0];
      const f2 e2 = splat(Z.eps), y02 = splat(y0);
      const float bf = (float)(4 * tid);   // T is the exclusive prefix sum: T'[i] = T[i] - cpiv*i
      const f2 l01 = splat(mec) * mk2(bf, bf + 1.f), l23 = splat(mec) * mk2(bf + 2.f, bf + 3.f);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        f4 t = *reinterpret_cast<const f4*>(&S.B[4 * (tid + NT * r)]);
        const f2 lr = splat(mec * (float)(4 * NT * r));
        t.xy = fma2(e2, t.xy, y[r].xy - y02) + (l01 + lr);
        t.zw = fma2(e2, t.zw, y[r].zw - y02) + (l23 + lr);
        *reinterpret_cast<f4*>(&S.B[4 * (tid + NT * r)]) = t;
      }
    }
    // ---- d[i] = (y[i]-y[i-1]) + eps*y[i-1] for 1 <= i < L, else 0   (S4)
    f2 d[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const f2 p0 = mk2(yprev[r], y[r].x), p1 = mk2(y[r].y, y[r].z);
      d[r][0] = fma2(splat(Z.eps), p0, (y[r].xy - p0) + splat(mec));
      d[r][1] = fma2(splat(Z.eps), p1, (y[r].zw - p1) + splat(mec));
      if (r == 0 && tid == 0) d[0][0].x = 0.f;
    }
    __syncthreads();
    STAMP(16); DSTOP(16);
    auto rd2 = [&](const float* p, int m) { return mk2(p[NT * m], p[NT * (m + 1)]); };
    // rows m and m+1 of a lane-strided array in ONE ds_write2st64_b32 (hipcc leaves them as two ds_write_b32 with different base
    // registers when the array lies behind the 64 KB an immediate offset reaches)
    auto wr2 = [&](float* p, int m, f2 v) {
      const uint32_t a = (uint32_t)(uintptr_t)(lds_float*)(p + NT * m);
      asm volatile("ds_write2st64_b32 %0, %1, %2 offset1:%3" : : "v"(a), "v"(v.x), "v"(v.y), "n"(NT / 64) : "memory");
    };
    // ---- flat top + last tap (LS); ZAC: u[n] = sum_e coef_e Dp[n - shift_e] -> A in place of y
    f2 ac[SP / 2], dz[SP / 2];
    {
      const f2 dwl = splat(ZZ.w_last - Z.w_last), wl = splat(Z.w_last), sc = splat(Z.sc);
      const f2 mwlc = splat(-Z.w_last * cpiv), mdwlc = splat(-(ZZ.w_last - Z.w_last) * cpiv);
      const float *ya = &S.A[tid], *dpa = &S.B[tid + Lf - 1 - lt], *dpb = &S.B[tid + Lf - 1 - f1];
#pragma unroll
      for (int m = 0; m < SP; m += 2) {
        const f2 yk = rd2(ya, m), pa = rd2(dpa, m), pb = rd2(dpb, m);
        ac[m / 2] = fma2(sc, pa - pb, fma2(wl, yk, mwlc));
        dz[m / 2] = fma2(dwl, yk, mdwlc);
        pin(ac[m / 2]); pin(dz[m / 2]);
        if ((m & 2) == 2) __builtin_amdgcn_sched_barrier(0);   // at most two pairs of rows of loads in flight (register pressure)
      }
      if constexpr (WZ) {
      f2 u[SP / 2];
#pragma unroll
      for (int m = 0; m < SP / 2; ++m) u[m] = splat(0.f);
      if (ZZ.zc_n == 9) {   // (block-uniform) the usual tap structure: one chain, every shift read once, links unrolled
        f2 prev[SP / 2];
        {
          const float* dp = &S.B[tid - ZZ.zc_s[0]];
#pragma unroll
          for (int m = 0; m < SP; m += 2) prev[m / 2] = rd2(dp, m);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const f2 ce = splat(ZZ.zc_r[e]);
          const float* dq = &S.B[tid - ZZ.zc_s[e + 1]];
#pragma unroll
          for (int m = 0; m < SP; m += 2) {
            const f2 cur = rd2(dq, m);
            u[m / 2] = fma2(ce, prev[m / 2] - cur, u[m / 2]);
            prev[m / 2] = cur;
          }
          pin(u[0]);
        }
      } else {
      const int nz = ZZ.zu_n;
      for (int e = 0; e < nz; ++e) {
        const f2 ce = splat(ZZ.zu_coef[e]);
        const float *dp = &S.B[tid - ZZ.zu_shift[e]], *dq = &S.B[tid - ZZ.zu_shift_b[e]];
#pragma unroll
        for (int m = 0; m < SP; m += 2) { u[m / 2] = fma2(ce, rd2(dp, m) - rd2(dq, m), u[m / 2]); pin(u[m / 2]); }
      }
      }
#pragma unroll
      for (int m = 0; m < SP; m += 2) wr2(&S.A[tid], m, u[m / 2]);   // own elements: race-free
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the compiler does not count the stores of an asm statement)
      }
    }
    STAMP(17); DSTOP(17);
    const float q1 = Z.qp1[1];
    const float* part_f;
    // ---- causal one-pole G -> B, rise(-) and fall(+) exponentials
    {
      float gl[R][4], b[R], s_in[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float g = d[r][0].x; gl[r][0] = g;
        g = fmaf(q1, g, d[r][0].y); gl[r][1] = g;
        g = fmaf(q1, g, d[r][1].x); gl[r][2] = g;
        g = fmaf(q1, g, d[r][1].y); gl[r][3] = g;
        b[r] = g;
      }
      s4_exscan_affine_fwd<NT, R>(b, s_in, Z.qp4, Z.qpw, S.part);   // barrier inside: the LS reads of Dp are done
      const f2 qa = mk2(Z.qp1[1], Z.qp1[2]), qb = mk2(Z.qp1[3], Z.qp1[4]);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const f2 s2 = splat(s_in[r]);
        const f2 va = fma2(qa, s2, mk2(gl[r][0], gl[r][1])), vb = fma2(qb, s2, mk2(gl[r][2], gl[r][3]));
        *reinterpret_cast<f4*>(&S.B[4 * (tid + NT * r)]) = (f4){va.x, va.y, vb.x, vb.y};
      }
    }
    __syncthreads();
    {
      const float *gn = &S.B[tid + Lf - 1], *gnl = &S.B[tid + Lf - 1 - lt], *gk = &S.B[tid], *gkl = &S.B[tid + ltp - 1];
      const f2 qlt = splat(Z.q_lt), qml = splat(Z.q_mltp), ql1 = splat(Z.q_ltp1), sh = splat(Z.sc_half_den);
#pragma unroll
      for (int m = 0; m < SP; m += 2) {
        const f2 pm = rd2(gn, m) - qlt * rd2(gnl, m);
        const f2 fp = qml * (rd2(gkl, m) - ql1 * rd2(gk, m));
        ac[m / 2] = fma2(sh, fp - pm, ac[m / 2]);
        pin(ac[m / 2]);
        if ((m & 2) == 2) __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(18); DSTOP(18);
    // ---- anti-causal one-pole A -> B, rise(+) and fall(-) exponentials
    {
      float al[R][4], b[R], s_in[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float a = d[r][1].y; al[r][3] = a;
        a = fmaf(q1, a, d[r][1].x); al[r][2] = a;
        a = fmaf(q1, a, d[r][0].y); al[r][1] = a;
        a = fmaf(q1, a, d[r][0].x); al[r][0] = a;
        b[r] = a;
      }
      s4_exscan_affine_bwd<NT, R>(b, s_in, Z.qp4, Z.qpw, S.part + R * NW);   // barrier inside: the LS reads of G are done
      const f2 qa = mk2(Z.qp1[4], Z.qp1[3]), qb = mk2(Z.qp1[2], Z.qp1[1]);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const f2 s2 = splat(s_in[r]);
        const f2 va = fma2(qa, s2, mk2(al[r][0], al[r][1])), vb = fma2(qb, s2, mk2(al[r][2], al[r][3]));
        *reinterpret_cast<f4*>(&S.B[4 * (tid + NT * r)]) = (f4){va.x, va.y, vb.x, vb.y};
      }
      if (tid == 0) S.B[Lp] = 0.f;   // A[L]
    }
    __syncthreads();
    {
      const float *a1 = &S.B[tid + Lf - lt], *a2 = &S.B[tid + Lf], *a3 = &S.B[tid + 1], *a4 = &S.B[tid + ltp];
      const f2 qm1 = splat(Z.q_mlt1), qq1 = splat(Z.q1), qq2 = splat(Z.q2), ql1 = splat(Z.q_ltp1), sh = splat(Z.sc_half_den);
#pragma unroll
      for (int m = 0; m < SP; m += 2) {
        const f2 pp = qm1 * rd2(a1, m) - qq1 * rd2(a2, m);
        const f2 fm = qq2 * (rd2(a3, m) - ql1 * rd2(a4, m));
        ac[m / 2] = fma2(sh, pp - fm, ac[m / 2]);
        pin(ac[m / 2]);
        if ((m & 2) == 2) __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(19); DSTOP(19);
    if constexpr (WZ) {
    // ---- ZAC parabolas: PRF = cumsum(cumsum(u)) (S4; u was parked in A).  Two levels: inside a wave-row (256 samples) the
    // single and double running sums l1, l2 start from zero and stay in float; the state entering each wave-row, (C1, C2), is
    // carried in double:  c2[j] = C2 + (j+1)*C1 + l2[j],  C1' = C1 + l1[255],  C2' = C2 + 256*C1 + l2[255].
    {
      float ex1[R], ex2[R], i1[R], i2[R], p3[R];
      f4 uq[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        uq[r] = *reinterpret_cast<const f4*>(&S.A[4 * (tid + NT * r)]);
        const float p0 = uq[r].x, p1 = p0 + uq[r].y, p2 = p1 + uq[r].z;
        p3[r] = p2 + uq[r].w;
        i1[r] = p3[r];
        i2[r] = (p0 + p1) + (p2 + p3[r]);   // the quad's own contribution to the double sum
      }
      LDSP_DPP_GROUP4("v_add_f32_dpp", i1[0], "v_add_f32_dpp", i1[1], "v_add_f32_dpp", i1[2], "v_add_f32_dpp", i1[3]);
#pragma unroll
      for (int r = 0; r < R; ++r) { ex1[r] = i1[r] - p3[r]; i2[r] = fmaf(4.f, ex1[r], i2[r]); ex2[r] = i2[r]; }
      LDSP_DPP_GROUP4("v_add_f32_dpp", i2[0], "v_add_f32_dpp", i2[1], "v_add_f32_dpp", i2[2], "v_add_f32_dpp", i2[3]);
#pragma unroll
      for (int r = 0; r < R; ++r) ex2[r] = i2[r] - ex2[r];
      double* pa = S.dpart; double* pb = S.dpart + R * NW;
      if (lane == 63) {
#pragma unroll
        for (int r = 0; r < R; ++r) { pa[r * NW + wave] = (double)i1[r]; pb[r * NW + wave] = (double)i2[r]; }
      }
      __syncthreads();
      const double t1 = (lane < R * NW) ? pa[lane] : 0.0;
      const double c1x = wave_incl_scan_sum_f64(t1) - t1;
      const double t2 = (lane < R * NW) ? pb[lane] + 256.0 * c1x : 0.0;
      const double c2x = wave_incl_scan_sum_f64(t2) - t2;
      const float mrho = -ZZ.rho_sc;
      const double jd0 = (double)(4 * lane + 1);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double C1 = readlane_d(c1x, r * NW + wave), C2 = readlane_d(c2x, r * NW + wave);
        // c2 at the quad's samples = t + e*C1 + l2[e],  t = C2 + (j+1) C1 at the first one; split t and C1 into float parts
        const double t = fma(C1, jd0, C2);
        const float th = (float)t, tl = (float)(t - (double)th), c1f = (float)C1;
        float l1 = ex1[r], l2 = ex2[r];
        f2 la, lb;
        l1 += uq[r].x; l2 += l1; la.x = l2;
        l1 += uq[r].y; l2 += l1; la.y = l2;
        l1 += uq[r].z; l2 += l1; lb.x = l2;
        l1 += uq[r].w; l2 += l1; lb.y = l2;
        const f2 c2 = splat(c1f), t2 = splat(tl), h2 = splat(th), m2 = splat(mrho);
        const f2 va = m2 * (h2 + (t2 + fma2(c2, mk2(0.f, 1.f), la))), vb = m2 * (h2 + (t2 + fma2(c2, mk2(2.f, 3.f), lb)));
        *reinterpret_cast<f4*>(&S.A[4 * (tid + NT * r)]) = (f4){va.x, va.y, vb.x, vb.y};
      }
    }
    __syncthreads();
    {
      const float* pr = &S.A[tid + Lf - 1];
#pragma unroll
      for (int m = 0; m < SP; m += 2) { dz[m / 2] += rd2(pr, m) + ac[m / 2]; pin(dz[m / 2]); }
    }
    }
    STAMP(20); DSTOP(20);
    // ---- extremestats + SignalEstimator of both outputs (dsp_icpc.jl:170-171,177-178): value first, then its first index
    float mxc = -INFINITY, mxz = -INFINITY;
    float pc = 0.f, pz_ = 0.f;
    float own_c, own_z;   // this thread's maxima (kept for the index look-up)
    {
#pragma unroll
      for (int m = 0; m < SP; m += 2) {
        f2 a = ac[m / 2], z = dz[m / 2];
        if (NT * (m + 2) > nout) {
          const bool in0 = tid + NT * m < nout, in1 = tid + NT * (m + 1) < nout;
          a.x = in0 ? a.x : -INFINITY; a.y = in1 ? a.y : -INFINITY; z.x = in0 ? z.x : -INFINITY; z.y = in1 ? z.y : -INFINITY;
        }
        mxc = vmax3(mxc, a.x, a.y); mxz = vmax3(mxz, z.x, z.y);
      }
      own_c = mxc; own_z = mxz;
      // estimator window [i0, i0+npts) (found by wave 0 in phase 6): at most one output per thread (npts <= 64 <= NT), in at
      // most two waves — the others skip the look-up of that output among their sixteen (wave-uniform test)
      if (nout >= P.sig_est.npts) {
        const f4 ew = (f4){S.misc[8], S.misc[9], S.misc[10], S.misc[11]};   // i0 (bits) and u of the CUSP and of the ZAC estimate
        const int i0c = __float_as_int(ew.x), i0z = __float_as_int(ew.z);
        const int msc = (i0c - tid + NT - 1) / NT, msz = (i0z - tid + NT - 1) / NT;   // smallest m with tid + NT*m >= i0
        const int lc = tid + NT * msc - i0c, lz = tid + NT * msz - i0z;
        const bool inc_ = lc >= 0 && lc < P.sig_est.npts && msc >= 0 && msc < SP, inz = lz >= 0 && lz < P.sig_est.npts && msz >= 0 && msz < SP;
        if (__ballot(inc_ || inz) != 0ull) {
          float vc_ = 0.f, vz_ = 0.f;
#pragma unroll
          for (int m = 0; m < SP; ++m) {
            vc_ = (m == msc) ? ((m & 1) ? ac[m / 2].y : ac[m / 2].x) : vc_;
            vz_ = (m == msz) ? ((m & 1) ? dz[m / 2].y : dz[m / 2].x) : vz_;
          }
          if (inc_) pc = est_weight(P.sig_est, S.estB, lc, ew.y) * vc_;
          if (inz) pz_ = est_weight(P.sig_est, S.estB, lz, ew.w) * vz_;
        }
      }
      LDSP_DPP_GROUP4("v_add_f32_dpp", pc, "v_add_f32_dpp", pz_, "v_max_f32_dpp", mxc, "v_max_f32_dpp", mxz);
      if (lane == 63) {
        if (WC) { S.wsum[(W_CZ + 0) * NW + wave] = pc; atomicMax(&S.sl->fmx[FX_CUSP], ford(mxc)); }
        if (WZ) { S.wsum[(W_CZ + 1) * NW + wave] = pz_; atomicMax(&S.sl->fmx[FX_ZAC], ford(mxz)); }
      }
    }
    __syncthreads();
    {
      const float vc = ford_inv(S.sl->fmx[FX_CUSP]), vz = ford_inv(S.sl->fmx[FX_ZAC]);
      if (__ballot((WC && own_c == vc) || (WZ && own_z == vz)) != 0ull) {   // only the waves that hold a maximum look its index up
        int bc = 0x7fffffff, bz = 0x7fffffff;
#pragma unroll
        for (int m = SP - 1; m >= 0; --m) {   // findmax: first occurrence
          const float a = (m & 1) ? ac[m / 2].y : ac[m / 2].x, z = (m & 1) ? dz[m / 2].y : dz[m / 2].x;
          const bool in = tid + NT * m < nout;
          bc = (in && a == vc) ? tid + NT * m : bc;
          bz = (in && z == vz) ? tid + NT * m : bz;
        }
        if (WC && bc != 0x7fffffff) atomicMin(&S.sl->imin[IM_CUSP], bc);
        if (WZ && bz != 0x7fffffff) atomicMin(&S.sl->imin[IM_ZAC], bz);
      }
    }
    STAMP(21); DSTOP(21);
    __syncthreads();
    if (tid < 2 && (tid == 0 ? WC : WZ)) {
      const int f = tid;
      float s = 0.f;
      for (int ww = 0; ww < NW; ++ww) s += S.wsum[(W_CZ + f) * NW + ww];
      const float v = ford_inv(S.sl->fmx[f ? FX_ZAC : FX_CUSP]);
      const int i = S.sl->imin[f ? IM_ZAC : IM_CUSP];
      const double back = (double)cpiv * (f ? ZZ.hsum : Z.hsum);   // the pivot's share of the output (estimator weights sum to one)
      S.outv[f ? C_e_zac : C_e_cusp] = (nout >= P.sig_est.npts) ? (float)((double)s + back) : NAN;
      S.outv[f ? C_e_zac_max : C_e_cusp_max] = (float)((double)v + back);
      S.outv[f ? C_t_zac_max : C_t_cusp_max] = P.t_first + P.dt * (float)(i + Lf - 1);
    }
  };
  using T_ = std::true_type; using F_ = std::false_type;
  if constexpr (!SEP) {
    cz_pass(T_{}, T_{}, P.cusp, P.zac, S.misc[12]);
  } else {
    cz_pass(T_{}, F_{}, P.cusp, P.zac, S.misc[12]);
    // put T back (the pass turned B into the anti-causal scan and left A = y alone): the statements of phase 4
    __syncthreads();
    {
      float tin[R], tt[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        y[r] = *reinterpret_cast<const f4*>(&S.A[4 * (tid + NT * r)]);   // A still holds y (the CUSP pass does not write it); read back so
        const f2 t = y[r].xy + y[r].zw; tt[r] = t.x + t.y; tin[r] = tt[r];   // that the registers are free during that pass
      }
      LDSP_DPP_GROUP4("v_add_f32_dpp", tin[0], "v_add_f32_dpp", tin[1], "v_add_f32_dpp", tin[2], "v_add_f32_dpp", tin[3]);
      float* pb = S.part + R * NW;
      if (lane == 63) *reinterpret_cast<f4*>(&pb[4 * wave]) = (f4){tin[0], tin[1], tin[2], tin[3]};
      __syncthreads();
      float* scn2 = reinterpret_cast<float*>(S.dpart);
      t_offsets_scan<NW>(pb, scn2 + R * NW, &S.B[Lp], wave, lane);
      __syncthreads();
      t_rows_store<NT>(y, tin, tt, scn2 + R * NW, S.B, tid, wave);
    }
    __syncthreads();
    cz_pass(F_{}, T_{}, P.zac, P.zac, S.misc[13]);
  }
  STAMP(22); DSTOP(22);
  // ------------------------------------------------------------------------------------------------ outputs
  __syncthreads();   // the output row was filled by lanes of different waves
  static_assert(C_NCOLS <= 64, "the output row is stored by wave 0");
  if (tid < C_NCOLS) {
    float* dst = reinterpret_cast<float*>(out.col[tid]);
    if (dst) dst[(size_t)blockIdx.x * (size_t)out.stride] = S.outv[tid];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// BASELINE config 2: blmean -> shift -> InvCR -> Trap(10 us, 4 us) -> maximum (reference src/dsp_icpc.jl:102-105,119-120,
// 147-148).  The same statements as icpc_lean_kernel, nothing else: blmean and e_10410 come out bit-identical to the fused
// chain's columns (tests/test_baseline_sizes_gpu.py).  One trace-sized LDS array (T): four workgroups per CU.
// U16: the traces are uint16 ADC counts, converted as they are loaded — a template parameter, not the block's in_u16 field: a
// scalar load + branch in front of the trace loads of this memory-bound kernel cost 4 % (5.68 -> 5.90 ms per 10^6 traces).
template <int NT, bool U16>
__global__ void __launch_bounds__(NT, 8)
pz_trap_lean_kernel(const float* __restrict__ wf, const IcpcDev* __restrict__ Pp, float* __restrict__ o_blmean, float* __restrict__ o_e10410) {
  constexpr int NW = NT / 64, Lp = NT * SP, L = Lp;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const IcpcDev& P = *Pp;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* B = reinterpret_cast<float*>(smem_raw);           // [Lp + 64] T, + 2 NT floats of slack: the lanes of the row pair that
  float* part = B + Lp + 64 + 2 * NT;                      // holds the end of the output range read past T (masked)   [2][R*NW]
  float* scn = part + 2 * R * NW;                          // [3][R*NW]: what wave 0 makes of the wave-row totals (row_prefix_f64)
  float* wred = scn + 3 * R * NW;                          // [2][NW]: s1 partials, trapezoid maxima
  const float* w = wf + (size_t)blockIdx.x * (size_t)L;
  const uint16_t* w16 = reinterpret_cast<const uint16_t*>(wf) + (size_t)blockIdx.x * (size_t)L;   // (in_u16: ADC counts, converted here)
  f4 x[R];
  if constexpr (U16) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint2 q = *reinterpret_cast<const uint2*>(w16 + 4 * (tid + NT * r));
      x[r] = (f4){(float)(q.x & 0xffffu), (float)(q.x >> 16), (float)(q.y & 0xffffu), (float)(q.y >> 16)};
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = *reinterpret_cast<const f4*>(w + 4 * (tid + NT * r));
  }
  asm volatile("; LDSP_PHASE 1");
  const float pv_bl = U16 ? (float)w16[P.bl.from] : w[P.bl.from];
  const uint32_t cls_bl = P.rowcls[0][wave];
  {   // baseline sum: the s1 chain of icpc_lean_kernel's phase 1
    f2 a1 = splat(0.f);
    const f2 pv = splat(pv_bl);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if ((cls_bl >> r) & 1u) continue;
      f2 d0 = x[r].xy - pv, d1 = x[r].zw - pv;
      if (!((cls_bl >> (4 + r)) & 1u)) {
        const int lo = P.bl.from - 4 * (tid + NT * r), hi = P.bl.until - 4 * (tid + NT * r);
        d0.x = (lo <= 0 && hi >= 0) ? d0.x : 0.f; d0.y = (lo <= 1 && hi >= 1) ? d0.y : 0.f;
        d1.x = (lo <= 2 && hi >= 2) ? d1.x : 0.f; d1.y = (lo <= 3 && hi >= 3) ? d1.y : 0.f;
      }
      a1 += d0 + d1;
    }
    float s1 = hsum(a1);
    LDSP_DPP_GROUP1("v_add_f32_dpp", s1);
    if (lane == 63) wred[wave] = s1;
  }
  asm volatile("; LDSP_PHASE 2");
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int ww = 0; ww < NW; ++ww) s += wred[ww];
  const float blmean = fmaf(s, (float)P.bl.inv_n, pv_bl);
  float inc[R], tot[R];
  {
    const f2 bm2 = splat(blmean);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      x[r].xy -= bm2; x[r].zw -= bm2;
      const f2 t = x[r].xy + x[r].zw;
      tot[r] = t.x + t.y; inc[r] = tot[r];
    }
    LDSP_DPP_GROUP4("v_add_f32_dpp", inc[0], "v_add_f32_dpp", inc[1], "v_add_f32_dpp", inc[2], "v_add_f32_dpp", inc[3]);
    if (lane == 63) *reinterpret_cast<f4*>(&part[4 * wave]) = (f4){inc[0], inc[1], inc[2], inc[3]};
  }
  __syncthreads();
  asm volatile("; LDSP_PHASE 3");
  pz_offsets_scan<NW>(part, scn, P.pz_c64, wave, lane);
  __syncthreads();
  pz_apply<NW>(x, inc, tot, scn, P.pz_c, wave);
  asm volatile("; LDSP_PHASE 4");
  auto& y = x;
  {
    float tin[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const f2 t = y[r].xy + y[r].zw; tot[r] = t.x + t.y; tin[r] = tot[r]; }
    LDSP_DPP_GROUP4("v_add_f32_dpp", tin[0], "v_add_f32_dpp", tin[1], "v_add_f32_dpp", tin[2], "v_add_f32_dpp", tin[3]);
    float* pb = part + R * NW;
    if (lane == 63) *reinterpret_cast<f4*>(&pb[4 * wave]) = (f4){tin[0], tin[1], tin[2], tin[3]};
    __syncthreads();
    t_offsets_scan<NW>(pb, scn + R * NW, &B[Lp], wave, lane);
    __syncthreads();
    t_rows_store<NT>(y, tin, tot, scn + R * NW, B, tid, wave);
    if (tid >= 1 && tid < 64) B[Lp + tid] = 0.f;
  }
  __syncthreads();
  asm volatile("; LDSP_PHASE 5");
  float mx0 = -INFINITY;
  {
    const TrapDev f0 = P.fixed[0];
    const float* tb = &B[tid];
    const float *f0a = tb + f0.n1, *f0b = tb + f0.n1 + f0.g, *f0c = tb + f0.flen;
    const f2 rr0 = splat(f0.rr);
    const int n0 = L - f0.flen + 1;
    auto rd2 = [&](const float* p, int m) { return mk2(p[NT * m], p[NT * (m + 1)]); };
    // np full row pairs, then (maybe) the pair that holds the end of the output range: a jump into a run of straight-line
    // pairs instead of two range tests per pair (the row offsets stay immediates)
    auto pair = [&](auto mtag, bool whole) {
      constexpr int m = decltype(mtag)::value;
      const f2 Tk = rd2(tb, m), a0 = rd2(f0a, m), b0 = rd2(f0b, m), c0_ = rd2(f0c, m);
      const f2 o0 = fma2(c0_ - b0, rr0, Tk - a0);
      if (whole) mx0 = vmax3(mx0, o0.x, o0.y);
      else mx0 = vmax3(mx0, tid + NT * m < n0 ? o0.x : -INFINITY, tid + NT * (m + 1) < n0 ? o0.y : -INFINITY);
    };
    static_assert(SP == 16, "eight row pairs");
#define LDSP_PAIR(k) std::integral_constant<int, 2 * (k)>{}
    const int np = n0 / (2 * NT);   // pairs wholly inside the output range (block-uniform)
    switch (np) {
      case 8: pair(LDSP_PAIR(7), true); [[fallthrough]];
      case 7: pair(LDSP_PAIR(6), true); [[fallthrough]];
      case 6: pair(LDSP_PAIR(5), true); [[fallthrough]];
      case 5: pair(LDSP_PAIR(4), true); [[fallthrough]];
      case 4: pair(LDSP_PAIR(3), true); [[fallthrough]];
      case 3: pair(LDSP_PAIR(2), true); [[fallthrough]];
      case 2: pair(LDSP_PAIR(1), true); [[fallthrough]];
      case 1: pair(LDSP_PAIR(0), true); [[fallthrough]];
      default: break;
    }
    if (n0 > 2 * NT * np) {
      switch (np) {
        case 0: pair(LDSP_PAIR(0), false); break;
        case 1: pair(LDSP_PAIR(1), false); break;
        case 2: pair(LDSP_PAIR(2), false); break;
        case 3: pair(LDSP_PAIR(3), false); break;
        case 4: pair(LDSP_PAIR(4), false); break;
        case 5: pair(LDSP_PAIR(5), false); break;
        case 6: pair(LDSP_PAIR(6), false); break;
        default: pair(LDSP_PAIR(7), false); break;
      }
    }
#undef LDSP_PAIR
    mx0 *= f0.inv1;
  }
  asm volatile("; LDSP_PHASE 6");
  LDSP_DPP_GROUP1("v_max_f32_dpp", mx0);
  if (lane == 63) wred[NW + wave] = mx0;
  __syncthreads();
  if (tid == 0) {
    float m = wred[NW];
    for (int ww = 1; ww < NW; ++ww) m = vmax(m, wred[NW + ww]);
    o_blmean[blockIdx.x] = blmean;
    o_e10410[blockIdx.x] = m;
  }
}

template <int NT, bool U16>
static hipError_t launch_pz_tu(const float* wf, int64_t n, const IcpcDev* dP, float* a, float* b, hipStream_t st) {
  constexpr int NW = NT / 64;
  const size_t smem = (size_t)(NT * SP + 64 + 2 * NT + 5 * R * NW + 2 * NW) * 4 + 16;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pz_trap_lean_kernel<NT, U16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((pz_trap_lean_kernel<NT, U16>), dim3((unsigned)n), dim3(NT), smem, st, wf, dP, a, b);
  return hipGetLastError();
}
template <int NT>
static hipError_t launch_pz_t(const float* wf, int64_t n, bool u16, const IcpcDev* dP, float* a, float* b, hipStream_t st) {
  return u16 ? launch_pz_tu<NT, true>(wf, n, dP, a, b, st) : launch_pz_tu<NT, false>(wf, n, dP, a, b, st);
}

// ---------------------------------------------------------------------------------------------------------------------
template <int NT, int M, bool SEP>
static hipError_t launch_t(const float* wf, int64_t n, const IcpcDev* dP, const IcpcOutDev& out, const float* ext_bl, float ext_bl_scale,
                           int Lf, hipStream_t st) {
  const size_t smem = Smem<NT>::bytes(cz_pad_floats(Lf)) + (size_t)g_dbg_lds_pad;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&icpc_lean_kernel<NT, M, SEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((icpc_lean_kernel<NT, M, SEP>), dim3((unsigned)n), dim3(NT), smem, st, wf, dP, out, ext_bl, ext_bl_scale);
  return hipGetLastError();
}

}  // namespace lean
// LDS bytes of the lean kernel for a tile of NT threads and CUSP/ZAC filters of Lf taps (the host checks it against the
// two-traces-per-CU budget before choosing this kernel)
size_t icpc_lean_smem_bytes(int NT, int Lf) {
  switch (NT) {
    case 64: return lean::Smem<64>::bytes(lean::cz_pad_floats(Lf));
    case 128: return lean::Smem<128>::bytes(lean::cz_pad_floats(Lf));
    case 256: return lean::Smem<256>::bytes(lean::cz_pad_floats(Lf));
    case 512: return lean::Smem<512>::bytes(lean::cz_pad_floats(Lf));
    case 1024: return lean::Smem<1024>::bytes(lean::cz_pad_floats(Lf));
    default: return (size_t)-1;
  }
}

hipError_t launch_pz_trap_lean(const float* wf, int64_t n, int NT, bool u16, const IcpcDev* dP, float* blmean, float* e10410, hipStream_t st) {
  switch (NT) {
#ifndef LDSP_DEV_512
    case 64: return lean::launch_pz_t<64>(wf, n, u16, dP, blmean, e10410, st);
    case 128: return lean::launch_pz_t<128>(wf, n, u16, dP, blmean, e10410, st);
    case 256: return lean::launch_pz_t<256>(wf, n, u16, dP, blmean, e10410, st);
    case 1024: return lean::launch_pz_t<1024>(wf, n, u16, dP, blmean, e10410, st);
#endif
    case 512: return lean::launch_pz_t<512>(wf, n, u16, dP, blmean, e10410, st);
    default: return hipErrorInvalidValue;
  }
}

// sg_slots: 7 or 13 (the smallest that holds the three Savitzky-Golay windows)
hipError_t launch_icpc_lean(const float* wf, int64_t n, int NT, int sg_slots, bool cz_shared, const IcpcDev* dP, const IcpcOutDev& out,
                            const float* ext_bl, float ext_bl_scale, int Lf, hipStream_t st) {
#ifdef LDSP_DEV_512
#define LDSP_LEAN_CASES LDSP_CASE(512)
#else
#define LDSP_LEAN_CASES LDSP_CASE(64) LDSP_CASE(128) LDSP_CASE(256) LDSP_CASE(512) LDSP_CASE(1024)
#endif
#define LDSP_CASE(N) \
  case N: return cz_shared ? (sg_slots <= 7 ? lean::launch_t<N, 7, false>(wf, n, dP, out, ext_bl, ext_bl_scale, Lf, st) : lean::launch_t<N, 13, false>(wf, n, dP, out, ext_bl, ext_bl_scale, Lf, st)) \
                           : (sg_slots <= 7 ? lean::launch_t<N, 7, true>(wf, n, dP, out, ext_bl, ext_bl_scale, Lf, st) : lean::launch_t<N, 13, true>(wf, n, dP, out, ext_bl, ext_bl_scale, Lf, st));
  switch (NT) {
    LDSP_LEAN_CASES
    default: return hipErrorInvalidValue;
  }
#undef LDSP_CASE
}

}  // namespace ldsp
