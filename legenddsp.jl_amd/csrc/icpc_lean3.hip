// icpc_lean3.hip — the fused dsp_icpc kernel (reference src/dsp_icpc.jl:62-230) for the standard geometry, round 3:
// THREE workgroups per CU.  Measured on the round-2 kernel (tools/occ_probe.py, tools/stamp_map.py dbg_lds_pad=60000): a workgroup
// lives 61 000 cycles alone on its CU and 71 800 next to a second one — the chain is a sequence of latencies (DPP chains, LDS
// round trips, ~20 barriers), the SIMDs and the LDS pipe are half idle, and throughput is nearly proportional to the number of
// resident workgroups.  A third one needs <= 53 760 B of LDS and <= 80 VGPRs:
//   * ONE trace-sized LDS array X, used in turns:  T (prefix sum of y: the trapezoid sweeps)  ->  y (Savitzky-Golay halo,
//     crossings, estimators, pile-up)  ->  Dp, u, PRF, Dp, G, A of the CUSP / ZAC stage.  y lives in registers (S4 view) while X
//     holds T; the few values of T-based filters that are needed later (the t0 crossing, the 44 points of the e_trap estimate)
//     are window sums of y taken by one wave — more accurate than differences of a float prefix sum of 1e8.
//   * sweep A (the t0 trapezoid's two threshold masks) runs in the S4 view: its 2-sample first leg comes from the thread's own
//     registers (+ two samples of the next lane by DPP), the long leg from two quad reads of T at the trapezoid's shifts
//     (ds_read_b128 / 2 x b64 / b32 + b64 + b32 by the shift's alignment: tools/micro/lds_s4shift_test.hip);
//   * the Savitzky-Golay output never goes to LDS: the pile-up and half-maximum masks are built from the registers that hold
//     it (four predicate bits per thread, eight lanes OR-ed into a word by three DPP steps);
//   * CUSP / ZAC: the chain of ZAC taps runs in two halves of the rows, the double prefix sum re-reads its own quads, and the
//     parabola kernel's last tap is folded into that chain: at most three 16-sample register arrays are live at any time.
// Same phases, same summation rules and the same 48 columns as the round-2 kernel; every other parameter set runs icpc_kernel.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include <stddef.h>
#include "icpc_dev.hpp"
#include "ldsp_device.hpp"
#include "qdrift.hpp"

#ifdef LDSP_STAMPS
#define STAMP(id) do { asm volatile("; LDSP_PHASE " #id); if ((threadIdx.x & 63) == 0 && blockIdx.x < LDSP_STAMP_BLOCKS && P.dbg_stamps) \
    P.dbg_stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * LDSP_STAMP_SLOTS + (id)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(id) asm volatile("; LDSP_PHASE " #id)
#endif
#ifdef LDSP_DSTOP
#define DSTOP(id) do { if (P.dbg_stop == 100 + (id)) return; } while (0)
#else
#define DSTOP(id) do { } while (0)
#endif

// what-if builds (timing experiments with wrong results, never shipped): -DLDSP_WHATIF_NOBAR_CZ / -DLDSP_WHATIF_NOBAR_MAIN drop the barriers
#ifdef LDSP_WHATIF_NOBAR_CZ
#define LDSP_BAR_CZ() ((void)0)
#else
#define LDSP_BAR_CZ() __syncthreads()
#endif
#ifdef LDSP_WHATIF_NOBAR_MAIN
#define LDSP_BAR_MAIN() ((void)0)
#else
#define LDSP_BAR_MAIN() __syncthreads()
#endif

// sensitivity probes (timing experiments: -DLDSP_PROBE_VALU=n / _LDS=n / _SALU=n adds n instructions of that kind at each of eight places)
#ifndef LDSP_PROBE_VALU
#define LDSP_PROBE_VALU 0
#endif
#ifndef LDSP_PROBE_LDS
#define LDSP_PROBE_LDS 0
#endif
#ifndef LDSP_PROBE_SALU
#define LDSP_PROBE_SALU 0
#endif
#define LDSP_PROBE() do { \
    if (LDSP_PROBE_VALU) { float pa_ = 1.f, pb_ = 2.f; _Pragma("unroll") for (int pi_ = 0; pi_ < LDSP_PROBE_VALU; pi_ += 4) \
        asm volatile("v_fmac_f32_e32 %0, %1, %1\n\tv_fmac_f32_e32 %1, %0, %0\n\tv_fmac_f32_e32 %0, %1, %1\n\tv_fmac_f32_e32 %1, %0, %0" : "+v"(pa_), "+v"(pb_)); } \
    if (LDSP_PROBE_LDS) { float pl_[4]; _Pragma("unroll") for (int pi_ = 0; pi_ < LDSP_PROBE_LDS; pi_ += 4) { \
        _Pragma("unroll") for (int pj_ = 0; pj_ < 4; ++pj_) pl_[pj_] = S.X[threadIdx.x + NT * (pi_ + pj_)]; \
        asm volatile("" :: "v"(pl_[0]), "v"(pl_[1]), "v"(pl_[2]), "v"(pl_[3])); } } \
    if (LDSP_PROBE_SALU) { int ps_ = 1; _Pragma("unroll") for (int pi_ = 0; pi_ < LDSP_PROBE_SALU; pi_ += 4) \
        asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1" : "+s"(ps_)); } \
  } while (0)
namespace ldsp {
extern int g_dbg_lds_pad;   // option "dbg_lds_pad" (defined in icpc_lean.hip): a profiling aid (tools/occ_probe.py), process-global — it inflates the LDS
                            // request of EVERY context's launches, after the admission check: a pad beyond the CU's LDS makes the launch fail with a HIP error
namespace lean3 {

typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) float __attribute__((ext_vector_type(4))) lds_f4;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

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

constexpr int R = 4, SP = 16;
constexpr int EST_TBL = LDSP_MAX_EST_PTS * (LDSP_MAX_EST_DEG + 1);
enum { M_T0, M_T0INV, M_INTR, M_FB, M_SG50 = 8, NMASKROWS = 9 };   // M_FB..M_FB+4: the general scan of the five y thresholds (rare)
enum { W_TAIL = 0, W_SGB = 3, W_PZ = 5, W_CZ = 8, NWSUM = 10 };   // rows of the per-wave window partial sums

__device__ __forceinline__ f2 mk2(float a, float b) { f2 v; v.x = a; v.y = b; return v; }
__device__ __forceinline__ f2 splat(float a) { return mk2(a, a); }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float hsum(f2 v) { return v.x + v.y; }
__device__ __forceinline__ void pin(f2& v) { asm volatile("" : "+v"(v)); }
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
__device__ __forceinline__ float est_weight(const EstDev& E, const float* Bt, int l, float u) { return dni_weight(E, Bt, l, u); }
__device__ __forceinline__ void est_window(const EstDev& E, Pos p, int nsig, int* i0, float* u) {
  if (p.ip < 0) { p.ip = 0; p.fp = 0.f; }
  if (p.ip >= nsig - 1) { p.ip = nsig - 1; p.fp = 0.f; }
  int a = p.ip + (int)ceilf(p.fp - 0.5f * (float)E.npts);
  a = max(0, min(a, nsig - E.npts));
  *i0 = a;
  *u = ((float)(p.ip - a) + p.fp - E.c) * E.s_inv;
}
__device__ __forceinline__ float wave_total(float v) {   // sum over the wave, in every lane
  LDSP_DPP_GROUP1("v_add_f32_dpp", v);
  return readlane_f(v, 63);
}
// sum of Y[from .. from + n) by one wave, in every lane (window sums of y in place of differences of a float prefix sum)
__device__ __forceinline__ float wave_box(const float* Y, int from, int n, int lane) {
  float acc = 0.f;
  for (int j = lane; j < n; j += 64) acc += Y[from + j];
  return wave_total(acc);
}

__device__ __forceinline__ void win_finish(float s1, float s2, float sx, const WinDev& w, float pivot, float t_first, float dt,
                                           float* mean, float* sigma, float* slope, float* offset) {
  const float inv_n = (float)w.inv_n;
  const float md = s1 * inv_n, m = pivot + md;
  const float var = fmaxf(fmaf(s2, inv_n, -md * md), 0.f);
  const float sl = (sx * inv_n) * __builtin_amdgcn_rcpf((float)w.var_i * dt);
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
static_assert(sizeof(Slots) % 8 == 0, "Slots keeps the 8-byte alignment of what follows");
enum { VI_OPT, VI_CUR0, VI_CUR1, VI_CUR2, VI_CUR3 };
enum { FX_F0, FX_F1, FX_F2, FX_G, FX_F0I, FX_F2I, FX_CUSP, FX_ZAC };
enum { IS_TAILBAD, IS_T0, IS_T0INV, IS_INTR };
enum { IM_TX0 = 0, IM_T0 = 5, IM_T0INV = 6, IM_SG50 = 7, IM_CUSP = 8, IM_ZAC = 9 };

__host__ __device__ inline int cz_pad_floats(int Lf) { return (Lf + 2 + 7) & ~3; }

// LDS: [gap: mask words / Dp[i < 0] = 0][X: Lp + 64][8-byte data][4-byte data][slack]: a lane-strided read X[k + shift] of the row
// pair that holds the end of an output range runs at most 2 NT floats past X (masked by the caller): those addresses stay
// inside the allocation.
constexpr int NH_MAX = 6;   // halo quads of the widest Savitzky-Golay window the kernel takes (25 taps: 4 + 24 samples)
template <int NT>
struct Smem {
  static constexpr int NW = NT / 64, Lp = NT * SP, NWORDS = Lp / 32;
  static constexpr int NSMALL = 2 * R * NW /*part*/ + 3 * R * NW /*scn*/ + 5 * NW /*wred*/ + NWSUM * NW /*wsum*/ + C_NCOLS + 32 /*outv, misc*/ +
                                2 * EST_TBL + (4 * NH_MAX + 1) * (R * NW + 1) /*hy: up to NH_MAX halo quads + the last sample per wave-row*/;
  uint32_t* bm;    // [NMASKROWS][NWORDS] in the gap in front of X
  float* X;        // [Lp + 64]
  double* dpart;   // [2][R*NW]  double prefix sum of the ZAC parabolas
  Slots* sl;
  float* part;     // [2][R*NW]  wave-row totals of the block scans (alternating buffers)
  float* scn;      // [3][R*NW]  what wave 0 makes of them (pz offsets; T offsets hi, lo)
  float* estB;     // [2][EST_TBL]
  float* hy;       // [R*NW + 1][NH] first NH quads of y of every wave-row (time order; the last entry: zeros) = the halo of a wave's
                   // last lanes; then [R*NW + 1]: the last sample of the wave-row BEFORE wave-row j
  lds_float* wred;     // [5][NW]    phase-1 per-wave partials: s1, s2, sx, max, min
  lds_float* wsum;     // [NWSUM][NW]
  lds_float* outv;     // [C_NCOLS]
  lds_float* misc;     // [32]
  static constexpr int gap_floats(int cz_pad) { return NMASKROWS * NWORDS > cz_pad ? NMASKROWS * NWORDS : cz_pad; }
  static constexpr size_t tail_bytes() {
    const size_t used = (size_t)2 * R * NW * 8 + sizeof(Slots) + (size_t)NSMALL * 4;
    const size_t slack = (size_t)2 * NT * 4 + 256;
    return used > slack ? used : slack;
  }
  static constexpr size_t bytes(int cz_pad) { return (size_t)(Lp + 64 + gap_floats(cz_pad)) * 4 + tail_bytes() + 64; }
  __device__ Smem(unsigned char* raw, int cz_pad) {
    bm = reinterpret_cast<uint32_t*>(raw);
    X = reinterpret_cast<float*>(raw) + gap_floats(cz_pad);
    dpart = reinterpret_cast<double*>(X + Lp + 64);
    sl = reinterpret_cast<Slots*>(dpart + 2 * R * NW);
    part = reinterpret_cast<float*>(sl + 1);
    scn = part + 2 * R * NW;
    estB = scn + 3 * R * NW;
    hy = estB + 2 * EST_TBL;
    float* wred_ = hy + (4 * NH_MAX + 1) * (R * NW + 1);
    lds_float* base = (lds_float*)wred_;
    asm volatile("" : "+v"(base));   // one VGPR base for the small arrays (they lie beyond the reach of a zero-based immediate)
    wred = base;
    wsum = base + 5 * NW;
    outv = wsum + NWSUM * NW;
    misc = outv + C_NCOLS;
  }
};

// Accumulators of a window's sums in the S4 view, as register pairs (see icpc_lean.hip, round 2)
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

// cross-wave prefix sums of the wave-row totals by ONE wave (see icpc_lean.hip)
template <int NW>
__device__ __forceinline__ double row_prefix_f64(const float* part, int lane, double* total) {
  const int jr = lane / NW, jw = lane - jr * NW;   // lane j <-> wave-row (r = jr, w = jw)
  const double pvv = (lane < R * NW) ? (double)part[4 * jw + jr] : 0.0;
  const double pinc = wave_incl_scan_sum_f64(pvv);
  if (total) *total = readlane_d(pinc, R * NW - 1);
  return pinc - pvv;
}
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
__device__ __forceinline__ f4 t_quad(f4 y, float p0, float hw, float lw) {
  const float p1 = p0 + y.x, p2 = p1 + y.y, p3 = p2 + y.z;
  const f2 h2 = splat(hw), lo2 = splat(lw);
  const f2 ta = h2 + (lo2 + mk2(p0, p1)), tb = h2 + (lo2 + mk2(p2, p3));
  return (f4){ta.x, ta.y, tb.x, tb.y};
}

// a thread's quad of a LINEAR LDS array at an arbitrary index (al = idx & 3, block-uniform): the widest reads the alignment
// allows (a 4-byte-aligned ds_read_b64 is 19x slower than an aligned one, two ds_read2_b32 of adjacent words 5x:
// tools/micro/lds_b64_test.hip, lds_s4shift_test.hip)
template <int AL>   // AL = the index's alignment class: 0 (multiple of 4), 2 (even), 1 (odd)
__device__ __forceinline__ f4 rdq_t(const float* X, int idx) {
  if constexpr (AL == 0) return *reinterpret_cast<const f4*>(&X[idx]);
  if constexpr (AL == 2) {
    const f2 a = *reinterpret_cast<const f2*>(&X[idx]), b = *reinterpret_cast<const f2*>(&X[idx + 2]);
    return (f4){a.x, a.y, b.x, b.y};
  }
  const float a = X[idx];
  const f2 m = *reinterpret_cast<const f2*>(&X[idx + 1]);
  const float d = X[idx + 3];
  return (f4){a, m.x, m.y, d};
}
__device__ __forceinline__ f4 rdq(const float* X, int idx, int al) {
  if (al == 0) return *reinterpret_cast<const f4*>(&X[idx]);
  if (al == 2) {
    const f2 a = *reinterpret_cast<const f2*>(&X[idx]), b = *reinterpret_cast<const f2*>(&X[idx + 2]);
    return (f4){a.x, a.y, b.x, b.y};
  }
  const float a = X[idx];
  const f2 m = *reinterpret_cast<const f2*>(&X[idx + 1]);
  const float d = X[idx + 3];
  return (f4){a, m.x, m.y, d};
}
// the four predicate bits of a thread's quad -> bits [4 (lane & 7), +4) of the word of its group of eight lanes; complete in
// the lanes with (lane & 7) == 7 (row_shr inside a row of 16 lanes, zeros shifted in)
__device__ __forceinline__ uint32_t s4_pack_word(uint32_t nib, int lane) {
  uint32_t v = nib << (4 * (lane & 7));
  // three fused OR steps in place (a lane whose source is out of the row ORs in zero: bound_ctrl), one asm statement
  asm("s_nop 1\n\t"
      "v_or_b32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
      "v_or_b32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
      "v_or_b32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0"
      : "+v"(v));
  return v;
}
// four words at once: the steps of the four chains interleave, so the two wait states between a step's write and the next step's DPP
// read of the same register are filled by the other chains (s4_pack_word alone: an s_nop in front of every step)
__device__ __forceinline__ void s4_pack4(uint32_t (&w)[4], int lane) {
  const int sh = 4 * (lane & 7);
  w[0] <<= sh; w[1] <<= sh; w[2] <<= sh; w[3] <<= sh;
#define LDSP_OR4(CTL) "v_or_b32_dpp %0, %0, %0 " CTL " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" "v_or_b32_dpp %1, %1, %1 " CTL " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
                     "v_or_b32_dpp %2, %2, %2 " CTL " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" "v_or_b32_dpp %3, %3, %3 " CTL " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#ifdef LDSP_WHATIF_S16
  asm("s_nop 1\n\tv_or_b32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(w[0]));
#else
  asm("s_nop 1\n\t" LDSP_OR4("row_shr:1") LDSP_OR4("row_shr:2") LDSP_OR4("row_shr:4") : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
#endif
#undef LDSP_OR4
}
// sample e of a quad inside the window [lo, hi] (both relative to the quad's first sample): ONE unsigned comparison, e - lo <= hi - lo —
// the two-sided form compiles to two compares joined by s_and_b64, and a select on a mask that SALU wrote stalls (tools/micro/valu_rate4.hip)
__device__ __forceinline__ bool in_win(int e, int lo, uint32_t span) { return (uint32_t)(e - lo) <= span; }
// the thread index as a value hipcc cannot connect to its other copies: index arithmetic (4 (tid + NT r) + e, window-edge tests ..) is
// recomputed in each phase — one VALU each — instead of living in a dozen registers from the first phase to the last
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// the four comparison bits of a quad as a nibble (bit e = sample e): compare into VCC, shift the bit in with an add-with-carry
// (n = 2 n + bit), most significant sample first — eight instructions, each select-free and on a VCC the instruction before it wrote
// (a NaN compares false, as in the C form)
__device__ __forceinline__ uint32_t nib_ge(f4 v, float t) {
  uint32_t n = 0;
  asm("v_cmp_ge_f32_e32 vcc, %1, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
      "v_cmp_ge_f32_e32 vcc, %2, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
      "v_cmp_ge_f32_e32 vcc, %3, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
      "v_cmp_ge_f32_e32 vcc, %4, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
      : "+v"(n) : "v"(v.w), "v"(v.z), "v"(v.y), "v"(v.x), "v"(t) : "vcc");
  return n;
}
__device__ __forceinline__ uint32_t nib_le(f4 v, float t) {
  uint32_t n = 0;
  asm("v_cmp_le_f32_e32 vcc, %1, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
      "v_cmp_le_f32_e32 vcc, %2, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
      "v_cmp_le_f32_e32 vcc, %3, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
      "v_cmp_le_f32_e32 vcc, %4, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
      : "+v"(n) : "v"(v.w), "v"(v.z), "v"(v.y), "v"(v.x), "v"(t) : "vcc");
  return n;
}

#ifndef LDSP_L3_SGC_VGPR
#define LDSP_L3_SGC_VGPR 0   // (1, measured: taps of the SG pass alone +-0, all loop constants -0.9 %: the copies and three spilled registers cost more) loop constants (filter taps, decay factors) live in vector registers: an instruction with a scalar source issues at 1.5x the cost of one without (tools/micro/valu_cost.hip)
#endif
// LDSP_LDS_WAIT_ALL(): every LDS read issued so far has returned.  Placed behind a GROUP of reads, in front of their first use: hipcc
// then drops its own progressive waits (lgkmcnt(3), (2), (1), (0) — one issue slot each, in a kernel that is bound by instruction issue)
// because the counter is known to be zero.  -DLDSP_L3_WAITALL=0 restores the progressive waits.
#ifndef LDSP_L3_WAITALL
#define LDSP_L3_WAITALL 0   // (measured: 19.42 M against 19.43 M waveforms/s — the progressive waits cost nothing that shows)
#endif
#if LDSP_L3_WAITALL
#define LDSP_LDS_WAIT_ALL() __builtin_amdgcn_s_waitcnt(0xc07f)
#else
#define LDSP_LDS_WAIT_ALL() ((void)0)
#endif
// LDSP_L3_WPS: minimum waves per SIMD the register allocation is bounded for (6: three 512-thread workgroups per CU, <= 80 VGPRs)
#ifndef LDSP_L3_WPS
#define LDSP_L3_WPS 6
#endif
#ifndef LDSP_L3_BFLY   // 1: the wave reductions of the sweeps' and the current windows' maxima as butterflies (wave_prims.hpp: LDSP_BFLY4): +1.5 %, same bits
#define LDSP_L3_BFLY 1
#endif
#ifndef LDSP_L3_BFLY_SUMS   // 1: also the small groups (tail sums, CUSP / ZAC maxima, raw extremes): no further gain measured (their wait states eat it), sums change their last bits
#define LDSP_L3_BFLY_SUMS 0
#endif
#ifndef LDSP_L3_TAB0   // 1: the cross-wave tables of the first exchange are computed by wave 0 only (see there)
#define LDSP_L3_TAB0 1
#endif
#ifndef LDSP_L3_RWPS   // the same for traces shorter than the tile (their bounds cost registers)
#define LDSP_L3_RWPS 6
#endif
// SEP: CUSP and ZAC have their own geometry (two passes of the closed-form stage); keeps a second copy of y in registers
// FULL: the trace fills the tile (L = 16 NT) — the production geometry, no bounds anywhere; !FULL: a shorter trace (any length)
template <int NT, int M, bool SEP, bool FULL>
__global__ void __launch_bounds__(NT, NT == 64 ? 5 : (FULL ? LDSP_L3_WPS : LDSP_L3_RWPS))   // (one-wave workgroups: 96 registers)
icpc_lean3_kernel(const float* __restrict__ wf, const IcpcDev* __restrict__ Pp, IcpcOutDev out, const float* __restrict__ ext_bl,
                  float ext_bl_scale, int skip_cz) {
  using SM = Smem<NT>;
  constexpr int NW = SM::NW, Lp = SM::Lp, NWORDS = SM::NWORDS;
  static_assert(R * NW <= 64, "wave-row partials must fit one wave");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const IcpcDev& P = *Pp;
#ifdef LDSP_ISA_STUDY   // static ISA of the bench configuration's path only (tools/l3_asm.sh -DLDSP_ISA_STUDY): never a product build
  constexpr bool STUDY = true;
#else
  constexpr bool STUDY = false;
#endif
  // Trace length: the tile (Lp = 16 NT samples) or less (more than half of it).  A lane whose quad of row 2 or 3
  // lies beyond L keeps a COPY OF ITS OWN ROW-0 QUAD there (rows 0 and 1 always lie inside the trace: the tile is the smallest that
  // holds it): real sample values, so the raw extremes are unchanged, and no load address is clamped; every filter of the chain is
  // causal up to its own output range, every output range is bounded by L below (nout, ng, the crossing tests), and the one
  // anti-causal recursion (CUSP / ZAC) runs on a difference signal that is set to zero beyond L.  (in_trace() below serves only the
  // rare re-read of the raw samples by the saturation count.)
  const int L = FULL ? Lp : P.L;
  auto in_trace = [&](int i4) { return FULL ? i4 : min(i4, L - 4); };   // offset of a quad, or of the last one
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Lf_max = max(P.cusp.Lf, P.zac.Lf);
  SM S(smem_raw, cz_pad_floats(Lf_max));
  const float* w = wf + (size_t)blockIdx.x * (size_t)L;
  auto wrow = [&](int r) { return 4 * (64 * wave + NT * r); };
  enum { WN_BL, WN_TAIL, WN_SGBL, WN_CUR0, WN_CURX, WN_CURI };
  const uint32_t cls_bl = P.rowcls[WN_BL][wave], cls_tail = P.rowcls[WN_TAIL][wave], cls_sgbl = P.rowcls[WN_SGBL][wave],
                 cls_cur0 = P.rowcls[WN_CUR0][wave], cls_curx = P.rowcls[WN_CURX][wave], cls_curi = P.rowcls[WN_CURI][wave];
  auto row_out = [&](uint32_t cls, int r) { return ((cls >> r) & 1u) != 0u; };
  auto row_in = [&](uint32_t cls, int r) { return ((cls >> (4 + r)) & 1u) != 0u; };

  // ------------------------------------------------------------------------------------------------ phase 0: load
  STAMP(0); DSTOP(0);
  f4 x[R];
  const uint16_t* w16 = reinterpret_cast<const uint16_t*>(wf) + (size_t)blockIdx.x * (size_t)L;
  auto wv = [&](int i) { return (!STUDY && P.in_u16) ? (float)w16[i] : w[i]; };
  // (!FULL: the tile is the smallest that holds the trace, so rows 0 and 1 lie inside it; a lane whose quad of row 2 or 3 lies beyond
  // the trace keeps a copy of its own row-0 quad there — real sample values: the raw extremes do not change.  L % 4 != 0: the one
  // quad that holds the end of the trace is read sample by sample — nothing behind the trace is touched — and keeps the row-0
  // copy in its last elements; the rows are then only 4-byte aligned, which global_load_dwordx4 / dwordx2 accept)
  auto load_last_quad = [&](f4 q, int i0, const float* w, const uint16_t* w16) __attribute__((always_inline)) {   // 1..3 samples of the trace from i0 on (one thread of the workgroup), the others as given
    auto wv = [&](int i) { return (!STUDY && P.in_u16) ? (float)w16[i] : w[i]; };
    const float a = wv(i0), b = (i0 + 1 < L) ? wv(i0 + 1) : q.y, c = (i0 + 2 < L) ? wv(i0 + 2) : q.z;
    return (f4){a, b, c, q.w};
  };
  auto load_raw = [&](f4 (&x)[R], const float* w, const uint16_t* w16) __attribute__((always_inline)) {   // (the row pointers as arguments: the second call hides them from the compiler, which would otherwise keep the first call's values alive — in scratch — instead of loading again)
    if ((!STUDY && P.in_u16)) {   // (block-uniform)
  #pragma unroll
      for (int r = 0; r < R; ++r) {
        if (!FULL && r >= 2) { x[r] = x[0]; if (4 * (tid + NT * r) >= L) continue; if (4 * (tid + NT * r) + 4 > L) { x[r] = load_last_quad(x[r], 4 * (tid + NT * r), w, w16); continue; } }
        const uint2 q = *reinterpret_cast<const uint2*>(w16 + 4 * (tid + NT * r));
        x[r] = (f4){(float)(q.x & 0xffffu), (float)(q.x >> 16), (float)(q.y & 0xffffu), (float)(q.y >> 16)};
      }
    } else {
  #pragma unroll
      for (int r = 0; r < R; ++r) {
        if (!FULL && r >= 2) { x[r] = x[0]; if (4 * (tid + NT * r) >= L) continue; if (4 * (tid + NT * r) + 4 > L) { x[r] = load_last_quad(x[r], 4 * (tid + NT * r), w, w16); continue; } }
        x[r] = *reinterpret_cast<const f4*>(w + 4 * (tid + NT * r));
      }
    }
  };
  load_raw(x, w, w16);
  const float pv_bl = wv(P.bl.from);      // pivot of the baseline sums: the window's first sample
  // LSQ basis tables of the two estimators -> LDS: both loads of a thread in flight together, and no loop where the tile has a thread per entry
  if constexpr (NT >= EST_TBL) {
    if (tid < EST_TBL) { const float a = P.sig_est.B[tid], b = P.int_est.B[tid]; S.estB[tid] = a; S.estB[EST_TBL + tid] = b; }
  } else {
    for (int i = tid; i < EST_TBL; i += NT) { const float a = P.sig_est.B[i], b = P.int_est.B[i]; S.estB[i] = a; S.estB[EST_TBL + i] = b; }
  }
  if (tid < (int)(sizeof(Slots) / 4)) {
    const int o = tid * 4;
    uint32_t init = 0;
    if (o >= (int)offsetof(Slots, imin) && o < (int)offsetof(Slots, imax)) init = 0x7fffffffu;
    else if (o >= (int)offsetof(Slots, imax)) init = 0xffffffffu;   // -1
    reinterpret_cast<uint32_t*>(S.sl)[tid] = init;
  }
  if (tid < 64) S.X[Lp + tid] = 0.f;

  // ================================================================== round 1: one pass over the raw trace in registers
  // Everything the chain needs before the first filter is linear in the baseline: with the pivot pv = the baseline window's first
  // sample (known at load time), x' = raw - pv and delta = blmean - pv,
  //   shift_waveform      x[i]  = x'[i] - delta                                                     (dsp_icpc.jl:105)
  //   InvCRFilter         y[i]  = x[i] + c cs[i],      cs[i] = cs'[i] - delta (i+1)                  (dsp_icpc.jl:119-120)
  //   its prefix sum      T[i]  = E1'[i] + c E2'[i] - delta (i + c i (i+1) / 2)                      (what the trapezoids read)
  // where cs' = cumsum(x') (inclusive), E1' its exclusive form and E2'[i] = sum_{m<i} cs'[m].  So the baseline sums, the raw
  // extremes and the single and double prefix sums of x' inside every wave-row (256 samples) come out of ONE group of DPP scans
  // and ONE exchange of per-wave partials; one wave turns the wave-row totals into the state entering each wave-row (in double:
  // C1 = cs' before the row, C2 = E2' at its first sample), and y and T follow directly — in place of three exchanges (baseline;
  // cumsum for the pole-zero; prefix sum of y) with a serial section each.
  float ex1[R], ex2[R];   // sums of x' / of cs' (row-local) over the samples of the wave-row before this lane's quad
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
    float t[R], q2[R], i1[R], i2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      x[r].xy -= pv; x[r].zw -= pv;
      const float c0 = x[r].x, c1 = c0 + x[r].y, c2 = c1 + x[r].z, c3 = c2 + x[r].w;
      t[r] = c3; i1[r] = c3;
      q2[r] = (c0 + c1) + (c2 + c3);   // the quad's own part of the double sum
      if (row_out(cls_bl, r)) continue;
      f2 d0 = x[r].xy, d1 = x[r].zw;
      if (!row_in(cls_bl, r)) {
        const int i0 = 4 * (opaque(tid) + NT * r), lo = P.bl.from - i0, hi = P.bl.until - i0;   // in-window e in [lo, hi]
        d0.x = in_win(0, lo, (uint32_t)(hi - lo)) ? d0.x : 0.f; d0.y = in_win(1, lo, (uint32_t)(hi - lo)) ? d0.y : 0.f;
        d1.x = in_win(2, lo, (uint32_t)(hi - lo)) ? d1.x : 0.f; d1.y = in_win(3, lo, (uint32_t)(hi - lo)) ? d1.y : 0.f;
      }
      wacc_quad(a, d0, d1, r);
    }
    float s1, s2, sx;
    wacc_lane<NT>(a, tid, (float)P.bl.ic, &s1, &s2, &sx);
#if LDSP_L3_BFLY_SUMS
    // (the three sums in the order config 2's kernel takes them: blmean bit for bit; the extremes as a butterfly: lane 31: max, lane 63: -min)
    LDSP_DPP_GROUP3("v_add_f32_dpp", s1, "v_add_f32_dpp", s2, "v_add_f32_dpp", sx);
    rmin = -rmin;
    LDSP_BFLY2("v_max_f32", "v_max_f32_dpp", rmax, rmin);
#else
    LDSP_DPP_GROUP5("v_add_f32_dpp", s1, "v_add_f32_dpp", s2, "v_add_f32_dpp", sx, "v_max_f32_dpp", rmax, "v_min_f32_dpp", rmin);
#endif
#ifdef LDSP_WHATIF_S16
    LDSP_DPP_GROUP1("v_add_f32_dpp", i1[0]);
#else
    LDSP_DPP_GROUP4("v_add_f32_dpp", i1[0], "v_add_f32_dpp", i1[1], "v_add_f32_dpp", i1[2], "v_add_f32_dpp", i1[3]);
#endif
#pragma unroll
    for (int r = 0; r < R; ++r) { ex1[r] = i1[r] - t[r]; i2[r] = fmaf(4.f, ex1[r], q2[r]); ex2[r] = i2[r]; }
#ifdef LDSP_WHATIF_S16
    LDSP_DPP_GROUP1("v_add_f32_dpp", i2[0]);
#else
    LDSP_DPP_GROUP4("v_add_f32_dpp", i2[0], "v_add_f32_dpp", i2[1], "v_add_f32_dpp", i2[2], "v_add_f32_dpp", i2[3]);
#endif
#pragma unroll
    for (int r = 0; r < R; ++r) ex2[r] = i2[r] - ex2[r];
    if (lane == 63) {
      S.wred[0 * NW + wave] = s1; S.wred[1 * NW + wave] = s2; S.wred[2 * NW + wave] = sx;
#if LDSP_L3_BFLY_SUMS
      S.wred[4 * NW + wave] = -rmax;   // (lane 63 of the butterfly: -min)
#else
      S.wred[3 * NW + wave] = rmax; S.wred[4 * NW + wave] = rmin;
#endif
      *reinterpret_cast<f4*>(&S.part[4 * wave]) = (f4){i1[0], i1[1], i1[2], i1[3]};            // [wave][r]
      *reinterpret_cast<f4*>(&S.part[R * NW + 4 * wave]) = (f4){i2[0], i2[1], i2[2], i2[3]};
    }
#if LDSP_L3_BFLY_SUMS
    if (lane == 31) S.wred[3 * NW + wave] = rmax;   // (lane 31 of the butterfly: max)
#endif
  }
  STAMP(1); DSTOP(1);
  LDSP_BAR_MAIN();
  // State entering each wave-row, lane j <-> wave-row j in time order (r = j / NW, w = j % NW).  LDSP_L3_TAB0 = 0: EVERY wave scans
  // the R*NW totals for itself (~120 instructions) and takes its own rows' entries by v_readlane — no wave waits for another (right
  // while the kernel waited on latency).  LDSP_L3_TAB0 = 1 (default since the VALU became the busy unit): wave 0
  // scans them for everybody while the other waves go on with the tail logarithms, which need the baseline mean only; one more barrier
  // in front of the y / T loop, whose rows then come from the table in LDS (which the CUSP / ZAC stage wants there anyway).
  float tab_hi = 0.f, tab_lo = 0.f, tab_b = 0.f;   // lane j: T at wave-row j's first sample (without the delta terms) as hi + lo; c * (sum of x' before it)
#ifdef LDSP_WHATIF_S16
  if (false) {
#else
  if (!LDSP_L3_TAB0 || wave == 0) {
#endif
    const int jr = lane / NW, jw = lane - jr * NW;
    const bool in = lane < R * NW;
    // C1_j = sum_{j'<j} S_j',  C2_j = sum_{j'<j} (Q_j' + 256 C1_j') = sum_{j'<j} Q_j' + 256 ((j-1) sum_{j'<j} S_j' - sum_{j'<j} j' S_j'):
    // three INDEPENDENT scans (their DPP steps interleave) instead of two dependent ones
    const double sv = in ? (double)S.part[4 * jw + jr] : 0.0, qv = in ? (double)S.part[R * NW + 4 * jw + jr] : 0.0, jv = sv * (double)lane;
    const double c1i = wave_incl_scan_sum_f64(sv), qi = wave_incl_scan_sum_f64(qv), ji = wave_incl_scan_sum_f64(jv);
    const double c1x = c1i - sv;
    const double c2x = (qi - qv) + 256.0 * ((double)(lane - 1) * c1x - (ji - jv));
    const double c2i = qi + 256.0 * ((double)lane * c1i - ji);   // (at lane R*NW - 1: the state after the last wave-row)
    const double A = fma(P.pz_c64, c2x, c1x);   // T at the row's first sample, but for the delta terms
    tab_hi = (float)A; tab_lo = (float)(A - (double)tab_hi); tab_b = (float)(P.pz_c64 * c1x);
    if (wave == 0) {
      if (in) { S.scn[4 * jw + jr] = tab_hi; S.scn[R * NW + 4 * jw + jr] = tab_lo; S.scn[2 * R * NW + 4 * jw + jr] = tab_b; }
      const double At = fma(P.pz_c64, readlane_d(c2i, R * NW - 1), readlane_d(c1i, R * NW - 1));   // ... at sample L
      if (lane == 0) { const float ah = (float)At; S.misc[16] = ah; S.misc[17] = (float)(At - (double)ah); }
    }
  }
  float blmean, raw_max, raw_min, delta;
  {
    const float s = fold_partials<NW>(S.wred, 0.f, [](float a, float b) { return a + b; });
    float mx, mn;
    if constexpr (NW == 8) {   // eight partials: four three-operand instructions in ONE statement (a chain of eight asm statements pays a wait state after each)
      const auto a = *(const lds_f4*)(S.wred + 3 * NW), b = *(const lds_f4*)(S.wred + 3 * NW + 4);
      const auto c = *(const lds_f4*)(S.wred + 4 * NW), d = *(const lds_f4*)(S.wred + 4 * NW + 4);
      asm("v_max3_f32 %0, %2, %3, %4\n\tv_min3_f32 %1, %10, %11, %12\n\tv_max3_f32 %0, %0, %5, %6\n\tv_min3_f32 %1, %1, %13, %14\n\t"
          "v_max3_f32 %0, %0, %7, %8\n\tv_min3_f32 %1, %1, %15, %16\n\tv_max_f32 %0, %0, %9\n\tv_min_f32 %1, %1, %17"
          : "=&v"(mx), "=&v"(mn)
          : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w), "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));
    } else {
      mx = fold_partials<NW>(S.wred + 3 * NW, -INFINITY, [](float a, float b) { return vmax(a, b); });
      mn = fold_partials<NW>(S.wred + 4 * NW, INFINITY, [](float a, float b) { return vmin(a, b); });
    }
    delta = s * (float)P.bl.inv_n;
    blmean = fmaf(s, (float)P.bl.inv_n, pv_bl);   // (the statement of pz_trap_lean_kernel, icpc_lean.hip: config 2 reports the same bits)
    if (!STUDY && ext_bl) { blmean = ext_bl[blockIdx.x] * ext_bl_scale; delta = blmean - pv_bl; }   // windowed traces of dsp_icpc_compressed (dsp_icpc.jl:353)
    raw_max = mx; raw_min = mn;
  }
  const float e_max = raw_max - blmean;
  // block-uniform scalars that are needed again much later wait in LDS, not in VGPRs
  enum { MS_BLMEAN = 18, MS_RAWMAX, MS_RAWMIN, MS_DELTA, MS_PVTL, MS_THRI, MS_THR5, MS_EMAX };
  if (tid == 0) { S.misc[MS_BLMEAN] = blmean; S.misc[MS_RAWMAX] = raw_max; S.misc[MS_RAWMIN] = raw_min; S.misc[MS_DELTA] = delta; S.misc[MS_EMAX] = e_max; }
  STAMP(2); DSTOP(2);

  // saturation (src/saturation.jl:28-65): only a trace whose extremes reach a rail can have saturated samples
  {
    int n_low = 0, n_high = 0, cons_low = 0, cons_high = 0;
    if (!STUDY && (raw_min <= P.sat_low || raw_max >= P.sat_high)) {   // block-uniform, rare
      const float lo_ = P.sat_low, hi_ = P.sat_high;   // exact equality on the RAW samples: read again (the registers hold x')
#pragma unroll
      for (int r = 0; r < R; ++r) {
        f4 v;
        const int i0 = 4 * (tid + NT * r);
        if ((!STUDY && P.in_u16)) {
          const uint2 q = *reinterpret_cast<const uint2*>(w16 + in_trace(i0));
          v = (f4){(float)(q.x & 0xffffu), (float)(q.x >> 16), (float)(q.y & 0xffffu), (float)(q.y >> 16)};
        } else {
          v = *reinterpret_cast<const f4*>(w + in_trace(i0));
        }
        if (!FULL && i0 + 4 > L) {   // beyond the trace: equal to neither rail (the quad that holds the end: sample by sample)
          v = (f4){NAN, NAN, NAN, NAN};
          if (i0 < L) v = load_last_quad(v, i0, w, w16);
        }
        n_low += (v.x == lo_) + (v.y == lo_) + (v.z == lo_) + (v.w == lo_);
        n_high += (v.x == hi_) + (v.y == hi_) + (v.z == hi_) + (v.w == hi_);
        *reinterpret_cast<f4*>(&S.X[4 * (tid + NT * r)]) = v;
      }
      n_low = wave_sum_all_i(n_low); n_high = wave_sum_all_i(n_high);
      if (lane == 0) { atomicAdd(&S.sl->imax[0], n_low); atomicAdd(&S.sl->imax[1], n_high); }   // the slots start at -1
      __syncthreads();
      for (int m = 0; m < SP; ++m) {
        const float v = S.X[tid + NT * m];
        ballot_store(v == lo_, S.bm + M_FB * NWORDS, (NT >> 5) * m + 2 * wave);
        ballot_store(v == hi_, S.bm + (M_FB + 1) * NWORDS, (NT >> 5) * m + 2 * wave);
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
  // ---- y (registers) and T (-> X) of every quad; tailstats sums of log(x) on the way (src/tailstats.jl:22-72)
  {
    // The sums run in log2 units (v_log_f32 alone: `__logf` expands to a dozen instructions of denormal scaling and an extended-precision
    // product with ln 2 that a float sum about a pivot cannot use) and are converted once, by the finishing lane.  Pivot: log2 of the
    // window's first sample x_p; a sample of an edge row that lies outside the window is replaced by x_p itself, whose term
    // log2(x_p) - pivot is exactly zero and which passes the sign test whenever the window does.  A non-positive sample makes the sums
    // NaN / -inf, and tail_bad discards them.
    const float x_p = wv(P.tail.from) - blmean;
    const float pv_tl = __builtin_amdgcn_logf(x_p);
    if (tid == 0) S.misc[MS_PVTL] = pv_tl;
    // ---- tailstats sums of log(x), x = x' - delta (src/tailstats.jl:22-72), reduced at once (a loop of its own, before y takes
    // the place of x': the two loops' temporaries do not add up)
    {
      float l1, l2, lx, tmin = INFINITY;
      WAcc a = {splat(0.f), splat(0.f), splat(0.f), splat(0.f)};
      const f2 pvl = splat(pv_tl);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (row_out(cls_tail, r)) continue;
        f2 v0 = mk2(x[r].x - delta, x[r].y - delta), v1 = mk2(x[r].z - delta, x[r].w - delta);
        if (!row_in(cls_tail, r)) {
          const int i0 = 4 * (opaque(tid) + NT * r);
          const int lo = P.tail.from - i0;
          const uint32_t span = (uint32_t)(P.tail.until - P.tail.from);
          v0.x = in_win(0, lo, span) ? v0.x : x_p; v0.y = in_win(1, lo, span) ? v0.y : x_p;
          v1.x = in_win(2, lo, span) ? v1.x : x_p; v1.y = in_win(3, lo, span) ? v1.y : x_p;
        }
        tmin = vmin3(tmin, v0.x, v0.y); tmin = vmin3(tmin, v1.x, v1.y);
        const f2 d0 = mk2(__builtin_amdgcn_logf(v0.x), __builtin_amdgcn_logf(v0.y)) - pvl;
        const f2 d1 = mk2(__builtin_amdgcn_logf(v1.x), __builtin_amdgcn_logf(v1.y)) - pvl;
        wacc_quad(a, d0, d1, r);
      }
      wacc_lane<NT>(a, tid, (float)P.tail.ic, &l1, &l2, &lx);
#if LDSP_L3_BFLY_SUMS
      float z0 = 0.f;
      LDSP_BFLY4("v_add_f32", "v_add_f32_dpp", l1, l2, lx, z0);   // totals in lanes 15 / 47 / 31 of l1
      LDSP_DPP_GROUP1("v_min_f32_dpp", tmin);
      if ((lane & 15) == 15 && lane != 63) S.wsum[(W_TAIL + (lane == 15 ? 0 : lane == 47 ? 1 : 2)) * NW + wave] = l1;
      if (lane == 63 && tmin <= 0.f) S.sl->isum[IS_TAILBAD] = 1;   // any wave may set it (same value)
#else
      LDSP_DPP_GROUP4("v_add_f32_dpp", l1, "v_add_f32_dpp", l2, "v_add_f32_dpp", lx, "v_min_f32_dpp", tmin);
      if (lane == 63) {
        S.wsum[(W_TAIL + 0) * NW + wave] = l1; S.wsum[(W_TAIL + 1) * NW + wave] = l2; S.wsum[(W_TAIL + 2) * NW + wave] = lx;
        if (tmin <= 0.f) S.sl->isum[IS_TAILBAD] = 1;   // any wave may set it (same value)
      }
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
#if LDSP_L3_TAB0 && !defined(LDSP_WHATIF_S16)
    LDSP_BAR_MAIN();   // wave 0's tables are in LDS
#endif
    // ---- y (registers) and T (-> X) of every quad
    const float c = P.pz_c, cd = c * delta, hc = 0.5f * c;
    const float n0 = (float)(4 * lane);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i0 = 4 * (opaque(tid) + NT * r);
      const float fi = (float)i0;
#if LDSP_L3_TAB0
      const float Hr = S.scn[4 * wave + r], Lr = S.scn[R * NW + 4 * wave + r], Br = S.scn[2 * R * NW + 4 * wave + r];   // the row's entries (wave-uniform addresses: broadcast reads)
#else
      const float Hr = readlane_f(tab_hi, r * NW + wave), Lr = readlane_f(tab_lo, r * NW + wave), Br = readlane_f(tab_b, r * NW + wave);   // the row's entries
#endif
      const float c0 = x[r].x, c1 = c0 + x[r].y, c2 = c1 + x[r].z, c3 = c2 + x[r].w;   // running sums of x' inside the quad
      const f4 xs = (f4){x[r].x - delta, x[r].y - delta, x[r].z - delta, x[r].w - delta};   // shift_waveform
      // y[i] = x[i] + c (C1 + cs_l) - c delta (i+1),  c C1 = B,  cs_l(e) = ex1 + c_e
      {
        const float ky = fmaf(c, ex1[r], Br) - cd * (fi + 1.f);
        x[r].x = xs.x + fmaf(c, c0, ky);
        x[r].y = xs.y + fmaf(c, c1, ky - cd);
        x[r].z = xs.z + fmaf(c, c2, ky - 2.f * cd);
        x[r].w = xs.w + fmaf(c, c3, ky - 3.f * cd);
      }
      // T at the quad's first sample = hi + (lo + inner),  inner = ex1 + n B + c ex2 - delta (i + c i (i+1) / 2)  (n = 4 lane,
      // i = i0); at its other samples inner grows by the y in front of them: one rounding at the magnitude of T
      {
        const float in0 = fmaf(-delta, fi * fmaf(hc, fi + 1.f, 1.f), fmaf(c, ex2[r], fmaf(n0, Br, ex1[r])));
        const float in1 = in0 + x[r].x, in2 = in1 + x[r].y, in3 = in2 + x[r].z;
        const float H = Hr, Lo = Lr;
        *reinterpret_cast<f4*>(&S.X[i0]) = (f4){H + (Lo + in0), H + (Lo + in1), H + (Lo + in2), H + (Lo + in3)};
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tid == 0) S.X[Lp] = S.misc[16] + (S.misc[17] - delta * ((float)Lp * fmaf(hc, (float)(Lp + 1), 1.f)));   // T[Lp] (a full tile's T[L]; a shorter trace has its T[L] at X[L])
  }
  auto& y = x;
  // get_threshold at 10 / 50 / 80 / 90 / 99 % of the pre-PZ maximum (dsp_icpc.jl:132-136): first quad of this wave reaching
  // each threshold (ballots on the quad maxima); confirmed when y is in LDS.
  if (e_max > 0.f) {
    const float thr_tx[5] = {e_max * 0.1f, e_max * 0.5f, e_max * 0.8f, e_max * 0.9f, e_max * 0.99f};
    int qfirst[5] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    bool all_found = false;   // wave-uniform
#pragma unroll
    for (int r = 0; r < R; ++r) {
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
      all_found = qfirst[4] != 0x7fffffff;
    }
    if (lane < 5) {
      const int qq = lane == 0 ? qfirst[0] : lane == 1 ? qfirst[1] : lane == 2 ? qfirst[2] : lane == 3 ? qfirst[3] : qfirst[4];
      if (qq != 0x7fffffff) atomicMin(&S.sl->imin[IM_TX0 + lane], qq);
    }
  }
  if (tid == 0) S.misc[15] = y[0].x;   // y[0] (the CUSP / ZAC stage: Dp = y - y[0] + eps T)
  {   // pivot of signalstats(pole-zero corrected tail): the window's first sample, from its owner's registers
    const int tq = P.tail.from >> 2, tr = tq / NT, te = P.tail.from & 3;   // (block-uniform)
    // (row by row with a static index: a conditional between the array's own elements is an lvalue — hipcc selects the ADDRESS, and an
    // array whose address is selected stays in scratch for the whole kernel once it is written a second time, as the SEP form does)
    if constexpr (SEP) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (r == tr && tid == tq - NT * tr) S.misc[14] = (te == 0) ? y[r].x : (te == 1) ? y[r].y : (te == 2) ? y[r].z : y[r].w;
    } else if (tid == tq - NT * tr) {   // (the form the single-pass kernel was tuned with: any other costs it three registers to scratch)
      const f4 v = (tr == 0) ? y[0] : (tr == 1) ? y[1] : (tr == 2) ? y[2] : y[3];
      S.misc[14] = (te == 0) ? v.x : (te == 1) ? v.y : (te == 2) ? v.z : v.w;
    }
  }
  // the first NH quads of every wave-row: halo of the previous wave-row's last lanes (sweep A's short leg, Savitzky-Golay)
  // (M <= 13: the whole window of a quad's four outputs — M + 3 samples — waits in registers; M = 25, optimised Savitzky-Golay windows
  // of up to 350 ns at 16 ns: the main filter streams through the halo quads, two at a time, and only the two fixed short filters
  // — at most MS = 13 taps, the host admits nothing else — use a register window)
  constexpr int NH = (M + 3 - 4 + 3) / 4;
  static_assert(NH <= NH_MAX, "halo table");
  constexpr int MS = M <= 13 ? M : 13, NHS = (MS + 3 - 4 + 3) / 4;
  float* hyl = S.hy + 4 * NH * (R * NW + 1);
  if (lane < NH) {
#pragma unroll
    for (int r = 0; r < R; ++r) *reinterpret_cast<f4*>(&S.hy[4 * (NH * (r * NW + wave) + lane)]) = y[r];
  }
  if (lane == 63) {
#pragma unroll
    for (int r = 0; r < R; ++r) hyl[r * NW + wave + 1] = y[r].w;
  }
  if (tid < 4 * NH) S.hy[4 * NH * R * NW + tid] = 0.f;
  STAMP(3); DSTOP(3); LDSP_PROBE();
  LDSP_BAR_MAIN();   // X = T; halo table, threshold candidates, pivot
  // quad j + 1, j + 2, .. of the thread's row (the next lanes' registers; for the last lanes the next wave-row's first quads)
  // (the table quad is read into the destination registers — a wave-uniform address, every lane receives it — and the DPP moves
  // overwrite every lane that has a successor: lane 63 keeps the table's value; no select)
  auto halo_next = [&](f4 q, int r, int level) {
#ifdef LDSP_WHATIF_S16
    if (r > 0 || level > 0) return q;
#endif
    f4 h = *reinterpret_cast<const f4*>(&S.hy[4 * (NH * (r * NW + wave + 1) + level)]);
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %4 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %5 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %2, %6 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %3, %7 wave_shl:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(h.x), "+v"(h.y), "+v"(h.z), "+v"(h.w) : "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w));
    return h;
  };
  auto halo_next2 = [&](f4 q, int r) {   // the first two samples of the next quad (sweep A's short leg)
#ifdef LDSP_WHATIF_S16
    if (r > 0) return (f2)q.xy;
#endif
    f2 h = *reinterpret_cast<const f2*>(&S.hy[4 * (NH * (r * NW + wave + 1))]);
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %3 wave_shl:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(h.x), "+v"(h.y) : "v"(q.x), "v"(q.y));
    return h;
  };

  // ================================================================== round 2: sweeps over T, Savitzky-Golay from registers
  float mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY, mn0 = INFINITY, mn2 = INFINITY;
  float bo_v = -INFINITY; int bo_i = 0x7fffffff;
  {
    // ---- sweep A, S4 view: the two threshold masks of the t0 trapezoid (get_t0, dsp_routines.jl:9-25; inverted: dsp_icpc.jl:207).
    // o'[k] = (T[k+flen] - T[k+n1+g]) * (inv2/inv1) - (first leg's sum).  A first leg of <= 3 samples (get_t0's 40 ns) is summed
    // from y itself — the thread's quad and the first two samples of the next lane's: on the tail T is 1e7..1e8 and a difference
    // of two of its float values is good to 1..8 counts, the size of the threshold the INVERTED trace is tested against there.
    {
      TrapDev t0 = P.t0;
      if (STUDY) { t0.n1 = 2; t0.g = 6; t0.flen = 133; }
      const int nout = L - t0.flen + 1;
      const float thr0 = P.t0_thr * t0.navg, mthr0 = -thr0;
      const int s_b = t0.n1 + t0.g, s_c = t0.flen;
      const bool short1 = t0.n1 <= 3;
      uint32_t wp[R], wn[R];
      // (get_t0's own geometry — second leg at a multiple of four samples, end at an odd one — has a copy of the loop without
      // alignment tests; other geometries decide per quad read)
      auto rows = [&](auto fast_tag) {
      constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        wp[r] = 0u; wn[r] = 0u;
        if (4 * NT * r >= nout) continue;   // row beyond the output range
        const int i0 = 4 * (opaque(tid) + NT * r);
        const int ic = min(i0, (nout - 1) & ~3);   // quads beyond the output range read the last one's addresses (masked below)
        const f4 tb4 = FAST ? rdq_t<0>(S.X, ic + s_b) : rdq(S.X, ic + s_b, s_b & 3), tc4 = FAST ? rdq_t<1>(S.X, ic + s_c) : rdq(S.X, ic + s_c, s_c & 3);
        f4 sl;
        if (short1) {
          const f2 hn = halo_next2(y[r], r);
          sl = y[r];
          if (t0.n1 >= 2) sl += (f4){y[r].y, y[r].z, y[r].w, hn.x};
          if (t0.n1 >= 3) sl += (f4){y[r].z, y[r].w, hn.x, hn.y};
        } else {
          sl = rdq(S.X, ic + t0.n1, t0.n1 & 3) - *reinterpret_cast<const f4*>(&S.X[ic]);
        }
        f4 o;
        o.xy = fma2(tc4.xy - tb4.xy, splat(t0.rr), -sl.xy);
        o.zw = fma2(tc4.zw - tb4.zw, splat(t0.rr), -sl.zw);
        if (4 * NT * (r + 1) > nout) {   // the row that holds the end of the output range: NaN fails both comparisons
          o.x = (i0 + 0 < nout) ? o.x : NAN; o.y = (i0 + 1 < nout) ? o.y : NAN;
          o.z = (i0 + 2 < nout) ? o.z : NAN; o.w = (i0 + 3 < nout) ? o.w : NAN;
        }
        wp[r] = nib_ge(o, thr0);
        wn[r] = nib_le(o, mthr0);   // -trap >= thr
      }
      };
      if ((s_b & 3) == 0 && (s_c & 1) == 1) rows(std::true_type{});
      else rows(std::false_type{});
      s4_pack4(wp, lane); s4_pack4(wn, lane);
      static_assert(M_T0INV == M_T0 + 1, "mask order");
      if ((lane & 7) == 7) {   // word of (row r, wave, lane group): samples 4 (64 wave + 8 (lane >> 3) + NT r) ..
        uint32_t* bw = S.bm + M_T0 * NWORDS + 8 * wave + (lane >> 3);
#pragma unroll
        for (int r = 0; r < R; ++r) { bw[(NT / 8) * r] = wp[r]; bw[NWORDS + (NT / 8) * r] = wn[r]; }
      }
    }
    STAMP(4); DSTOP(4); LDSP_PROBE(); LDSP_PROBE();
    // ---- sweep B, LS view: extrema of the three fixed trapezoids, arg-max of the optimised one (dsp_icpc.jl:147-164, 202-204);
    // two rows per step (one ds_read2st64_b32 per shift), packed arithmetic
    const float* tb = &S.X[tid];
    auto rd2 = [&](const float* p, int m) { return mk2(p[NT * m], p[NT * (m + 1)]); };
    // (two trapezoids per pass over the rows — seven address registers and two trapezoids' reads in flight at a time — and T[k]
    // read once more: the register budget of three workgroups per CU)
    // Row pairs wholly inside both output ranges of a pass (most: one scalar test per pair, straight-line code, no selects, no joins);
    // the one or two pairs that hold the end of a range run a rolled loop with bounds tests.
    {
      const TrapDev f0 = P.fixed[0], f1 = P.fixed[1];
      const float *f0a = tb + f0.n1, *f0b = tb + f0.n1 + f0.g, *f0c = tb + f0.flen;
      const float *f1a = tb + f1.n1, *f1b = tb + f1.n1 + f1.g, *f1c = tb + f1.flen;
      f2 rr0 = splat(f0.rr), rr1 = splat(f1.rr);
      if (LDSP_L3_SGC_VGPR) { pin(rr0); pin(rr1); }
      const int n0 = L - f0.flen + 1, n1 = L - f1.flen + 1;
      const int pfull = min(n0, n1) / (2 * NT), pend = (max(n0, n1) + 2 * NT - 1) / (2 * NT);   // (block-uniform)
#pragma unroll
      for (int m = 0; m < SP; m += 2) {
        if (m / 2 >= pfull) break;
        const f2 Tk = rd2(tb, m);
        const f2 a0 = rd2(f0a, m), b0 = rd2(f0b, m), c0_ = rd2(f0c, m);
        const f2 a1 = rd2(f1a, m), b1 = rd2(f1b, m), c1_ = rd2(f1c, m);
        LDSP_LDS_WAIT_ALL();
        const f2 o0 = fma2(c0_ - b0, rr0, Tk - a0), o1 = fma2(c1_ - b1, rr1, Tk - a1);
        mx0 = vmax3(mx0, o0.x, o0.y); mn0 = vmin3(mn0, o0.x, o0.y);
        mx1 = vmax3(mx1, o1.x, o1.y);
        __builtin_amdgcn_sched_barrier(0);
      }
      for (int pm = pfull; pm < min(pend, SP / 2); ++pm) {
        const int off = 2 * NT * pm;
        const int k0 = opaque(tid) + off, k1 = k0 + NT;
        const f2 Tk = rd2(tb + off, 0);
        const f2 a0 = rd2(f0a + off, 0), b0 = rd2(f0b + off, 0), c0_ = rd2(f0c + off, 0);
        const f2 a1 = rd2(f1a + off, 0), b1 = rd2(f1b + off, 0), c1_ = rd2(f1c + off, 0);
        const f2 o0 = fma2(c0_ - b0, rr0, Tk - a0), o1 = fma2(c1_ - b1, rr1, Tk - a1);
        mx0 = vmax3(mx0, k0 < n0 ? o0.x : -INFINITY, k1 < n0 ? o0.y : -INFINITY);
        mn0 = vmin3(mn0, k0 < n0 ? o0.x : INFINITY, k1 < n0 ? o0.y : INFINITY);
        mx1 = vmax3(mx1, k0 < n1 ? o1.x : -INFINITY, k1 < n1 ? o1.y : -INFINITY);
      }
      mx0 *= f0.inv1; mn0 *= f0.inv1; mx1 *= f1.inv1;
    }
    {
      const TrapDev f2_ = P.fixed[2], fo = P.opt;
      const float *f2a = tb + f2_.n1, *f2b = tb + f2_.n1 + f2_.g, *f2c = tb + f2_.flen;
      const float *foa = tb + fo.n1, *fob = tb + fo.n1 + fo.g, *foc = tb + fo.flen;
      f2 rr2 = splat(f2_.rr), rro = splat(fo.rr);
      if (LDSP_L3_SGC_VGPR) { pin(rr2); pin(rro); }
      const int n2 = L - f2_.flen + 1, no = L - fo.flen + 1;
      const int pfull = min(n2, no) / (2 * NT), pend = (max(n2, no) + 2 * NT - 1) / (2 * NT);
      // the optimised trapezoid's arg-max is tracked as the ROW of the sample (an inline constant in the straight-line part); the
      // index follows at the end
#pragma unroll
      for (int m = 0; m < SP; m += 2) {
        if (m / 2 >= pfull) break;
        const f2 Tk = rd2(tb, m);
        const f2 a2 = rd2(f2a, m), b2 = rd2(f2b, m), c2_ = rd2(f2c, m);
        const f2 ao = rd2(foa, m), bo = rd2(fob, m), co = rd2(foc, m);
        LDSP_LDS_WAIT_ALL();
        const f2 o2 = fma2(c2_ - b2, rr2, Tk - a2);
        const f2 oo = fma2(co - bo, rro, Tk - ao);
        mx2 = vmax3(mx2, o2.x, o2.y); mn2 = vmin3(mn2, o2.x, o2.y);
        if (oo.x > bo_v) { bo_v = oo.x; bo_i = m; }
        if (oo.y > bo_v) { bo_v = oo.y; bo_i = m + 1; }
        __builtin_amdgcn_sched_barrier(0);
      }
      for (int pm = pfull; pm < min(pend, SP / 2); ++pm) {
        const int off = 2 * NT * pm;
        const int k0 = opaque(tid) + off, k1 = k0 + NT;
        const f2 Tk = rd2(tb + off, 0);
        const f2 a2 = rd2(f2a + off, 0), b2 = rd2(f2b + off, 0), c2_ = rd2(f2c + off, 0);
        const f2 ao = rd2(foa + off, 0), bo = rd2(fob + off, 0), co = rd2(foc + off, 0);
        const f2 o2 = fma2(c2_ - b2, rr2, Tk - a2);
        f2 oo = fma2(co - bo, rro, Tk - ao);
        mx2 = vmax3(mx2, k0 < n2 ? o2.x : -INFINITY, k1 < n2 ? o2.y : -INFINITY);
        mn2 = vmin3(mn2, k0 < n2 ? o2.x : INFINITY, k1 < n2 ? o2.y : INFINITY);
        oo.x = k0 < no ? oo.x : -INFINITY; oo.y = k1 < no ? oo.y : -INFINITY;
        if (oo.x > bo_v) { bo_v = oo.x; bo_i = 2 * pm; }
        if (oo.y > bo_v) { bo_v = oo.y; bo_i = 2 * pm + 1; }
      }
      bo_i = (bo_i != 0x7fffffff) ? opaque(tid) + NT * bo_i : bo_i;
      mx2 *= f2_.inv1; mn2 *= f2_.inv1; bo_v *= fo.inv1;
    }
  }
  STAMP(5); DSTOP(5); LDSP_PROBE();
  // ---- signalstats of the pole-zero corrected tail (dsp_icpc.jl:122) from the registers, pivot = its first sample
  float t1, t2, tx;
  {
    WAcc a = {splat(0.f), splat(0.f), splat(0.f), splat(0.f)};
    const f2 pvz = splat(S.misc[14]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (row_out(cls_tail, r)) continue;
      const int i0 = 4 * (opaque(tid) + NT * r);
      f2 d0 = y[r].xy - pvz, d1 = y[r].zw - pvz;
      if (!row_in(cls_tail, r)) {
        const int lo = P.tail.from - i0, hi = P.tail.until - i0;
        d0.x = in_win(0, lo, (uint32_t)(hi - lo)) ? d0.x : 0.f; d0.y = in_win(1, lo, (uint32_t)(hi - lo)) ? d0.y : 0.f;
        d1.x = in_win(2, lo, (uint32_t)(hi - lo)) ? d1.x : 0.f; d1.y = in_win(3, lo, (uint32_t)(hi - lo)) ? d1.y : 0.f;
      }
      wacc_quad(a, d0, d1, r);
    }
    wacc_lane<NT>(a, tid, (float)P.tail.ic, &t1, &t2, &tx);
  }
  // ---- reductions of the sweeps and of the tail sums (before the SG pass: fourteen registers less across it)
  {
#if LDSP_L3_BFLY
    // (maxima by butterfly, wave_prims.hpp: four values in ten instructions, totals in lanes 15 / 47 / 31 / 63 = mx0 / mx1 / mx2 / -mn0,
    // each of which posts its own; max and min do not depend on the order: the same bits)
    float nm0 = -mn0, nm2 = -mn2;   // max(trap(-y)) = -min(trap(y))
    LDSP_BFLY4("v_max_f32", "v_max_f32_dpp", mx0, mx1, mx2, nm0);
#if LDSP_L3_BFLY_SUMS
    float z0 = 0.f;
    LDSP_BFLY4("v_add_f32", "v_add_f32_dpp", t1, t2, tx, z0);   // totals in lanes 15 / 47 / 31 of t1
    LDSP_DPP_GROUP1("v_max_f32_dpp", nm2);
#else
    LDSP_DPP_GROUP4("v_max_f32_dpp", nm2, "v_add_f32_dpp", t1, "v_add_f32_dpp", t2, "v_add_f32_dpp", tx);
#endif
    wave_argmax(bo_v, bo_i);
    if ((lane & 15) == 15) {
      atomicMax(&S.sl->fmx[lane == 15 ? FX_F0 : lane == 47 ? FX_F1 : lane == 31 ? FX_F2 : FX_F0I], ford(mx0));
#if LDSP_L3_BFLY_SUMS
      if (lane != 63) S.wsum[(W_PZ + (lane == 15 ? 0 : lane == 47 ? 1 : 2)) * NW + wave] = t1;
#endif
    }
    if (lane == 63) {
      atomicMax(&S.sl->fmx[FX_F2I], ford(nm2));
      atomicMax(&S.sl->vi[VI_OPT], pack_vi(bo_v, bo_i));
#if !LDSP_L3_BFLY_SUMS
      S.wsum[(W_PZ + 0) * NW + wave] = t1; S.wsum[(W_PZ + 1) * NW + wave] = t2; S.wsum[(W_PZ + 2) * NW + wave] = tx;
#endif
    }
#else
    LDSP_DPP_GROUP8("v_max_f32_dpp", mx0, "v_max_f32_dpp", mx1, "v_max_f32_dpp", mx2, "v_min_f32_dpp", mn0, "v_min_f32_dpp", mn2, "v_add_f32_dpp", t1, "v_add_f32_dpp", t2, "v_add_f32_dpp", tx);
    wave_argmax(bo_v, bo_i);
    if (lane == 63) {
      atomicMax(&S.sl->fmx[FX_F0], ford(mx0));
      atomicMax(&S.sl->fmx[FX_F1], ford(mx1));
      atomicMax(&S.sl->fmx[FX_F2], ford(mx2));
      atomicMax(&S.sl->fmx[FX_F0I], ford(-mn0));   // max(trap(-y)) = -min(trap(y))
      atomicMax(&S.sl->fmx[FX_F2I], ford(-mn2));
      atomicMax(&S.sl->vi[VI_OPT], pack_vi(bo_v, bo_i));
      S.wsum[(W_PZ + 0) * NW + wave] = t1; S.wsum[(W_PZ + 1) * NW + wave] = t2; S.wsum[(W_PZ + 2) * NW + wave] = tx;
    }
#endif
  }
  LDSP_BAR_MAIN();   // every read of T is done
  // ---- Savitzky-Golay derivatives, current maxima (dsp_icpc.jl:181-186): g[k] = sum_i c[i] y[k+i] (valid mode, trailing time
  // axis).  The four outputs of a quad from the M + 3 samples w[0 .. M+2] = the quad and its halo (the next lanes' registers by
  // DPP), one multiply-add per tap and output.  This pass takes the statistics (maximum, baseline sums, current maxima); the
  // masks of the main filter's output need thresholds that follow from them: the outputs wait in X — each thread's own quads, which
  // nobody else reads — instead of in sixteen registers across the exchange.
  const int ng = L - P.sg_npts[0] + 1;
  auto sg_window = [&](int r, float (&wv_)[4 + 4 * NHS]) {
    wv_[0] = y[r].x; wv_[1] = y[r].y; wv_[2] = y[r].z; wv_[3] = y[r].w;
    f4 h = y[r];
#pragma unroll
    for (int j = 0; j < NHS; ++j) {
      h = halo_next(h, r, j);
      wv_[4 + 4 * j] = h.x; wv_[5 + 4 * j] = h.y; wv_[6 + 4 * j] = h.z; wv_[7 + 4 * j] = h.w;
    }
  };
  float sgc0[M];
#pragma unroll
  for (int i = 0; i < M; ++i) {
    sgc0[i] = P.sg_c[0][i];   // zero beyond the filter's own taps (the host zero-fills the block)
    if (LDSP_L3_SGC_VGPR) asm volatile("" : "+v"(sgc0[i]));
  }
  auto sg_main = [&](int r, const float (&wv_)[4 + 4 * NHS], float (&go)[4]) {   // -inf beyond the output axis
    const int i0 = 4 * (opaque(tid) + NT * r);
#pragma unroll
    for (int e = 0; e < 4; ++e) go[e] = 0.f;
    if constexpr (M <= 13) {
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const float c = sgc0[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) go[e] = fmaf(c, wv_[e + i], go[e]);
      }
    } else {   // taps 4j .. 4j+3 read the samples 4j .. 4j+6 of the window: the quads j and j + 1
      f4 h0 = y[r];
#pragma unroll
      for (int j = 0; j < (M + 3) / 4; ++j) {
        const int last = (4 * j + 3 < M - 1) ? 4 * j + 3 : M - 1;   // the chunk's last tap
        f4 h1 = h0;
        if (last + 3 >= 4 * j + 4) h1 = halo_next(h0, r, j);
        const float win[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (4 * j + t >= M) continue;
          const float c = sgc0[4 * j + t];
#pragma unroll
          for (int e = 0; e < 4; ++e) go[e] = fmaf(c, win[e + t], go[e]);
        }
        h0 = h1;
      }
    }
    if (wrow(r) + 255 >= ng) {   // only the wave-row holding the end of the output axis
#pragma unroll
      for (int e = 0; e < 4; ++e) go[e] = (i0 + e < ng) ? go[e] : -INFINITY;
    }
  };
  {
    float gmax = -INFINITY, g_s1 = 0.f, g_s2 = 0.f;   // maximum; sums over the baseline window (pivot 0: a derivative has no level)
    float bv[4]; int bi[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { bv[f] = -INFINITY; bi[f] = 0x7fffffff; }
    const int cw_from[4] = {P.cur_from[0], P.cur_from[1], P.cur_from[2], P.cur_from[3]};
    const uint32_t cw_len[4] = {(uint32_t)(P.cur_until[0] - P.cur_from[0]), (uint32_t)(P.cur_until[1] - P.cur_from[1]),
                                (uint32_t)(P.cur_until[2] - P.cur_from[2]), (uint32_t)(P.cur_until[3] - P.cur_from[3])};
    // (IN: the row lies wholly inside the filter's current window — what rowcls says of most rows that meet it — and the window test
    // of every sample, a subtraction, a compare and a select, falls away)
    auto track = [&](int f, int k, float gv, bool in) {
      const float gw = (in || (uint32_t)(k - cw_from[f]) <= cw_len[f]) ? gv : -INFINITY;
      const bool gt = gw > bv[f];
      bv[f] = gt ? gw : bv[f];
      bi[f] = gt ? k : bi[f];
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i0 = 4 * (opaque(tid) + NT * r);
      float wv_[4 + 4 * NHS], go[4];
      if constexpr (M <= 13) sg_window(r, wv_);
      sg_main(r, wv_, go);
      gmax = vmax3(vmax3(gmax, go[0], go[1]), go[2], go[3]);
      *reinterpret_cast<f4*>(&S.X[i0]) = (f4){go[0], go[1], go[2], go[3]};
      if (!row_out(cls_sgbl, r)) {   // sgbl.until <= ng-1: -inf never enters
        float dd[4] = {go[0], go[1], go[2], go[3]};
        if (!row_in(cls_sgbl, r)) {
          const int lo = P.sgbl.from - i0, hi = P.sgbl.until - i0;
#pragma unroll
          for (int e = 0; e < 4; ++e) dd[e] = in_win(e, lo, (uint32_t)(hi - lo)) ? dd[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { g_s1 += dd[e]; g_s2 = fmaf(dd[e], dd[e], g_s2); }
      }
      if (!row_out(cls_cur0, r)) {
        if (row_in(cls_cur0, r)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) track(0, i0 + e, go[e], true);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) track(0, i0 + e, go[e], false);
        }
      }
      if (!row_out(cls_curx, r)) {   // SG(60 ns), SG(100 ns), plain derivative: only rows that touch the current window
        // (one filter at a time, its four outputs tracked before the next one's are computed)
        if constexpr (M > 13) sg_window(r, wv_);
        auto others = [&](auto in_tag) {
        constexpr bool IN = decltype(in_tag)::value;
#pragma unroll
        for (int f = 1; f < 3; ++f) {
          if (f == 2 && (STUDY || P.sg_same_02)) continue;
          float gf[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < MS; ++i) {
            const float c = P.sg_c[f][i];
#pragma unroll
            for (int e = 0; e < 4; ++e) gf[e] = fmaf(c, wv_[e + i], gf[e]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) track(f, i0 + e, gf[e], IN);
          __builtin_amdgcn_sched_barrier(0);
        }
        // y[k] - y[k-1]: the sample before the quad is the previous lane's last one (lane 0: the previous wave-row's, from the table)
        float ypv;   // (as an asm statement: through __builtin_amdgcn_update_dpp hipcc 7.2 shifted y[r].x here, not y[r].w)
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(ypv) : "v"(y[r].w));
        if (lane == 0) ypv = (i0 > 0) ? hyl[r * NW + wave] : 0.f;
        float g3[4] = {y[r].x - ypv, y[r].y - y[r].x, y[r].z - y[r].y, y[r].w - y[r].z};   // y[k] - y[k-1]
        if (i0 == 0) g3[0] = y[r].y - y[r].x;                                                // y[max(i,1)] - y[max(i-1,0)] at i = 0
#pragma unroll
        for (int e = 0; e < 4; ++e) track(3, i0 + e, g3[e], IN);
        };
        if (row_in(cls_curi, r)) others(std::true_type{});
        else others(std::false_type{});
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  STAMP(6); DSTOP(6); LDSP_PROBE();
  // ---- the SG pass's own reductions
#if LDSP_L3_BFLY_SUMS
  LDSP_BFLY2("v_add_f32", "v_add_f32_dpp", g_s1, g_s2);   // totals in lanes 31 / 63 of g_s1
  LDSP_DPP_GROUP1("v_max_f32_dpp", gmax);
#else
  LDSP_DPP_GROUP3("v_max_f32_dpp", gmax, "v_add_f32_dpp", g_s1, "v_add_f32_dpp", g_s2);
#endif
  {
    float v0 = bv[0], v1 = bv[1], v2 = bv[2], v3 = bv[3];
#if LDSP_L3_BFLY
    // (the four current-window maxima together, then the first index of each: totals in lanes 15 / 47 / 31 / 63 = filter 0 / 1 / 2 / 3,
    // each of which posts its own packed (value, index))
    LDSP_BFLY4("v_max_f32", "v_max_f32_dpp", v0, v1, v2, v3);
    const float m0 = readlane_f(v0, 15), m1 = readlane_f(v0, 47), m2 = readlane_f(v0, 31), m3 = readlane_f(v0, 63);
    uint32_t k0 = (bv[0] == m0) ? (uint32_t)bi[0] : 0x7fffffffu, k1 = (bv[1] == m1) ? (uint32_t)bi[1] : 0x7fffffffu,
             k2 = (bv[2] == m2) ? (uint32_t)bi[2] : 0x7fffffffu, k3 = (bv[3] == m3) ? (uint32_t)bi[3] : 0x7fffffffu;
    LDSP_BFLY4("v_min_u32", "v_min_u32_dpp", k0, k1, k2, k3);
    if ((lane & 15) == 15) {
      const int f = lane == 15 ? 0 : lane == 47 ? 1 : lane == 31 ? 2 : 3;
      if (k0 != 0x7fffffffu) atomicMax(&S.sl->vi[VI_CUR0 + f], pack_vi(f == 0 ? m0 : f == 1 ? m1 : f == 2 ? m2 : m3, (int)k0));
    }
#if LDSP_L3_BFLY_SUMS
    if (lane == 63) atomicMax(&S.sl->fmx[FX_G], ford(gmax));
    if ((lane & 31) == 31) S.wsum[(W_SGB + (lane >> 5)) * NW + wave] = g_s1;
#else
    if (lane == 63) {
      atomicMax(&S.sl->fmx[FX_G], ford(gmax));
      S.wsum[(W_SGB + 0) * NW + wave] = g_s1; S.wsum[(W_SGB + 1) * NW + wave] = g_s2;
    }
#endif
#else
    LDSP_DPP_GROUP4("v_max_f32_dpp", v0, "v_max_f32_dpp", v1, "v_max_f32_dpp", v2, "v_max_f32_dpp", v3);
    uint32_t k0 = (bv[0] == readlane_f(v0, 63)) ? (uint32_t)bi[0] : 0x7fffffffu, k1 = (bv[1] == readlane_f(v1, 63)) ? (uint32_t)bi[1] : 0x7fffffffu,
             k2 = (bv[2] == readlane_f(v2, 63)) ? (uint32_t)bi[2] : 0x7fffffffu, k3 = (bv[3] == readlane_f(v3, 63)) ? (uint32_t)bi[3] : 0x7fffffffu;
    LDSP_DPP_GROUP4("v_min_u32_dpp", k0, "v_min_u32_dpp", k1, "v_min_u32_dpp", k2, "v_min_u32_dpp", k3);
    if (lane == 63) {
      atomicMax(&S.sl->fmx[FX_G], ford(gmax));
      if (k0 != 0x7fffffffu) atomicMax(&S.sl->vi[VI_CUR0], pack_vi(v0, (int)k0));
      if (k1 != 0x7fffffffu) atomicMax(&S.sl->vi[VI_CUR1], pack_vi(v1, (int)k1));
      if (k2 != 0x7fffffffu) atomicMax(&S.sl->vi[VI_CUR2], pack_vi(v2, (int)k2));
      if (k3 != 0x7fffffffu) atomicMax(&S.sl->vi[VI_CUR3], pack_vi(v3, (int)k3));
      S.wsum[(W_SGB + 0) * NW + wave] = g_s1; S.wsum[(W_SGB + 1) * NW + wave] = g_s2;
    }
#endif
  }
  }
  STAMP(7); DSTOP(7);
  LDSP_BAR_MAIN();   // the round's results are posted
  STAMP(8); DSTOP(8);

  // ================================================================== round 3: the masks of the SG output (own quads of X); y -> X
  // in-trace pile-up threshold (dsp_routines.jl:75-77) and t50_current threshold (dsp_icpc.jl:192) (g = -inf beyond the output
  // axis: bit 0)
  {
    float thr_intr, thr_sg50;
    const float s1 = fold_partials<NW>(S.wsum + (W_SGB + 0) * NW, 0.f, [](float a, float b) { return a + b; });
    const float s2 = fold_partials<NW>(S.wsum + (W_SGB + 1) * NW, 0.f, [](float a, float b) { return a + b; });
    const float m_ = s1 * (float)P.sgbl.inv_n;
    const float var_ = fmaxf(fmaf(s2, (float)P.sgbl.inv_n, -m_ * m_), 0.f);
    thr_intr = __builtin_amdgcn_sqrtf(var_) * P.intrace_nsigma;
    if (thr_intr == 0.f) thr_intr = 1.f;
    thr_sg50 = ford_inv(S.sl->fmx[FX_G]) * 0.5f;
    if (tid == 0) { S.misc[MS_THRI] = thr_intr; S.misc[MS_THR5] = thr_sg50; }
    uint32_t wi[R], w5[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const f4 gq = *reinterpret_cast<const f4*>(&S.X[4 * (tid + NT * r)]);
      wi[r] = nib_ge(gq, thr_intr); w5[r] = nib_ge(gq, thr_sg50);
    }
    s4_pack4(wi, lane); s4_pack4(w5, lane);
    // y -> X (each thread overwrites the quads it has just read)
#pragma unroll
    for (int r = 0; r < R; ++r) *reinterpret_cast<f4*>(&S.X[4 * (tid + NT * r)]) = y[r];
    if (tid < 64) S.X[Lp + tid] = 0.f;
    if ((lane & 7) == 7) {
      uint32_t* bi_ = S.bm + M_INTR * NWORDS + 8 * wave + (lane >> 3);
      uint32_t* b5_ = S.bm + M_SG50 * NWORDS + 8 * wave + (lane >> 3);
#pragma unroll
      for (int r = 0; r < R; ++r) { bi_[(NT / 8) * r] = wi[r]; b5_[(NT / 8) * r] = w5[r]; }
    }
  }
  STAMP(9); DSTOP(9);
  LDSP_BAR_MAIN();   // X = y and the four masks are complete
  STAMP(10); DSTOP(10);

  // ================================================================== round 4: run scans on the masks, threshold confirmation
  const float e_maxl = S.misc[MS_EMAX];
  // filter f at output index k from y in LDS (the few samples the parabolas and crossing interpolations need)
  auto flt_at = [&](int f, int k) -> float {
    if (f < 3) {
      float gq = 0.f;
      constexpr int CH = M <= 13 ? M : 9;   // samples read together (one wait)
#pragma unroll
      for (int c0 = 0; c0 < M; c0 += CH) {
        float a[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) a[i] = (c0 + i < M) ? S.X[k + c0 + i] : 0.f;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < CH; ++i)
          if (c0 + i < M) gq = fmaf(P.sg_c[f][c0 + i], (c0 + i < P.sg_npts[f]) ? a[i] : 0.f, gq);
      }
      return gq;
    }
    return S.X[max(k, 1)] - S.X[max(k - 1, 0)];
  };
  // Intersect scans on the bit-masks (thread <-> word), see icpc_lean.hip
  static_assert(2 * NWORDS == NT, "one t0 / inverted-t0 word per thread");
  if (STUDY || (P.t0_mintot <= 97 && P.intrace_mintot <= 32)) {   // block-uniform
    const int q = tid / NWORDS, wd = tid % NWORDS;
    const uint32_t* b0 = S.bm + (M_T0 + q) * NWORDS;
    const bool has_i = tid < NWORDS;
    const uint32_t* bi2 = S.bm + (has_i ? M_INTR : M_SG50) * NWORDS;
    uint32_t t[5], u[3];
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] = b0[min(max(wd + k - 1, 0), NWORDS - 1)];
#pragma unroll
    for (int k = 0; k < 3; ++k) u[k] = bi2[min(max(wd + k - 1, 0), NWORDS - 1)];
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
  // Confirmation of the five threshold candidates (see icpc_lean.hip): imin[q] = first QUAD with a sample at or above
  // threshold q; lane q of every wave finds the sample, checks that it is not sample 0 and that the next tx_mintot - 1 samples
  // stay at or above the threshold.  A trace that fails runs the general scan (bit-masks of y by ballot, run scan on the words).
  int p_tx = 0x7fffffff;   // lane q < 5: first confirmed sample of threshold q
  {
    const int q = min(lane, 4);
    const float thrq = e_maxl * ((q == 0) ? 0.1f : (q == 1) ? 0.5f : (q == 2) ? 0.8f : (q == 3) ? 0.9f : 0.99f);
    const int qd = S.sl->imin[IM_TX0 + q];
    bool ok = e_maxl > 0.f;
    if (ok && qd != 0x7fffffff) {
      const f4 v = *reinterpret_cast<const f4*>(&S.X[4 * qd]);
      const int e = (v.x >= thrq) ? 0 : (v.y >= thrq) ? 1 : (v.z >= thrq) ? 2 : 3;
      p_tx = 4 * qd + e;
      ok = p_tx >= 1 && p_tx + P.tx_mintot <= L;
      for (int j = 1; ok && j < P.tx_mintot; ++j) ok = S.X[p_tx + j] >= thrq;
    }
    if (!STUDY && __ballot(lane < 5 && !ok) != 0ull) {   // block-uniform
      __syncthreads();
      if (tid < 5) S.sl->imin[IM_TX0 + tid] = 0x7fffffff;
      for (int m = 0; m < SP; ++m) {
        const float yv = (FULL || tid + NT * m < L) ? S.X[tid + NT * m] : -INFINITY;   // (nothing beyond the trace crosses)
#pragma unroll
        for (int qq = 0; qq < 5; ++qq) {
          const unsigned long long b = __ballot(yv >= e_maxl * ((qq == 0) ? 0.1f : (qq == 1) ? 0.5f : (qq == 2) ? 0.8f : (qq == 3) ? 0.9f : 0.99f));
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
  STAMP(11); DSTOP(11);
  LDSP_BAR_MAIN();
  STAMP(12); DSTOP(12);

  // ================================================================== round 5: crossings, estimators, finishing lanes
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
      if (has50) t50cur_us = (tg_first + P.dt * ((float)(p - 1) + (S.misc[MS_THR5] - yl5) / (yh5 - yl5))) * P.inv_unit_per_us;
      if (intr_n > 0) {   // reversed index pos' = ng-1-e; r[pos'-1] = g[e+1], r[pos'] = g[e]
        const int pr = ng - 1 - e;
        const float xl = tg_first + P.dt * (float)(pr - 1);
        const float xr_ = (S.misc[MS_THRI] - yli) * P.dt / (yhi - yli) + xl;
        intr_x = (tg_first + P.dt * (float)(ng - 1)) - xr_;   // last(time) - x   (dsp_routines.jl:81)
      }
      S.outv[C_t50_current] = t50cur_us; S.outv[C_inTrace_intersect] = intr_x; S.outv[C_inTrace_n] = __int_as_float(intr_n);
    }
  }
  // finishing lanes: the window statistics from the per-wave partials, the results of sweep B (waves that have nothing else to do here)
  if (tid == (64 * (4 % NW))) {   // signalstats(bl): sigma, slope, offset (dsp_icpc.jl:102)
    float s = 0.f, s2 = 0.f, sx = 0.f;
    for (int ww = 0; ww < NW; ++ww) { s += S.wred[ww]; s2 += S.wred[NW + ww]; sx += S.wred[2 * NW + ww]; }
    float m_, blsigma, blslope, bloffset;
    win_finish(s, s2, sx, P.bl, pv_bl, P.t_first, P.dt, &m_, &blsigma, &blslope, &bloffset);
    const float blm = S.misc[MS_BLMEAN];
    S.outv[C_blmean] = blm; S.outv[C_blsigma] = blsigma; S.outv[C_blslope] = blslope; S.outv[C_bloffset] = bloffset;
    S.outv[C_e_max] = S.misc[MS_RAWMAX] - blm; S.outv[C_e_min] = S.misc[MS_RAWMIN] - blm;
  }
  if (tid == (64 * (4 % NW)) + 1 % NT) {   // tailstats -> (mean, sigma, tau)
    float tail_mean = 0.f, tail_sigma = 0.f, tail_tau = 0.f;
    if (S.sl->isum[IS_TAILBAD] == 0) {
      float s1 = 0.f, s2 = 0.f, sx = 0.f, sl, of;
      for (int ww = 0; ww < NW; ++ww) { s1 += S.wsum[(W_TAIL + 0) * NW + ww]; s2 += S.wsum[(W_TAIL + 1) * NW + ww]; sx += S.wsum[(W_TAIL + 2) * NW + ww]; }
      constexpr float LN2 = 0.693147180559945309f;   // the sums and their pivot are in log2 units
      win_finish(s1 * LN2, s2 * (LN2 * LN2), sx * LN2, P.tail, S.misc[MS_PVTL] * LN2, P.t_first, P.dt, &tail_mean, &tail_sigma, &sl, &of);
      tail_tau = -__builtin_amdgcn_rcpf(sl);
    }
    S.outv[C_tail_tau] = tail_tau; S.outv[C_tail_mean] = tail_mean; S.outv[C_tail_sigma] = tail_sigma;
  }
  if (tid == (64 * (6 % NW)) + 2 % NT) {   // signalstats of the pole-zero corrected tail
    float s1 = 0.f, s2 = 0.f, sx = 0.f;
    for (int ww = 0; ww < NW; ++ww) { s1 += S.wsum[(W_PZ + 0) * NW + ww]; s2 += S.wsum[(W_PZ + 1) * NW + ww]; sx += S.wsum[(W_PZ + 2) * NW + ww]; }
    float tailmean, tailsigma, tailslope, tailoffset;
    win_finish(s1, s2, sx, P.tail, S.misc[14], P.t_first, P.dt, &tailmean, &tailsigma, &tailslope, &tailoffset);
    S.outv[C_tailmean] = tailmean; S.outv[C_tailsigma] = tailsigma; S.outv[C_tailslope] = tailslope; S.outv[C_tailoffset] = tailoffset;
  }
  if (tid == (64 * (6 % NW)) + 3 % NT) {   // results of sweep B
    float v; int i;
    unpack_vi(S.sl->vi[VI_OPT], &v, &i);
    S.outv[C_e_trap_max] = v; S.outv[C_t_trap_max] = P.t_first + P.dt * (float)(i + P.opt.flen - 1);
    S.outv[C_e_10410] = ford_inv(S.sl->fmx[FX_F0]); S.outv[C_e_535] = ford_inv(S.sl->fmx[FX_F1]); S.outv[C_e_313] = ford_inv(S.sl->fmx[FX_F2]);
    S.outv[C_e_10410_inv] = ford_inv(S.sl->fmx[FX_F0I]); S.outv[C_e_313_inv] = ford_inv(S.sl->fmx[FX_F2I]);   // trap(-y) = -trap(y)  (dsp_icpc.jl:199-204)
  }
  // crossing positions (sample units, split int + frac); NaN -> 0 us (dsp_routines.jl:24,41).  Seven interpolations, one per
  // lane (q < 5: threshold q of y; 5: t0; 6: inverted t0), evaluated by every wave that uses a position.  The two values of the
  // t0 trapezoid about its crossing are window sums of y (T has left the LDS): the wave sums the long leg once, at p - 1, and
  // slides it by one sample.
  Pos ptx1 = {0, 0.f}, ptx2 = {0, 0.f}, pt0 = {0, 0.f};
  constexpr int W_CZWIN = 4 % NW;   // the wave that places the CUSP / ZAC estimator windows
  if (wave <= 2 || wave == W_CZWIN || wave == NW - 1) {
    const int q = min(lane, 6);
    const int p = (q < 5) ? p_tx : S.sl->imin[q];
    const bool has = (q < 5) ? p != 0x7fffffff : S.sl->isum[IS_T0 + q - 5] > 0;
    const float frac = (q == 0) ? 0.1f : (q == 1) ? 0.5f : (q == 2) ? 0.8f : (q == 3) ? 0.9f : 0.99f;
    const float thr = (q < 5) ? e_maxl * frac : P.t0_thr;
    float tl = 0.f, th = 0.f;   // lanes 5, 6: the t0 trapezoid at p - 1 and p
    if (wave == 1 % NW || wave == NW - 1) {   // the waves that use the t0 position (qdrift; the times)
      const TrapDev t0 = P.t0;
#pragma unroll
      for (int qq = 5; qq < 7; ++qq) {
        if (S.sl->isum[IS_T0 + qq - 5] <= 0) continue;   // (wave-uniform)
        const int pp = S.sl->imin[qq] - 1;
        const float a = wave_box(S.X, pp + t0.n1 + t0.g, t0.n2, lane);
        const float b0 = (t0.n1 <= 3) ? 0.f : wave_box(S.X, pp, t0.n1, lane);
        if (lane == qq) {
          const float a1 = a + S.X[pp + t0.flen] - S.X[pp + t0.n1 + t0.g];
          float bl_, bh_;
          if (t0.n1 <= 3) {
            const float y0_ = S.X[pp], y1_ = S.X[pp + 1], y2_ = S.X[pp + 2], y3_ = S.X[pp + 3];
            bl_ = y0_ + ((t0.n1 >= 2) ? y1_ : 0.f) + ((t0.n1 >= 3) ? y2_ : 0.f);
            bh_ = y1_ + ((t0.n1 >= 2) ? y2_ : 0.f) + ((t0.n1 >= 3) ? y3_ : 0.f);
          } else {
            bl_ = b0; bh_ = b0 + S.X[pp + t0.n1] - S.X[pp];
          }
          tl = a * t0.inv2 - bl_ * t0.inv1; th = a1 * t0.inv2 - bh_ * t0.inv1;
          if (qq == 6) { tl = -tl; th = -th; }
        }
      }
    }
    Pos pp; pp.ip = 0; pp.fp = -P.t_first / P.dt;   // sample position of t = 0
    float us = 0.f;
    if (has) {
      float yl, yh; int base;
      if (q < 5) {
        yl = S.X[p - 1]; yh = S.X[p]; base = p - 1;
      } else {
        yl = tl; yh = th;
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
  STAMP(13); DSTOP(13);

  // ---- signal estimators
  {
    lds_float* eslot = S.misc + 4;
    if (wave == 0) {
      // e_trap = SignalEstimator(trap_opt output, t50 + rt + ft/2)  (dsp_icpc.jl:163): lane l = window point l.  The trapezoid
      // at the window's first point from two window sums of y, at the others by sliding both legs (prefix sums over the lanes of
      // the samples entering and leaving).
      Pos p = pos_add(ptx1, P.trap_pickoff);
      p.ip -= (P.opt.flen - 1);
      const int nsig = L - P.opt.flen + 1;
      float v = NAN;
      if (nsig >= P.sig_est.npts) {
        const TrapDev fo = P.opt;
        int i0; float u;
        est_window(P.sig_est, p, nsig, &i0, &u);
        float a = 0.f, b = 0.f;
        for (int j = lane; j < fo.n2; j += 64) a += S.X[i0 + fo.n1 + fo.g + j];
        for (int j = lane; j < fo.n1; j += 64) b += S.X[i0 + j];
        float da = 0.f, db = 0.f;   // lane l >= 1: what the legs gain from point l - 1 to point l
        if (lane >= 1 && lane < P.sig_est.npts) {
          const int k = i0 + lane - 1;
          da = S.X[k + fo.flen] - S.X[k + fo.n1 + fo.g];
          db = S.X[k + fo.n1] - S.X[k];
        }
        LDSP_DPP_GROUP4("v_add_f32_dpp", a, "v_add_f32_dpp", b, "v_add_f32_dpp", da, "v_add_f32_dpp", db);
        const float al = readlane_f(a, 63) + da, bl_ = readlane_f(b, 63) + db;
        float t = 0.f;
        if (lane < P.sig_est.npts) t = est_weight(P.sig_est, S.estB, lane, u) * (al * fo.inv2 - bl_ * fo.inv1);
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
        if (lane == 0) { S.misc[8] = __int_as_float(i0c); S.misc[9] = uc; S.misc[12] = S.X[i0c]; }
      }
      if (nout_z >= P.sig_est.npts) {
        Pos pz2 = pos_add(ptx1, P.zac_pickoff);
        pz2.ip -= (P.zac.Lf - 1);
        int i0z; float uz;
        est_window(P.sig_est, pz2, nout_z, &i0z, &uz);
        if (lane == 0) { S.misc[10] = __int_as_float(i0z); S.misc[11] = uz; S.misc[13] = S.X[i0z]; }
      }
    }
    // (two-wave tiles: wave 1 alone, both passes — `2 % NW` named wave 0 as well, which has no t0 position (the block above computes it in
    // waves 1 % NW and NW - 1 only), and the two waves raced for the result slots: qdrift = NaN in some launches of the 128-thread tile,
    // found by the second-launch comparison of tests/test_icpc_gpu.py::test_small_tiles_run_the_lean_kernel in round 4)
    if ((NW > 2) ? (wave == 1 || wave == 2) : (wave == 1 % NW)) {
      const bool lq = (NW > 2) ? wave == 2 : false;
      for (int pass = 0; pass < ((NW > 2) ? 1 : 2); ++pass) {
        const bool is_lq = (NW > 2) ? lq : pass == 1;
        const Pos base = is_lq ? ptx2 : pt0;
        const float d1 = is_lq ? P.lq_d1 : P.qdrift_d1, d2 = is_lq ? P.lq_d2 : P.qdrift_d2;
        const Pos p1 = pos_add(base, d1), p2 = pos_add(base, d2);
        const int ips[3] = {base.ip, p1.ip, p2.ip};
        const float fps[3] = {base.fp, p1.fp, p2.fp};
        float* scr = (5 * NWORDS >= 1024) ? reinterpret_cast<float*>(S.bm + M_FB * NWORDS) + (is_lq ? 512 : 0) : nullptr;
        const float res = qdrift_wave(P.int_est, S.estB + EST_TBL, S.X, L, ips, fps, scr);
        if (lane == 0) eslot[is_lq ? 2 : 1] = res;
      }
    }
    // get_wvf_maximum of the four current signals (src/interpolation.jl:30-46)
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
    STAMP(14); DSTOP(14);
  }
  {
    lds_float* eslot = S.misc + 4;
    LDSP_BAR_MAIN();   // every read of X = y is done
    if (tid == 0) {
      S.outv[C_e_trap] = eslot[0];
      S.outv[C_qdrift] = eslot[1];
      S.outv[C_lq] = eslot[2];
    }
  }
  STAMP(15); DSTOP(15);

  // ------------------------------------------------------------------------------------ phase 7: CUSP / ZAC (dsp_icpc.jl:167-178)
  // Closed form (CUSP / ZAC): with d[i] = y[i] - a*y[i-1],
  //   out[k] = sc * ( sum_{j<=Lf-2} w[j] d[n-j] + w[Lf-1] y[k] ),  n = k+Lf-1,
  // w = sinh flanks + flat top (+ parabolas for ZAC) splits into a causal one-pole G, an anti-causal one-pole A, the prefix sum
  // Dp of d, and a double prefix sum of a sparse combination u of Dp; each is built in the S4 view, stored to X and read back
  // lane-strided, two rows per step.  X takes them in turns:  Dp -> u -> PRF -> G -> A  (ZAC), Dp -> G -> A (CUSP alone).
  // y is still in the thread's registers (S4 view).
  // SEP (CUSP and ZAC optimised separately: two passes): the second pass needs y again.  Kept in sixteen more registers across the
  // first pass it cost the third workgroup per CU (124 VGPRs); it is REBUILT instead — the raw samples read again (32 KB that left
  // the L2 a few tens of microseconds ago) and put through the statements of the y loop above, with the wave-row table that is
  // still in LDS: the same bits.
  auto rebuild_y = [&]() __attribute__((always_inline)) {
    const float* w2 = w; const uint16_t* w16_2 = w16;
    asm volatile("" : "+s"(w2), "+s"(w16_2));
    load_raw(y, w2, w16_2);
    const f2 pv = splat(wv(P.bl.from));
    const float delta = S.misc[MS_DELTA];
    float t[R], i1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      y[r].xy -= pv; y[r].zw -= pv;
      t[r] = ((y[r].x + y[r].y) + y[r].z) + y[r].w;   // (c3 of the first pass: the same order of additions)
      i1[r] = t[r];
    }
    LDSP_DPP_GROUP4("v_add_f32_dpp", i1[0], "v_add_f32_dpp", i1[1], "v_add_f32_dpp", i1[2], "v_add_f32_dpp", i1[3]);
    const float c = P.pz_c, cd = c * delta;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float fi = (float)(4 * (opaque(tid) + NT * r));
      const float Br = S.scn[2 * R * NW + 4 * wave + r];
      const float c0 = y[r].x, c1 = c0 + y[r].y, c2 = c1 + y[r].z, c3 = c2 + y[r].w;
      const f4 xs = (f4){y[r].x - delta, y[r].y - delta, y[r].z - delta, y[r].w - delta};
      const float ky = fmaf(c, i1[r] - t[r], Br) - cd * (fi + 1.f);
      y[r].x = xs.x + fmaf(c, c0, ky);
      y[r].y = xs.y + fmaf(c, c1, ky - cd);
      y[r].z = xs.z + fmaf(c, c2, ky - 2.f * cd);
      y[r].w = xs.w + fmaf(c, c3, ky - 3.f * cd);
    }
  };
  const float y0 = S.misc[15];
  float* hyl_cz = S.hy + 4 * NH * (R * NW + 1);
  // the sample before each quad (d[i] = y[i] - a*y[i-1]): the previous lane's last one; lane 0: the previous wave-row's, from the table
  auto y_before = [&](int r) {
    float v;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(y[r].w));
    if (lane == 0) v = (r * NW + wave > 0) ? hyl_cz[r * NW + wave] : 0.f;
    return v;
  };
  auto cz_pass = [&](auto wc_tag, auto wz_tag, const CuspZacDev& Z, const CuspZacDev& ZZ, const float cpiv) {
    constexpr bool WC = decltype(wc_tag)::value, WZ = decltype(wz_tag)::value;
    const int Lf = Z.Lf;
    const int nout = L - Lf + 1, lt = Z.lt, f1 = Z.f1, ltp = Z.ltp;
    const int pad = cz_pad_floats(Lf);
    // Pivot (see icpc_lean.hip): the stage runs on y' = y - cpiv, cpiv = the level at the left edge of the pick-off window.
    // y' is never materialised:  Dp' = Dp - eps*cpiv*i,  d' = d - eps*cpiv,  the taps on y[k] fold cpiv into their fma.
    const float mec = -Z.eps * cpiv;
    auto rd2 = [&](const float* p, int m) { return mk2(p[NT * m], p[NT * (m + 1)]); };
    auto wr2 = [&](float* p, int m, f2 v) {
      const uint32_t a = (uint32_t)(uintptr_t)(lds_float*)(p + NT * m);
      asm volatile("ds_write2st64_b32 %0, %1, %2 offset1:%3" : : "v"(a), "v"(v.x), "v"(v.y), "n"(NT / 64) : "memory");
    };
    // ---- Dp'[i] = y[i] - y[0] + eps*T[i] + mec*i -> X (each thread its own quads; T rebuilt from the wave-row prefix)
    auto store_dp = [&]() {
      const f2 e2 = splat(Z.eps), y02 = splat(y0);
      const float bf = (float)(4 * tid);
      const f2 l01 = splat(mec) * mk2(bf, bf + 1.f), l23 = splat(mec) * mk2(bf + 2.f, bf + 3.f);
      // T of a quad = T at the wave-row's first sample (the row table - delta (i + c i (i+1) / 2)) + the sum of the wave-row's
      // samples before the quad (a DPP scan of the quad sums) + the running sum inside the quad
      float p0[R], tt[R];
#pragma unroll
      for (int r = 0; r < R; ++r) { const f2 t = y[r].xy + y[r].zw; tt[r] = t.x + t.y; p0[r] = tt[r]; }
#ifdef LDSP_WHATIF_S16
      LDSP_DPP_GROUP1("v_add_f32_dpp", p0[0]);
#else
      LDSP_DPP_GROUP4("v_add_f32_dpp", p0[0], "v_add_f32_dpp", p0[1], "v_add_f32_dpp", p0[2], "v_add_f32_dpp", p0[3]);
#endif
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float fs = (float)wrow(r);
        const float hw_ = S.scn[4 * wave + r], lw_ = S.scn[R * NW + 4 * wave + r] - S.misc[MS_DELTA] * (fs * fmaf(0.5f * P.pz_c, fs + 1.f, 1.f));
        f4 t = t_quad(y[r], p0[r] - tt[r], hw_, lw_);
        const f2 lr = splat(mec * (float)(4 * NT * r));
        t.xy = fma2(e2, t.xy, y[r].xy - y02) + (l01 + lr);
        t.zw = fma2(e2, t.zw, y[r].zw - y02) + (l23 + lr);
        *reinterpret_cast<f4*>(&S.X[4 * (tid + NT * r)]) = t;
      }
    };
    for (int i = tid; i < pad; i += NT) S.X[i - pad] = 0.f;   // the gap (dead mask words) becomes Dp[i < 0] = 0
    if (tid < 64) S.X[Lp + tid] = 0.f;
    store_dp();
    LDSP_BAR_CZ();
    STAMP(16); DSTOP(16); LDSP_PROBE(); LDSP_PROBE();
    f2 ac[SP / 2];
    // ---- flat top + last tap (LS).  The last tap multiplies y'[k] = Dp'[k] + (y0 - cpiv) - eps T'[k]; the host admits this kernel
    // only where eps * |w_last| * (rail * L) is far below the columns' resolution (dsp_icpc sets tau = 1e7 us "to switch off CR",
    // src/dsp_icpc.jl:98: eps = 1.6e-9) and the third term is dropped.
    auto flat_top = [&]() {
      f2 wl = splat(Z.w_last), sc = splat(Z.sc);
      const float yc = y0 - cpiv;
      f2 wlc = splat(Z.w_last * yc);
      if (LDSP_L3_SGC_VGPR) { pin(wl); pin(sc); pin(wlc); }
      const float *dk = &S.X[tid], *dpa = &S.X[tid + Lf - 1 - lt], *dpb = &S.X[tid + Lf - 1 - f1];
#pragma unroll
      for (int m = 0; m < SP; m += 2) {
        ac[m / 2] = splat(0.f);
        if (NT * m >= nout) continue;   // row pair beyond the output range (block-uniform)
        const f2 yk = rd2(dk, m), pa = rd2(dpa, m), pb = rd2(dpb, m);
        LDSP_LDS_WAIT_ALL();
        ac[m / 2] = fma2(sc, pa - pb, fma2(wl, yk, wlc));
        pin(ac[m / 2]);
        if ((m & 2) == 2) __builtin_amdgcn_sched_barrier(0);
      }
    };
    f2 dz[SP / 2];   // ZAC - CUSP: the parabola part (the difference of the last taps is folded into it)
    if constexpr (WZ) {
      // ---- u[n] = sum_e zc_r[e] (Dp[n - s_e] - Dp[n - s_{e+1}])  (LS) -> X in place of Dp
      {
        f2 u[SP / 2];
#pragma unroll
        for (int m = 0; m < SP / 2; ++m) u[m] = splat(0.f);
        // the chain of shifts s_0 < s_1 < ..: every shift read once, u += R_e (Dp[n - s_e] - Dp[n - s_{e+1}]).  When CUSP shares the pass
        // (its last tap is the one the flat-top step applies) the chain has the parabola's last tap folded in (icpc_dev.hpp: zf_*)
        const bool fold = WC && ZZ.zf_n > 0;
        const int nch = fold ? ZZ.zf_n : ZZ.zc_n;
        const int32_t* cs = fold ? ZZ.zf_s : ZZ.zc_s;
        const float* cr = fold ? ZZ.zf_r : ZZ.zc_r;
        // (two halves of the rows in turn: half the chain's registers — previous reads, reads in flight — at a time)
#ifndef LDSP_L3_UHALVES
#define LDSP_L3_UHALVES 2   // the rows in two halves: half the chain's registers at a time (1: one pass, 15 registers spill)
#endif
#pragma unroll
        for (int hf = 0; hf < LDSP_L3_UHALVES; ++hf) {
          constexpr int HP = SP / (2 * LDSP_L3_UHALVES);   // row pairs per part
          f2 prev[HP];
          {
            const float* dp = &S.X[tid - cs[0]];
#pragma unroll
            for (int m = 0; m < HP; ++m) prev[m] = rd2(dp, 2 * (HP * hf + m));
          }
          auto link = [&](int e) {
            f2 ce = splat(cr[e]);
            if (LDSP_L3_SGC_VGPR) pin(ce);
            const float* dq = &S.X[tid - cs[e + 1]];
            f2 cur[HP];
#pragma unroll
            for (int m = 0; m < HP; ++m) cur[m] = rd2(dq, 2 * (HP * hf + m));
            LDSP_LDS_WAIT_ALL();
#pragma unroll
            for (int m = 0; m < HP; ++m) {
              u[HP * hf + m] = fma2(ce, prev[m] - cur[m], u[HP * hf + m]);
              prev[m] = cur[m];
            }
          };
          if (STUDY || nch == 9) {   // (block-uniform) the usual tap structure: links unrolled
#pragma unroll
            for (int e = 0; e < 8; ++e) link(e);
          } else {
            for (int e = 0; e + 1 < nch; ++e) link(e);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        flat_top();   // (after the chain: its sixteen accumulators are not alive across it)
        LDSP_BAR_CZ();   // every read of Dp is done
#pragma unroll
        for (int m = 0; m < SP; m += 2) wr2(&S.X[tid], m, u[m / 2]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the compiler does not count the stores of an asm statement)
      }
      LDSP_BAR_CZ();
      STAMP(17); DSTOP(17); LDSP_PROBE(); LDSP_PROBE();
      // ---- PRF = cumsum(cumsum(u)) (S4).  Two levels: inside a wave-row (256 samples) the single and double running sums l1, l2
      // start from zero and stay in float; the state entering each wave-row, (C1, C2), is carried in double:
      //   c2[j] = C2 + (j+1)*C1 + l2[j],  C1' = C1 + l1[255],  C2' = C2 + 256*C1 + l2[255].   Each thread rewrites its own quads.
      {
        float ex1[R], ex2[R], i1[R], i2[R], p3[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const f4 uq = *reinterpret_cast<const f4*>(&S.X[4 * (tid + NT * r)]);
          const float q0 = uq.x, q1_ = q0 + uq.y, q2_ = q1_ + uq.z;
          p3[r] = q2_ + uq.w;
          i1[r] = p3[r];
          i2[r] = (q0 + q1_) + (q2_ + p3[r]);   // the quad's own contribution to the double sum
        }
#ifdef LDSP_WHATIF_S16
        LDSP_DPP_GROUP1("v_add_f32_dpp", i1[0]);
#else
        LDSP_DPP_GROUP4("v_add_f32_dpp", i1[0], "v_add_f32_dpp", i1[1], "v_add_f32_dpp", i1[2], "v_add_f32_dpp", i1[3]);
#endif
#pragma unroll
        for (int r = 0; r < R; ++r) { ex1[r] = i1[r] - p3[r]; i2[r] = fmaf(4.f, ex1[r], i2[r]); ex2[r] = i2[r]; }
#ifdef LDSP_WHATIF_S16
        LDSP_DPP_GROUP1("v_add_f32_dpp", i2[0]);
#else
        LDSP_DPP_GROUP4("v_add_f32_dpp", i2[0], "v_add_f32_dpp", i2[1], "v_add_f32_dpp", i2[2], "v_add_f32_dpp", i2[3]);
#endif
#pragma unroll
        for (int r = 0; r < R; ++r) ex2[r] = i2[r] - ex2[r];
        double* pa = S.dpart; double* pb = S.dpart + R * NW;
        if (lane == 63) {
#pragma unroll
          for (int r = 0; r < R; ++r) { pa[r * NW + wave] = (double)i1[r]; pb[r * NW + wave] = (double)i2[r]; }
        }
        LDSP_BAR_CZ();
        const double t1 = (lane < R * NW) ? pa[lane] : 0.0;
        const double c1x = wave_incl_scan_sum_f64(t1) - t1;
        const double t2 = (lane < R * NW) ? pb[lane] + 256.0 * c1x : 0.0;
        const double c2x = wave_incl_scan_sum_f64(t2) - t2;
        const float mrho = -ZZ.rho_sc;
        const double jd0 = (double)(4 * lane + 1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const double C1 = readlane_d(c1x, r * NW + wave), C2 = readlane_d(c2x, r * NW + wave);
          const double t = fma(C1, jd0, C2);
          const float th = (float)t, tl = (float)(t - (double)th), c1f = (float)C1;
          const f4 uq = *reinterpret_cast<const f4*>(&S.X[4 * (tid + NT * r)]);
          float l1 = ex1[r], l2 = ex2[r];
          f2 la, lb;
          l1 += uq.x; l2 += l1; la.x = l2;
          l1 += uq.y; l2 += l1; la.y = l2;
          l1 += uq.z; l2 += l1; lb.x = l2;
          l1 += uq.w; l2 += l1; lb.y = l2;
          const f2 c2 = splat(c1f), t2_ = splat(tl), h2 = splat(th), m2 = splat(mrho);
          const f2 va = m2 * (h2 + (t2_ + fma2(c2, mk2(0.f, 1.f), la))), vb = m2 * (h2 + (t2_ + fma2(c2, mk2(2.f, 3.f), lb)));
          *reinterpret_cast<f4*>(&S.X[4 * (tid + NT * r)]) = (f4){va.x, va.y, vb.x, vb.y};
        }
      }
      LDSP_BAR_CZ();
      {
        const float* pr = &S.X[tid + Lf - 1];
#pragma unroll
        for (int m = 0; m < SP; m += 2) { dz[m / 2] = splat(0.f); if (NT * m >= nout) continue; dz[m / 2] = rd2(pr, m); pin(dz[m / 2]); }
      }
      STAMP(18); DSTOP(18);
    } else {
      flat_top();
#pragma unroll
      for (int m = 0; m < SP / 2; ++m) dz[m] = splat(0.f);
    }
    // ---- d[i] = (y[i]-y[i-1]) + eps*y[i-1] for 1 <= i < L, else 0   (S4)
    f2 d[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      // (element by element: the odd-aligned pair (y1, y2) of a packed form is one the Savitzky-Golay pass also builds, and hipcc
      // then keeps that pass's copies alive — in scratch — until here)
      if constexpr (!FULL)   // (keeps hipcc from carrying the sample differences of the Savitzky-Golay pass — in scratch — to this point)
        asm volatile("" : "+v"(y[r].x), "+v"(y[r].y), "+v"(y[r].z), "+v"(y[r].w));
      const float yb = y_before(r);
      d[r][0].x = fmaf(Z.eps, yb, (y[r].x - yb) + mec);
      d[r][0].y = fmaf(Z.eps, y[r].x, (y[r].y - y[r].x) + mec);
      d[r][1].x = fmaf(Z.eps, y[r].y, (y[r].z - y[r].y) + mec);
      d[r][1].y = fmaf(Z.eps, y[r].z, (y[r].w - y[r].z) + mec);
      if (r == 0 && tid == 0) d[0][0].x = 0.f;
      if (!FULL && 4 * NT * (r + 1) > L) {   // (block-uniform) a row that reaches beyond the trace: d = 0 there
        const int nv = L - 4 * (opaque(tid) + NT * r);   // samples of the trace in this quad
        if (nv <= 0) { d[r][0] = splat(0.f); d[r][1] = splat(0.f); }
        else if (nv < 4) { if (nv < 2) d[r][0].y = 0.f; if (nv < 3) d[r][1].x = 0.f; d[r][1].y = 0.f; }
      }
    }
    STAMP(19); DSTOP(19);
    const float q1 = Z.qp1[1];
    // ---- causal one-pole G -> X, rise(-) and fall(+) exponentials
    {
      float b[R], s_in[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float gq = d[r][0].x;
        gq = fmaf(q1, gq, d[r][0].y);
        gq = fmaf(q1, gq, d[r][1].x);
        gq = fmaf(q1, gq, d[r][1].y);
        b[r] = gq;
      }
      if constexpr (SEP) s4_exscan_affine_fwd<NT, R>(b, s_in, Z.qp4, Z.qpw, S.part, opaque(tid) >> 6);   // (a wave index of its own: the two passes' lane selects are not shared — through scratch)
      else s4_exscan_affine_fwd<NT, R>(b, s_in, Z.qp4, Z.qpw, S.part);   // (the wave index as a scalar argument, or the four chains in one statement: five / six registers to scratch, -1.6 % / -3 %)   // barrier inside: the LS reads of Dp are done
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float g0 = fmaf(q1, s_in[r], d[r][0].x), g1 = fmaf(q1, g0, d[r][0].y), g2 = fmaf(q1, g1, d[r][1].x), g3 = fmaf(q1, g2, d[r][1].y);
        *reinterpret_cast<f4*>(&S.X[4 * (tid + NT * r)]) = (f4){g0, g1, g2, g3};
      }
    }
    LDSP_BAR_CZ();
    {
      const float *gn = &S.X[tid + Lf - 1], *gnl = &S.X[tid + Lf - 1 - lt], *gk = &S.X[tid], *gkl = &S.X[tid + ltp - 1];
      f2 qlt = splat(Z.q_lt), qml = splat(Z.q_mltp), ql1 = splat(Z.q_ltp1), sh = splat(Z.sc_half_den);
      if (LDSP_L3_SGC_VGPR) { pin(qlt); pin(qml); pin(ql1); pin(sh); }   // (vector registers: no scalar source in the loop's instructions)
#pragma unroll
      for (int m = 0; m < SP; m += 2) {
        if (NT * m >= nout) continue;
        const f2 g_n = rd2(gn, m), g_nl = rd2(gnl, m), g_kl = rd2(gkl, m), g_k = rd2(gk, m);
        LDSP_LDS_WAIT_ALL();
        const f2 pm = g_n - qlt * g_nl;
        const f2 fp = qml * (g_kl - ql1 * g_k);
        ac[m / 2] = fma2(sh, fp - pm, ac[m / 2]);
        pin(ac[m / 2]);
        if ((m & 2) == 2) __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(20); DSTOP(20); LDSP_PROBE(); LDSP_PROBE();
    // ---- anti-causal one-pole A -> X, rise(+) and fall(-) exponentials
    {
      float b[R], s_in[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float a = d[r][1].y;
        a = fmaf(q1, a, d[r][1].x);
        a = fmaf(q1, a, d[r][0].y);
        a = fmaf(q1, a, d[r][0].x);
        b[r] = a;
      }
      if constexpr (SEP) s4_exscan_affine_bwd<NT, R>(b, s_in, Z.qp4, Z.qpw, S.part + R * NW, opaque(tid) >> 6);
      else s4_exscan_affine_bwd<NT, R>(b, s_in, Z.qp4, Z.qpw, S.part + R * NW);   // barrier inside: the LS reads of G are done
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float a3 = fmaf(q1, s_in[r], d[r][1].y), a2 = fmaf(q1, a3, d[r][1].x), a1 = fmaf(q1, a2, d[r][0].y), a0 = fmaf(q1, a1, d[r][0].x);
        *reinterpret_cast<f4*>(&S.X[4 * (tid + NT * r)]) = (f4){a0, a1, a2, a3};
      }
      if (tid == 0) S.X[Lp] = 0.f;   // A[L]
    }
    LDSP_BAR_CZ();
    {
      const float *a1 = &S.X[tid + Lf - lt], *a2 = &S.X[tid + Lf], *a3 = &S.X[tid + 1], *a4 = &S.X[tid + ltp];
      f2 qm1 = splat(Z.q_mlt1), qq1 = splat(Z.q1), qq2 = splat(Z.q2), ql1 = splat(Z.q_ltp1), sh = splat(Z.sc_half_den);
      if (LDSP_L3_SGC_VGPR) { pin(qm1); pin(qq1); pin(qq2); pin(ql1); pin(sh); }
#pragma unroll
      for (int m = 0; m < SP; m += 2) {
        if (NT * m >= nout) continue;
        const f2 a_1 = rd2(a1, m), a_2 = rd2(a2, m), a_3 = rd2(a3, m), a_4 = rd2(a4, m);
        LDSP_LDS_WAIT_ALL();
        const f2 pp = qm1 * a_1 - qq1 * a_2;
        const f2 fm = qq2 * (a_3 - ql1 * a_4);
        ac[m / 2] = fma2(sh, pp - fm, ac[m / 2]);
        if constexpr (WZ) dz[m / 2] += ac[m / 2];   // the ZAC output
        pin(ac[m / 2]);
        if ((m & 2) == 2) __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(21); DSTOP(21); LDSP_PROBE(); LDSP_PROBE();
    // ---- extremestats + SignalEstimator of both outputs (dsp_icpc.jl:170-171,177-178): value first, then its first index
    const int tqf = opaque(tid);
    float mxc = -INFINITY, mxz = -INFINITY;
    float pc = 0.f, pz_ = 0.f;
    float own_c, own_z;
    {
#pragma unroll
      for (int m = 0; m < SP; m += 2) {
        f2 a = ac[m / 2], z = dz[m / 2];
        if (NT * (m + 2) > nout) {
          const bool in0 = tqf + NT * m < nout, in1 = tqf + NT * (m + 1) < nout;
          a.x = in0 ? a.x : -INFINITY; a.y = in1 ? a.y : -INFINITY; z.x = in0 ? z.x : -INFINITY; z.y = in1 ? z.y : -INFINITY;
        }
        mxc = vmax3(mxc, a.x, a.y); mxz = vmax3(mxz, z.x, z.y);
      }
      own_c = mxc; own_z = mxz;
      if (nout >= P.sig_est.npts) {
        const f4 ew = (f4){S.misc[8], S.misc[9], S.misc[10], S.misc[11]};   // i0 (bits) and u of the CUSP and of the ZAC estimate
        const int i0c = __float_as_int(ew.x), i0z = __float_as_int(ew.z);
        const int msc = (i0c - tqf + NT - 1) / NT, msz = (i0z - tqf + NT - 1) / NT;   // smallest m with tid + NT*m >= i0
        const int lc = tqf + NT * msc - i0c, lz = tqf + NT * msz - i0z;
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
#if LDSP_L3_BFLY_SUMS
      LDSP_BFLY2("v_add_f32", "v_add_f32_dpp", pc, pz_);      // totals in lanes 31 (CUSP) / 63 (ZAC)
      LDSP_BFLY2("v_max_f32", "v_max_f32_dpp", mxc, mxz);
      if ((lane & 31) == 31) {
        const bool zz = lane == 63;
        if (zz ? WZ : WC) { S.wsum[(W_CZ + (zz ? 1 : 0)) * NW + wave] = pc; atomicMax(&S.sl->fmx[zz ? FX_ZAC : FX_CUSP], ford(mxc)); }
      }
#else
      LDSP_DPP_GROUP4("v_add_f32_dpp", pc, "v_add_f32_dpp", pz_, "v_max_f32_dpp", mxc, "v_max_f32_dpp", mxz);
      if (lane == 63) {
        if (WC) { S.wsum[(W_CZ + 0) * NW + wave] = pc; atomicMax(&S.sl->fmx[FX_CUSP], ford(mxc)); }
        if (WZ) { S.wsum[(W_CZ + 1) * NW + wave] = pz_; atomicMax(&S.sl->fmx[FX_ZAC], ford(mxz)); }
      }
#endif
    }
    LDSP_BAR_CZ();
    {
      const float vc = ford_inv(S.sl->fmx[FX_CUSP]), vz = ford_inv(S.sl->fmx[FX_ZAC]);
      if (__ballot((WC && own_c == vc) || (WZ && own_z == vz)) != 0ull) {   // only the waves that hold a maximum look its index up
        int bc = 0x7fffffff, bz = 0x7fffffff;
#pragma unroll
        for (int m = SP - 1; m >= 0; --m) {   // findmax: first occurrence
          const float a = (m & 1) ? ac[m / 2].y : ac[m / 2].x, z = (m & 1) ? dz[m / 2].y : dz[m / 2].x;
          const bool in = tqf + NT * m < nout;
          bc = (in && a == vc) ? tqf + NT * m : bc;
          bz = (in && z == vz) ? tqf + NT * m : bz;
        }
        if (WC && bc != 0x7fffffff) atomicMin(&S.sl->imin[IM_CUSP], bc);
        if (WZ && bz != 0x7fffffff) atomicMin(&S.sl->imin[IM_ZAC], bz);
      }
    }
    STAMP(22); DSTOP(22);
    LDSP_BAR_CZ();
    if (tid < 2 && (tid == 0 ? WC : WZ)) {
      const int f = tid;
      float s = 0.f;
      for (int ww = 0; ww < NW; ++ww) s += S.wsum[(W_CZ + f) * NW + ww];
      const float v = ford_inv(S.sl->fmx[f ? FX_ZAC : FX_CUSP]);
      const int i = S.sl->imin[f ? IM_ZAC : IM_CUSP];
      double back = (double)cpiv * (f ? ZZ.hsum : Z.hsum);   // the pivot's share of the output (estimator weights sum to one)
      if (f && WC && ZZ.zf_n > 0) back += (double)(ZZ.w_last - Z.w_last) * (double)(y0 - cpiv);   // ... and the folded last tap's constant
      S.outv[f ? C_e_zac : C_e_cusp] = (nout >= P.sig_est.npts) ? (float)((double)s + back) : NAN;
      S.outv[f ? C_e_zac_max : C_e_cusp_max] = (float)((double)v + back);
      S.outv[f ? C_t_zac_max : C_t_cusp_max] = P.t_first + P.dt * (float)(i + Lf - 1);
    }
  };
  using T_ = std::true_type; using F_ = std::false_type;
  if (skip_cz) {
    // the windowed traces of dsp_icpc_compressed (dsp_icpc.jl:352-353 take no CUSP / ZAC column from them): the stage is left out
    if (tid < 6) S.outv[tid < 3 ? (tid == 0 ? C_e_cusp : tid == 1 ? C_e_cusp_max : C_t_cusp_max) : (tid == 3 ? C_e_zac : tid == 4 ? C_e_zac_max : C_t_zac_max)] = NAN;
  } else if constexpr (!SEP) {
    cz_pass(T_{}, T_{}, P.cusp, P.zac, S.misc[12]);
  } else {
    const float cp_c = S.misc[12], cp_z = S.misc[13];
    cz_pass(T_{}, F_{}, P.cusp, P.zac, cp_c);
    LDSP_BAR_CZ();   // every read of A is done
    rebuild_y();
    cz_pass(F_{}, T_{}, P.zac, P.zac, cp_z);
  }
  STAMP(23); DSTOP(23);
  // ------------------------------------------------------------------------------------------------ outputs
  LDSP_BAR_MAIN();   // the output row was filled by lanes of different waves
  static_assert(C_NCOLS <= 64, "the output row is stored by wave 0");
  if (tid < C_NCOLS) {
    float* dst = reinterpret_cast<float*>(out.col[tid]);
    if (dst) dst[(size_t)blockIdx.x * (size_t)out.stride] = S.outv[tid];
  }
}

template <int NT, int M, bool SEP, bool FULL>
static hipError_t launch_t(const float* wf, int64_t n, const IcpcDev* dP, const IcpcOutDev& out, const float* ext_bl, float ext_bl_scale,
                           int Lf, bool skip_cz, hipStream_t st) {
  const size_t smem = Smem<NT>::bytes(cz_pad_floats(Lf)) + (size_t)g_dbg_lds_pad;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&icpc_lean3_kernel<NT, M, SEP, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((icpc_lean3_kernel<NT, M, SEP, FULL>), dim3((unsigned)n), dim3(NT), smem, st, wf, dP, out, ext_bl, ext_bl_scale, (int)skip_cz);
  return hipGetLastError();
}

}  // namespace lean3

// LDS bytes of the kernel for a tile of NT threads and CUSP/ZAC filters of Lf taps
size_t icpc_lean3_smem_bytes(int NT, int Lf) {
  switch (NT) {
    case 64: return lean3::Smem<64>::bytes(lean3::cz_pad_floats(Lf));
    case 128: return lean3::Smem<128>::bytes(lean3::cz_pad_floats(Lf));
    case 256: return lean3::Smem<256>::bytes(lean3::cz_pad_floats(Lf));
    case 512: return lean3::Smem<512>::bytes(lean3::cz_pad_floats(Lf));
    default: return (size_t)-1;
  }
}

// sg_slots: 7, 13 or 25 (the smallest that holds the main Savitzky-Golay window; the two fixed ones have at most 13 taps)
// full: the traces fill the tile (L = 16 NT); otherwise shorter traces (more than half the tile)
hipError_t launch_icpc_lean3(const float* wf, int64_t n, int NT, int sg_slots, bool cz_shared, bool full, const IcpcDev* dP, const IcpcOutDev& out,
                             const float* ext_bl, float ext_bl_scale, int Lf, bool skip_cz, hipStream_t st) {
#ifdef LDSP_DEV_512
#define LDSP_LEAN_CASES LDSP_CASE(512)
#else
#define LDSP_LEAN_CASES LDSP_CASE(64) LDSP_CASE(128) LDSP_CASE(256) LDSP_CASE(512)
#endif
#define LDSP_ARGS wf, n, dP, out, ext_bl, ext_bl_scale, Lf, skip_cz, st
#define LDSP_CASE(N) \
  case N: \
    if (!full) return cz_shared ? (sg_slots <= 7 ? lean3::launch_t<N, 7, false, false>(LDSP_ARGS) : sg_slots <= 13 ? lean3::launch_t<N, 13, false, false>(LDSP_ARGS) : lean3::launch_t<N, 25, false, false>(LDSP_ARGS)) \
                                : (sg_slots <= 7 ? lean3::launch_t<N, 7, true, false>(LDSP_ARGS) : sg_slots <= 13 ? lean3::launch_t<N, 13, true, false>(LDSP_ARGS) : lean3::launch_t<N, 25, true, false>(LDSP_ARGS)); \
    return cz_shared ? (sg_slots <= 7 ? lean3::launch_t<N, 7, false, true>(LDSP_ARGS) : sg_slots <= 13 ? lean3::launch_t<N, 13, false, true>(LDSP_ARGS) : lean3::launch_t<N, 25, false, true>(LDSP_ARGS)) \
                     : (sg_slots <= 7 ? lean3::launch_t<N, 7, true, true>(LDSP_ARGS) : sg_slots <= 13 ? lean3::launch_t<N, 13, true, true>(LDSP_ARGS) : lean3::launch_t<N, 25, true, true>(LDSP_ARGS));
  switch (NT) {
    LDSP_LEAN_CASES
    default: return hipErrorInvalidValue;
  }
#undef LDSP_CASE
#undef LDSP_ARGS
}

}  // namespace ldsp
