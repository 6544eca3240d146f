// icpc_lean.hip — BASELINE config 2's kernel, pz_trap_lean_kernel: blmean -> shift -> InvCR -> Trap(10 us, 4 us) -> maximum
// (reference src/dsp_icpc.jl:102-105,119-120,147-148) for the standard geometry (the trace fills the tile, L = 16 NT).
// Round 2's fused dsp_icpc kernel of this file (icpc_lean_kernel, two trace-sized LDS arrays, two workgroups per CU) was replaced
// in round 3 by icpc_lean3.hip (one array, three workgroups per CU, fewer exchanges; same throughput); what
// stays here is the sub-chain kernel, which is bound by its ~400 instructions per wave and was measured FASTER in this form
// (three exchanges of partial sums, six barriers: 170 M waveforms/s) than with icpc_lean3's single exchange (146 M: the merged
// prefix scans cost ~100 instructions more).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include <stddef.h>
#include "icpc_dev.hpp"
#include "ldsp_device.hpp"

namespace ldsp {
int g_dbg_lds_pad = 0;   // option "dbg_lds_pad": extra dynamic LDS per workgroup of icpc_lean3_kernel (occupancy experiments, tools/occ_probe.py)
namespace lean {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int R = 4, SP = 16;

__device__ __forceinline__ f2 mk2(float a, float b) { f2 v; v.x = a; v.y = b; return v; }
__device__ __forceinline__ f2 splat(float a) { return mk2(a, a); }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float hsum(f2 v) { return v.x + v.y; }


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


// ---------------------------------------------------------------------------------------------------------------------
// BASELINE config 2: blmean -> shift -> InvCR -> Trap(10 us, 4 us) -> maximum (reference src/dsp_icpc.jl:102-105,119-120,
// 147-148).  blmean comes out bit-identical to the fused chain's column, e_10410 to the rounding of T (the fused chain sums
// the same partial sums in another order: tests/test_baseline_sizes_gpu.py).  One trace-sized LDS array (T): four workgroups per CU.
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


}  // namespace lean

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

}  // namespace ldsp
