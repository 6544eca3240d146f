// valu_rate3.hip — does a SIMD issue scalar, LDS and vector instructions of DIFFERENT waves in parallel, and what do the
// bookkeeping instructions (s_nop, s_waitcnt, branches, exec save/restore) cost?  Method as valu_rate.hip; the "mix" kernels
// give the waves of one SIMD different instruction kinds (wave parity), so overlap shows as time(mix) < time(a) + time(b).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate3.hip -o tools/micro/valu_rate3
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
  X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

__device__ __forceinline__ void blk_valu(float (&a)[32], float b, float c) {
#define X(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
  REP32(X)
#undef X
}
__device__ __forceinline__ void blk_salu() {
#define X(i) asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
  REP32(X)
#undef X
}
__device__ __forceinline__ void blk_lds(float (&a)[32], unsigned lp) {
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[i]) : "v"(lp), "i"(i * 1024));
  REP32(X)
#undef X
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// MODE: 0 all waves VALU, 1 all SALU, 2 all LDS, 3 even waves VALU / odd SALU, 4 even VALU / odd LDS,
// 5 every wave alternates a VALU block and a SALU block (same-wave interleave), 6 every wave VALU block + LDS block
template <int KIND>
__global__ void __launch_bounds__(256) k(const float* in, float* out, long long* cyc, int iters) {
  __shared__ float lds[8192 + 64];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192 + 64; i += 256) lds[i] = in[i & 255];
  __syncthreads();
  float a[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = in[(tid + i) & 255];
  float b = in[tid & 255] * 1e-3f, c = in[(tid + 7) & 255] * 1e-3f;
  const float sb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b)));
  const unsigned lp = (unsigned)(size_t)&lds[tid & 63];
  // blockIdx parity decides the role, so that the two kinds share SIMDs (blocks of 256 threads = one wave per SIMD each)
  const bool odd = (blockIdx.x & 1) != 0;
  unsigned long long m64 = __ballot(b > c);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) blk_valu(a, b, c);
    else if constexpr (KIND == 1) blk_salu();
    else if constexpr (KIND == 2) blk_lds(a, lp);
    else if constexpr (KIND == 3) { if (odd) blk_salu(); else blk_valu(a, b, c); }
    else if constexpr (KIND == 4) { if (odd) blk_lds(a, lp); else blk_valu(a, b, c); }
    else if constexpr (KIND == 5) { blk_valu(a, b, c); blk_salu(); }
    else if constexpr (KIND == 6) { blk_valu(a, b, c); blk_lds(a, lp); }
    else if constexpr (KIND == 7) {
#define X(i) asm volatile("s_nop 0");
      REP32(X)
#undef X
    } else if constexpr (KIND == 8) {
#define X(i) asm volatile("s_nop 1");
      REP32(X)
#undef X
    } else if constexpr (KIND == 9) {
#define X(i) asm volatile("s_waitcnt lgkmcnt(0)");
      REP32(X)
#undef X
    } else if constexpr (KIND == 10) {  // VOP2 multiply with an SGPR operand
#define X(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));
      REP32(X)
#undef X
    } else if constexpr (KIND == 11) {  // VOP2 fmac with an SGPR operand
#define X(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "s"(sb), "v"(c));
      REP32(X)
#undef X
    } else if constexpr (KIND == 12) {  // inline constant operand
#define X(i) asm volatile("v_fmac_f32_e32 %0, 2.0, %1" : "+v"(a[i]) : "v"(c));
      REP32(X)
#undef X
    } else if constexpr (KIND == 13) {  // literal constant operand
#define X(i) asm volatile("v_fmac_f32_e32 %0, 0x3f9d70a4, %1" : "+v"(a[i]) : "v"(c));
      REP32(X)
#undef X
    } else if constexpr (KIND == 14) {  // exec save / restore around one VALU (divergent if)
#define X(i) asm volatile("s_and_saveexec_b64 s[20:21], %1\n\tv_add_f32_e32 %0, %0, %2\n\ts_or_b64 exec, exec, s[20:21]" : "+v"(a[i]) : "s"(m64), "v"(b) : "s20", "s21", "memory");
      REP16(X)
#undef X
    } else if constexpr (KIND == 15) {  // not-taken scalar branch
#define X(i) asm volatile("s_cmp_eq_u32 %0, 12345\n\ts_cbranch_scc1 1f\n1:" ::"s"(iters) : "scc");
      REP16(X)
#undef X
    } else if constexpr (KIND == 16) {  // v_cmp to vcc alone
#define X(i) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" ::"v"(a[i]), "v"(b) : "vcc");
      REP32(X)
#undef X
    } else if constexpr (KIND == 17) {  // v_pk_mul_f32
      typedef float float2_t __attribute__((ext_vector_type(2)));
      float2_t pb = {b, c};
#define X(i) { float2_t p = {a[2 * i], a[2 * i + 1]}; asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(pb)); a[2 * i] = p.x; a[2 * i + 1] = p.y; }
      REP16(X)
#undef X
    } else if constexpr (KIND == 18) {  // v_writelane
#define X(i) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(a[i]) : "s"(iters));
      REP32(X)
#undef X
    } else if constexpr (KIND == 19) {  // ds_read_b32 then immediately dependent VALU each (8 reads, wait after each pair)
      float q0, q1;
#define X(i) asm volatile("ds_read_b32 %0, %2 offset:%3\n\tds_read_b32 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q0), "=&v"(q1) : "v"(lp), "i"(i * 2048), "i"(i * 2048 + 1024)); a[i] += q0 * q1;
      REP16(X)
#undef X
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += a[i];
  out[blockIdx.x * 256 + tid] = s + lds[(tid * 7) & 8191];
  if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int ninstr, const float* in, float* out, long long* cyc, bool split = false) {
  const int iters = 400;
  for (int wps : {1, 2, 4}) {
    if (split && wps == 1) continue;
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, 5);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    if (split) {   // even / odd blocks separately
      std::vector<long long> ev, od;
      for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) ((b & 1) ? od : ev).push_back(h[b * 4 + w]);
      std::sort(ev.begin(), ev.end()); std::sort(od.begin(), od.end());
      printf("%-40s wps=%d  even-wave cycles %8.0f  odd-wave cycles %8.0f  wall %.3f ms\n", name, wps, (double)ev[ev.size() / 2], (double)od[od.size() / 2], ms);
    } else {
      std::sort(h.begin(), h.end());
      const double med = (double)h[h.size() / 2];
      printf("%-40s wps=%d  cyc/instr/SIMD = %6.2f   (per wave: %6.2f; total %8.0f)  wall %.3f ms\n", name, wps, med / ((double)iters * ninstr * wps), med / ((double)iters * ninstr), med, ms);
    }
    fflush(stdout);
  }
}

int main() {
  float *in, *out; long long* cyc;
  (void)hipMalloc(&in, 256 * 4); (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 4 * 8);
  std::vector<float> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 1.f + 0.37f * (float)((i * 7919) % 101);
  (void)hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
  run<0>("all waves: 32 v_fmac", 32, in, out, cyc);
  run<1>("all waves: 32 s_add", 32, in, out, cyc);
  run<2>("all waves: 32 ds_read_b32", 32, in, out, cyc);
  run<3>("even waves v_fmac / odd waves s_add", 32, in, out, cyc, true);
  run<4>("even waves v_fmac / odd waves ds_read", 32, in, out, cyc, true);
  run<5>("every wave: 32 v_fmac + 32 s_add", 64, in, out, cyc);
  run<6>("every wave: 32 v_fmac + 32 ds_read", 64, in, out, cyc);
  run<7>("s_nop 0", 32, in, out, cyc);
  run<8>("s_nop 1", 32, in, out, cyc);
  run<9>("s_waitcnt lgkmcnt(0) (nothing pending)", 32, in, out, cyc);
  run<10>("v_mul_f32_e32 with SGPR src0", 32, in, out, cyc);
  run<11>("v_fmac_f32_e32 with SGPR src0", 32, in, out, cyc);
  run<12>("v_fmac_f32_e32 inline constant", 32, in, out, cyc);
  run<13>("v_fmac_f32_e32 literal constant", 32, in, out, cyc);
  run<14>("saveexec + v_add + restore (x16 = 48)", 48, in, out, cyc);
  run<15>("s_cmp + not-taken s_cbranch (x16 = 32)", 32, in, out, cyc);
  run<16>("v_cmp_gt_f32_e32 vcc", 32, in, out, cyc);
  run<17>("v_pk_mul_f32", 16, in, out, cyc);
  run<18>("v_writelane_b32", 32, in, out, cyc);
  run<19>("2x ds_read_b32 + wait + fma (x16)", 48, in, out, cyc);
  return 0;
}
