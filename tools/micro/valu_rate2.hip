// valu_rate2.hip — second round of issue-rate measurements on gfx950: selects, compares, lane reads, exec-mask bookkeeping,
// conversions, LDS stores.  Same method as valu_rate.hip (cycles per wave-instruction per SIMD; 2.0 = the SIMD-32 rate).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate2.hip -o tools/micro/valu_rate2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
  X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ void __launch_bounds__(256) k(const float* in, float* out, long long* cyc, int iters) {
  __shared__ float lds[8192 + 64];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192 + 64; i += 256) lds[i] = in[i & 255];
  __syncthreads();
  float a[32];
  double d[16];
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = in[(tid + i) & 255];
#pragma unroll
  for (int i = 0; i < 16; ++i) d[i] = (double)a[i];
  float b = in[tid & 255] * 1e-3f, c = in[(tid + 7) & 255] * 1e-3f;
  double db = b;
  unsigned long long m64 = __ballot(b > c);
  const unsigned lp = (unsigned)(size_t)&lds[tid];
  const unsigned lp4 = (unsigned)(size_t)&lds[4 * tid];
  int sacc = 0;
  asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" ::"v"(b), "v"(c) : "vcc");
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {         // select on VCC (set before the loop)
#define X(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
      REP32(X)
#undef X
    } else if constexpr (KIND == 1) {  // select on an SGPR pair
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m64));
      REP32(X)
#undef X
    } else if constexpr (KIND == 2) {  // compare to VCC + select: the usual predicated update (16 pairs = 32 instructions)
#define X(i) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      REP16(X)
#undef X
    } else if constexpr (KIND == 3) {
#define X(i) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP32(X)
#undef X
    } else if constexpr (KIND == 4) {
#define X(i) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[i]) : "v"(b));
      REP32(X)
#undef X
    } else if constexpr (KIND == 5) {
#define X(i) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP32(X)
#undef X
    } else if constexpr (KIND == 6) {
#define X(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP32(X)
#undef X
    } else if constexpr (KIND == 7) {  // fma with one SGPR operand
      const float sb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b)));
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sb), "v"(c));
      REP32(X)
#undef X
    } else if constexpr (KIND == 8) {  // v_readlane -> SGPR, no consumer in between
#define X(i) { int s; asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(s) : "v"(a[i])); sacc ^= s; }
      REP32(X)
#undef X
    } else if constexpr (KIND == 9) {  // compare into 16 different SGPR pairs, consumed only after all 16
      unsigned long long m[16];
#define X(i) asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m[i]) : "v"(a[i]), "v"(b));
      REP16(X)
#undef X
      unsigned long long x = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) x ^= m[i];
      sacc ^= (int)x;
    } else if constexpr (KIND == 10) { // exec-mask bookkeeping of a divergent `if`: saveexec + restore around one VALU
#define X(i) asm volatile("s_and_saveexec_b64 s[20:21], %1\n\tv_add_f32_e32 %0, %0, %2\n\ts_or_b64 exec, exec, s[20:21]" : "+v"(a[i]) : "s"(m64), "v"(b) : "s20", "s21");
      REP16(X)
#undef X
    } else if constexpr (KIND == 11) {
#define X(i) asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
      REP16(X)
#undef X
    } else if constexpr (KIND == 12) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
      REP16(X)
#undef X
    } else if constexpr (KIND == 13) {
#define X(i) asm volatile("v_cvt_f32_f64_e32 %0, %1" : "=v"(a[i]) : "v"(d[i]));
      REP16(X)
#undef X
    } else if constexpr (KIND == 14) { // ds_write_b32 x32
#define X(i) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(lp), "v"(a[i]), "i"(i * 1024));
      REP32(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == 15) { // ds_write_b128 x8 (the same 32 dwords)
      typedef float float4_t __attribute__((ext_vector_type(4)));
#define X(i) { float4_t q = {a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]}; asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(lp4), "v"(q), "i"(i * 4096)); }
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == 16) { // scalar ALU
#define X(i) asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
      REP32(X)
#undef X
    } else if constexpr (KIND == 17) { // bitwise select without a mask register: v_bfi_b32
#define X(i) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      REP32(X)
#undef X
    } else if constexpr (KIND == 18) { // v_med3 (clamp)
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP32(X)
#undef X
    } else if constexpr (KIND == 19) { // v_add_u32 / v_lshl_add_u32 (address arithmetic)
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
      REP32(X)
#undef X
    } else if constexpr (KIND == 20) { // LDS read + dependent use: latency of ONE read (chain of 8)
      unsigned p = lp;
#define X(i) asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32_e32 %0, 0x3ffc, %0" : "+v"(p));
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
      sacc ^= (int)p;
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += (float)d[i];
  out[blockIdx.x * 256 + tid] = s + (float)sacc + lds[(tid * 7) & 8191];
  if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int ninstr, const float* in, float* out, long long* cyc) {
  const int iters = 400;
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, 5);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    printf("%-36s wps=%d  cyc/instr/SIMD = %6.2f   (per wave: %6.2f)\n", name, wps, med / ((double)iters * ninstr * wps), med / ((double)iters * ninstr));
    fflush(stdout);
  }
}

int main() {
  float *in, *out; long long* cyc;
  (void)hipMalloc(&in, 256 * 4); (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 4 * 8);
  std::vector<float> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 1.f + 0.37f * (float)((i * 7919) % 101);
  (void)hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
  run<0>("v_cndmask_b32_e32 (vcc)", 32, in, out, cyc);
  run<1>("v_cndmask_b32_e64 (sgpr pair)", 32, in, out, cyc);
  run<2>("v_cmp_e32 vcc + v_cndmask pairs", 32, in, out, cyc);
  run<3>("v_max_f32_e32", 32, in, out, cyc);
  run<4>("v_mov_b32", 32, in, out, cyc);
  run<5>("v_sub_f32_e32", 32, in, out, cyc);
  run<6>("v_fmac_f32_e32", 32, in, out, cyc);
  run<7>("v_fma_f32 (sgpr operand)", 32, in, out, cyc);
  run<8>("v_readlane_b32", 32, in, out, cyc);
  run<9>("v_cmp_e64 -> 16 sgpr pairs, late use", 16, in, out, cyc);
  run<10>("saveexec + v_add + restore (x16)", 48, in, out, cyc);
  run<11>("v_cvt_f64_f32", 16, in, out, cyc);
  run<12>("v_add_f64", 16, in, out, cyc);
  run<13>("v_cvt_f32_f64", 16, in, out, cyc);
  run<14>("ds_write_b32 x32", 32, in, out, cyc);
  run<15>("ds_write_b128 x8", 8, in, out, cyc);
  run<16>("s_add_u32", 32, in, out, cyc);
  run<17>("v_bfi_b32", 32, in, out, cyc);
  run<18>("v_med3_f32", 32, in, out, cyc);
  run<19>("v_lshl_add_u32", 32, in, out, cyc);
  run<20>("ds_read_b32 dependent chain (latency)", 8, in, out, cyc);
  return 0;
}
