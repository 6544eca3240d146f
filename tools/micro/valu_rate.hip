// valu_rate.hip — issue rate of the instruction kinds the fused kernels are made of, on gfx950, at 1 / 2 / 4 waves per SIMD.
// Every test is a loop over a block of 32 independent instructions of one kind (inline asm, nothing for hipcc to fold);
// reported: shader cycles per wave-instruction per SIMD (s_memtime around the loop, median wave), i.e. 2.0 = the
// SIMD-32 rate of a wave64 instruction.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
  X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

typedef float float2_t __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k(const float* in, float* out, long long* cyc, int iters) {
  __shared__ float lds[8192 + 64];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192 + 64; i += 256) lds[i] = in[i & 255];
  __syncthreads();
  float a[32];
  float2_t p[16];
  double d[16];
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = in[(tid + i) & 255];
#pragma unroll
  for (int i = 0; i < 16; ++i) { p[i].x = a[2 * i]; p[i].y = a[2 * i + 1]; d[i] = (double)a[i]; }
  float b = in[tid & 255] * 1e-3f, c = in[(tid + 7) & 255] * 1e-3f;
  float2_t pb = {b, c}, pc = {c, b};
  double db = b, dc = c;
  const float* lp = &lds[tid];
  unsigned long long bal = 0;
  unsigned sc32 = 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      REP32(X)
#undef X
    } else if constexpr (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pb), "v"(pc));
      REP16(X) REP16(X)
#undef X
    } else if constexpr (KIND == 2) {
#define X(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pb));
      REP16(X) REP16(X)
#undef X
    } else if constexpr (KIND == 3) {
#define X(i) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      REP32(X)
#undef X
    } else if constexpr (KIND == 4) {   // independent DPP adds (row_shr:1), no nops
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
      REP32(X)
#undef X
    } else if constexpr (KIND == 5) {   // dependent DPP scan chain as the kernels issue it (s_nop 1 between steps)
#define X(i) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[0]));
      REP32(X)
#undef X
    } else if constexpr (KIND == 6) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(db), "v"(dc));
      REP16(X) REP16(X)
#undef X
    } else if constexpr (KIND == 7) {
#define X(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
      REP32(X)
#undef X
    } else if constexpr (KIND == 8) {   // compare + ballot into SGPR pair (v_cmp_ge_f32 writes an SGPR pair)
#define X(i) { unsigned long long m; asm volatile("v_cmp_ge_f32 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(b)); bal ^= m; }
      REP32(X)
#undef X
    } else if constexpr (KIND == 9) {   // ds_read_b32, 32 in flight, one wait
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[i]) : "v"((unsigned)(size_t)lp), "i"(i * 1024));
      REP32(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == 10) {  // ds_read_b64 (aligned), 16 in flight = the same 32 dwords
      const float* lp2 = &lds[2 * tid];
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(p[i]) : "v"((unsigned)(size_t)lp2), "i"(i * 2048));
      REP16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == 11) {  // ds_read2st64_b32: two rows 256 B apart per instruction
#define X(i) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(p[i]) : "v"((unsigned)(size_t)lp), "i"(2 * i), "i"(2 * i + 1));
      REP16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == 12) {  // ds_read_b128 aligned, 8 in flight = 32 dwords
      const float* lp4 = &lds[4 * tid];
      typedef float float4_t __attribute__((ext_vector_type(4)));
      float4_t q[8];
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i & 7]) : "v"((unsigned)(size_t)lp4), "i"((i & 7) * 4096));
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] += q[i].x + q[i].w;
    } else if constexpr (KIND == 13) {  // v_cvt_f64_f32 + v_add_f64 pair (the double running sums)
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1\n\tv_add_f64 %0, %0, %2" : "=&v"(d[i]) : "v"(a[i]), "v"(db));
      REP16(X)
#undef X
    } else if constexpr (KIND == 14) {  // v_cndmask with vcc
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
      REP32(X)
#undef X
    } else if constexpr (KIND == 15) {  // s_ ops: scalar issue beside nothing
#define X(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc32));
      REP32(X)
#undef X
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += p[i].x + p[i].y + (float)d[i];
  out[blockIdx.x * 256 + tid] = s + (float)(bal & 1) + (float)sc32;
  if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int ninstr, const float* in, float* out, long long* cyc) {
  const int iters = 2000;
  for (int wps : {1, 2, 4}) {        // waves per SIMD = 256-thread blocks per CU
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    // per SIMD: wps waves each issuing iters*ninstr instructions in `med` cycles
    printf("%-28s wps=%d  cyc/instr/SIMD = %6.2f   (wave alone: %6.2f)  wall %.3f ms\n", name, wps, med / ((double)iters * ninstr * wps),
           med / ((double)iters * ninstr), ms);
  }
}

int main() {
  float *in, *out; long long* cyc;
  hipMalloc(&in, 256 * 4); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 4 * 8);
  std::vector<float> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 1.f + 0.37f * (float)((i * 7919) % 101);
  hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
  run<0>("v_fma_f32", 32, in, out, cyc);
  run<1>("v_pk_fma_f32 (2 lanes)", 32, in, out, cyc);
  run<2>("v_pk_add_f32 (2 lanes)", 32, in, out, cyc);
  run<3>("v_max3_f32", 32, in, out, cyc);
  run<4>("v_add_f32_dpp indep", 32, in, out, cyc);
  run<5>("v_add_f32_dpp chain+nop1", 32, in, out, cyc);
  run<6>("v_fma_f64", 32, in, out, cyc);
  run<7>("v_log_f32", 32, in, out, cyc);
  run<8>("v_cmp_ge_f32 -> sgpr", 32, in, out, cyc);
  run<9>("ds_read_b32 x32", 32, in, out, cyc);
  run<10>("ds_read_b64 x16", 16, in, out, cyc);
  run<11>("ds_read2st64_b32 x16", 16, in, out, cyc);
  run<12>("ds_read_b128 x8", 8, in, out, cyc);
  run<13>("cvt_f64_f32+add_f64 x16", 32, in, out, cyc);
  run<14>("v_cndmask_b32", 32, in, out, cyc);
  run<15>("s_add_u32", 32, in, out, cyc);
  return 0;
}
