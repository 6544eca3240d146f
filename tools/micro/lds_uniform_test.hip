// How expensive is an LDS read whose 64 lanes all carry the SAME address (a table entry every lane needs)?
// ds_read_b32 / b64 / b128 uniform vs per-lane-consecutive.  Build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_uniform_test.hip -o /tmp/lds_uniform_test
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W, bool UNI>
__global__ void k(int iters, float* out) {
  __shared__ __align__(16) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)(i & 255);
  __syncthreads();
  const unsigned base = (unsigned)(size_t)lds;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    float4 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned a = base + (UNI ? 64u * r + 16u * (it & 7) : (unsigned)(W * 4) * threadIdx.x % 16384u + 4096u * r);
      if (W == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(v[r].x) : "v"(a));
      if (W == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(*(float2*)&v[r]) : "v"(a));
      if (W == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(v[r]) : "v"(a));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 4; ++r) acc += v[r].x;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  float* d; hipMalloc(&d, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, blocks = 256 * 4;
  auto run = [&](auto kern, const char* nm) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, iters, d); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
    const double instr_per_cu = (double)blocks / 256 * 8 * iters * 4;   // wave-instructions per CU
    printf("%-28s %.3f ms -> %.1f cycles per wave-instruction per CU (2.4 GHz)\n", nm, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
  };
  run(k<1, true>, "b32 uniform address"); run(k<2, true>, "b64 uniform address"); run(k<4, true>, "b128 uniform address");
  run(k<1, false>, "b32 consecutive lanes"); run(k<2, false>, "b64 consecutive lanes"); run(k<4, false>, "b128 consecutive lanes");
  return 0;
}
