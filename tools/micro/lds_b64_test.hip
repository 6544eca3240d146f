// Does gfx950 serve a 4-byte-aligned (not 8-byte-aligned) ds_read_b64, and at what rate?
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_b64_test.hip -o /tmp/lds_b64_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_check(int shift, float* out) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  unsigned addr = (unsigned)(size_t)(lds) + 4u * (2u * threadIdx.x + (unsigned)shift);  // LDS byte address (low 32 bits of the flat ptr are the offset)
  float2 v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  out[2 * threadIdx.x] = v.x;
  out[2 * threadIdx.x + 1] = v.y;
}

template <int MODE>  // 0: b32 pairs, 1: b64
__global__ void k_time(int shift, int iters, float* out) {
  __shared__ float lds[8192 + 64];
  for (int i = threadIdx.x; i < 8192 + 64; i += blockDim.x) lds[i] = (float)(i & 255);
  __syncthreads();
  float acc = 0.f;
  unsigned base = (unsigned)(size_t)(lds);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MODE == 1) {
        unsigned addr = base + 4u * (2u * threadIdx.x + 1024u * (unsigned)(m & 3) + (unsigned)shift);
        float2 v;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += v.x + v.y;
      } else {
        unsigned a0 = base + 4u * (threadIdx.x + 512u * (unsigned)(m & 3) + (unsigned)shift);
        float x, y;
        asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(a0));
        asm volatile("ds_read_b32 %0, %1 offset:8192" : "=v"(y) : "v"(a0));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += x + y;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  float* d;
  hipMalloc(&d, 1 << 24);
  std::vector<float> h(1024);
  for (int s = 0; s < 4; ++s) {
    hipLaunchKernelGGL(k_check, dim3(1), dim3(256), 0, 0, s, d);
    hipMemcpy(h.data(), d, 512 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 512; ++i) bad += (h[i] != (float)(i + s));
    printf("shift %d: ds_read_b64 at 4-byte alignment -> %s (first: %g %g %g %g)\n", s, bad ? "WRONG" : "correct", h[0], h[1], h[2], h[3]);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, blocks = 256 * 4;
  for (int mode = 0; mode < 2; ++mode)
    for (int s = 0; s < 3; ++s) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode) hipLaunchKernelGGL(k_time<1>, dim3(blocks), dim3(512), 0, 0, s, iters, d);
        else hipLaunchKernelGGL(k_time<0>, dim3(blocks), dim3(512), 0, 0, s, iters, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double bytes = (double)blocks * 512 * iters * 8 * 8;
      printf("%s shift %d: %.3f ms  -> %.1f TB/s LDS aggregate\n", mode ? "b64    " : "b32 x 2", s, ms, bytes / ms * 1e-9);
    }
  return 0;
}
