// LDS ds_add_u32 (no return) throughput vs same-address multiplicity inside a wave.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int ways, int iters, unsigned* out) {
  __shared__ unsigned hist[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // `ways` lanes share one address; distinct groups hit distinct banks
  unsigned idx = (unsigned)((lane / ways) + 64 * (wave & 15));
  for (int it = 0; it < iters; ++it) {
    atomicAdd(&hist[(idx + 7u * it) & 4095u], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = hist[5];
}
int main() {
  unsigned* d; hipMalloc(&d, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, blocks = 256;
  for (int ways : {1, 2, 4, 8, 16, 32, 64}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, ways, iters, d);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // one workgroup of 16 waves per CU: clocks per wave-instruction per CU
    printf("%2d lanes per address: %.3f ms -> %.1f clk per wave-atomic per CU (16 waves, 2.4 GHz)\n", ways, ms, ms * 1e-3 * 2.4e9 / (iters * 16.0));
  }
  return 0;
}
