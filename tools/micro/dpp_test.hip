// micro-test of the DPP wave primitives used by the kernels (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../legenddsp.jl_amd/csrc/wave_prims.hpp"
using namespace ldsp;
__global__ void k(const float* in, float* o_scan, float* o_max, double* o_dscan, float* o_aff, float* o_affr, float q4) {
  int l = threadIdx.x;
  float v = in[l];
  o_scan[l] = wave_incl_scan_sum(v);
  o_max[l] = wave_max_all(v);
  o_dscan[l] = wave_incl_scan_sum_f64((double)v * 1e8);
  AffinePow P = {q4, q4 * q4, powf(q4, 4.f), powf(q4, 8.f)};
  o_aff[l] = wave_incl_scan_affine(v, P, powf(q4, (float)((l & 15) + 1)), powf(q4, (float)((l & 31) + 1)));
  float pw[6] = {q4, powf(q4, 2.f), powf(q4, 4.f), powf(q4, 8.f), powf(q4, 16.f), powf(q4, 32.f)};
  o_affr[l] = wave_incl_scan_affine_rev(v, pw);
}
int main() {
  float h[64]; for (int i = 0; i < 64; ++i) h[i] = sinf(i * 0.7f) * 10 + i * 0.01f;
  float *d, *s, *m, *a, *ar; double* ds;
  hipMalloc(&d, 256); hipMalloc(&s, 256); hipMalloc(&m, 256); hipMalloc(&a, 256); hipMalloc(&ar, 256); hipMalloc(&ds, 512);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  float q4 = 0.9873f;
  k<<<1, 64>>>(d, s, m, ds, a, ar, q4);
  float hs[64], hm[64], ha[64], har[64]; double hd[64];
  hipMemcpy(hs, s, 256, hipMemcpyDeviceToHost); hipMemcpy(hm, m, 256, hipMemcpyDeviceToHost);
  hipMemcpy(ha, a, 256, hipMemcpyDeviceToHost); hipMemcpy(har, ar, 256, hipMemcpyDeviceToHost); hipMemcpy(hd, ds, 512, hipMemcpyDeviceToHost);
  double run = 0, mx = -1e30, aff = 0; int bad = 0;
  double affr[65]; affr[64] = 0; for (int i = 63; i >= 0; --i) affr[i] = h[i] + q4 * affr[i + 1];
  for (int i = 0; i < 64; ++i) mx = fmax(mx, h[i]);
  for (int i = 0; i < 64; ++i) {
    run += h[i]; aff = h[i] + q4 * aff;
    if (fabs(hs[i] - run) > 1e-3 || hm[i] != (float)mx || fabs(hd[i] - run * 1e8) > 1e3 || fabs(ha[i] - aff) > 1e-3 || fabs(har[i] - affr[i]) > 1e-3) {
      ++bad; printf("lane %d scan %g/%g max %g/%g d %g/%g aff %g/%g affr %g/%g\n", i, hs[i], run, hm[i], mx, hd[i], run * 1e8, ha[i], aff, har[i], affr[i]);
    }
  }
  printf(bad ? "DPP TEST FAILED (%d)\n" : "DPP TEST OK\n", bad);
  return bad != 0;
}
