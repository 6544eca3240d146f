// micro-test of the butterfly reductions of wave_prims.hpp (LDSP_BFLY4 / LDSP_BFLY2) against plain maxima / minima (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstring>
#include "../../legenddsp.jl_amd/csrc/wave_prims.hpp"
using namespace ldsp;
__global__ void k(const float* in, float* o4, float* o2, uint32_t* ou) {
  const int l = threadIdx.x;
  float a = in[l], b = in[64 + l], c = in[128 + l], d = in[192 + l];
  LDSP_BFLY4("v_max_f32", "v_max_f32_dpp", a, b, c, d);
  o4[l] = a;
  float e = in[l], f = in[64 + l];
  LDSP_BFLY2("v_min_f32", "v_min_f32_dpp", e, f);
  o2[l] = e;
  uint32_t p = __float_as_uint(fabsf(in[l])), q = __float_as_uint(fabsf(in[64 + l])), r = __float_as_uint(fabsf(in[128 + l])), s = __float_as_uint(fabsf(in[192 + l]));
  LDSP_BFLY4("v_min_u32", "v_min_u32_dpp", p, q, r, s);
  ou[l] = p;
}
int main() {
  float h[256]; for (int i = 0; i < 256; ++i) h[i] = sinf(i * 1.7f + (i / 64)) * 10 + (i % 7) * 0.3f + 1.f;
  float *d, *o4, *o2; uint32_t* ou;
  hipMalloc(&d, 1024); hipMalloc(&o4, 256); hipMalloc(&o2, 256); hipMalloc(&ou, 256);
  hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o4, o2, ou);
  float r4[64], r2[64]; uint32_t ru[64];
  hipMemcpy(r4, o4, 256, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 256, hipMemcpyDeviceToHost); hipMemcpy(ru, ou, 256, hipMemcpyDeviceToHost);
  float mx[4], mn[4]; uint32_t mu[4];
  for (int v = 0; v < 4; ++v) { mx[v] = -1e30f; mn[v] = 1e30f; mu[v] = 0xffffffffu;
    for (int i = 0; i < 64; ++i) { mx[v] = fmaxf(mx[v], h[64 * v + i]); mn[v] = fminf(mn[v], h[64 * v + i]); uint32_t u; float f = fabsf(h[64 * v + i]); memcpy(&u, &f, 4); mu[v] = u < mu[v] ? u : mu[v]; } }
  int bad = 0;
  // BFLY4: lane 15: a, 31: c, 47: b, 63: d
  bad += r4[15] != mx[0]; bad += r4[31] != mx[2]; bad += r4[47] != mx[1]; bad += r4[63] != mx[3];
  bad += ru[15] != mu[0]; bad += ru[31] != mu[2]; bad += ru[47] != mu[1]; bad += ru[63] != mu[3];
  // BFLY2: lane 31: a, lane 63: b
  bad += r2[31] != mn[0]; bad += r2[63] != mn[1];
  printf("max4 %g %g %g %g (want %g %g %g %g)  min2 %g %g (want %g %g)\n", r4[15], r4[47], r4[31], r4[63], mx[0], mx[1], mx[2], mx[3], r2[31], r2[63], mn[0], mn[1]);
  printf(bad ? "BUTTERFLY TEST FAILED (%d)\n" : "BUTTERFLY TEST OK\n", bad);
  return bad != 0;
}
