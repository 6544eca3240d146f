// S4-view reads of a LINEAR LDS array at an arbitrary dword shift s: thread t reads the four consecutive floats
// a[4t + s .. 4t + s + 3].  Which instruction mix serves them, and at what rate compared with the aligned ds_read_b128 and
// with the lane-strided (LS) ds_read2st64_b32 pairs?  Also: v_fmac_f32 with an SGPR multiplicand vs a VGPR one.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_s4shift_test.hip -o /tmp/lds_s4shift_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// MODE 0: ds_read_b128 (s % 4 == 0 only)   1: 2 x ds_read2_b32 (0,1)(2,3)   2: b32 + b64 + b32 (odd s) / b64 + b64 (even s)
// MODE 3: LS reference, 2 x ds_read2st64_b32 (4 samples of 4 rows)
template <int MODE, bool CHECK>
__global__ void k_rd(int s, int iters, float* out) {
  __shared__ __align__(16) float lds[8192 + 64];
  for (int i = threadIdx.x; i < 8192 + 64; i += blockDim.x) lds[i] = CHECK ? (float)i : (float)(i & 255);
  __syncthreads();
  const unsigned base = (unsigned)(size_t)lds;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  for (int it = 0; it < iters; ++it) {
    float v[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // four quads in flight (rows of 512 quads; the last row stays inside the array for s <= 64)
      const unsigned rq = (r == 3) ? 1408u : 512u * r;   // quad row offsets (in quads)
      if (MODE == 0) {
        const unsigned a = base + 16u * (threadIdx.x + rq) + 4u * s;
        asm volatile("ds_read_b128 %0, %1" : "=v"(*(float4*)v[r]) : "v"(a));
      } else if (MODE == 1) {
        const unsigned a = base + 16u * (threadIdx.x + rq) + 4u * s;
        asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(*(float2*)&v[r][0]) : "v"(a));
        asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=v"(*(float2*)&v[r][2]) : "v"(a));
      } else if (MODE == 2) {
        const unsigned a = base + 16u * (threadIdx.x + rq) + 4u * s;
        if (s & 1) {
          asm volatile("ds_read_b32 %0, %1" : "=v"(v[r][0]) : "v"(a));
          asm volatile("ds_read_b64 %0, %1 offset:4" : "=v"(*(float2*)&v[r][1]) : "v"(a));
          asm volatile("ds_read_b32 %0, %1 offset:12" : "=v"(v[r][3]) : "v"(a));
        } else {
          asm volatile("ds_read_b64 %0, %1" : "=v"(*(float2*)&v[r][0]) : "v"(a));
          asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(*(float2*)&v[r][2]) : "v"(a));
        }
      } else {
        const unsigned a = base + 4u * (threadIdx.x + 1024u * (r & 1) + s) + ((r & 2) ? 256u : 0u);
        asm volatile("ds_read2st64_b32 %0, %1 offset0:0 offset1:8" : "=v"(*(float2*)&v[r][0]) : "v"(a));    // rows m, m+1 (512 floats apart)
        asm volatile("ds_read2st64_b32 %0, %1 offset0:16 offset1:24" : "=v"(*(float2*)&v[r][2]) : "v"(a));  // rows m+2, m+3
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc0 += v[r][0]; acc1 += v[r][1]; acc2 += v[r][2]; acc3 += v[r][3]; }
  }
  if (CHECK) {
    // one more read of row 0 for the check
    float4 q;
    const unsigned a = base + 16u * threadIdx.x + 4u * s;
    if (MODE == 1) {
      asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1\n\tds_read2_b32 %2, %1 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(*(float2*)&q.x), "+v"(const_cast<unsigned&>(a)), "=&v"(*(float2*)&q.z));
    } else {
      q = make_float4(0, 0, 0, 0);
    }
    out[4 * threadIdx.x] = q.x; out[4 * threadIdx.x + 1] = q.y; out[4 * threadIdx.x + 2] = q.z; out[4 * threadIdx.x + 3] = q.w;
  } else {
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
  }
}

// VALU: 16 independent accumulators, fmac with the multiplicand in an SGPR (VOP2 src0) or in a VGPR
template <int MODE>
__global__ void k_valu(float c, int iters, float* out) {
  float a[16];
  const float x = (float)threadIdx.x * 1e-3f;
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (float)i;
  float cv = c + (float)(threadIdx.x & 1) * 1e-9f;   // VGPR copy
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(c), "v"(x));
      else if (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(cv), "v"(x));
      else if (MODE == 2) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
      else if (MODE == 3) asm volatile("v_max3_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(x), "v"(cv));
      else if (MODE == 4) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(x) : );
      else if (MODE == 5) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(a[i]), "v"(x) : "vcc");
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* d;
  hipMalloc(&d, 1 << 24);
  std::vector<float> h(2048);
  for (int s = 0; s < 8; ++s) {
    hipLaunchKernelGGL((k_rd<1, true>), dim3(1), dim3(512), 0, 0, s, 1, d);
    hipMemcpy(h.data(), d, 2048 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 2048; ++i) bad += (h[i] != (float)(i + s));
    printf("shift %d: 2 x ds_read2_b32 adjacent dwords -> %s (first: %g %g %g %g)\n", s, bad ? "WRONG" : "correct", h[0], h[1], h[2], h[3]);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, blocks = 256 * 4;
  const char* nm[4] = {"b128 aligned      ", "2 x read2_b32     ", "b32+b64+b32 / 2b64", "LS 2 x read2st64  "};
  for (int mode = 0; mode < 4; ++mode)
    for (int s = 0; s < 4; ++s) {
      if (mode == 0 && s) continue;
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL((k_rd<0, false>), dim3(blocks), dim3(512), 0, 0, s, iters, d);
        if (mode == 1) hipLaunchKernelGGL((k_rd<1, false>), dim3(blocks), dim3(512), 0, 0, s, iters, d);
        if (mode == 2) hipLaunchKernelGGL((k_rd<2, false>), dim3(blocks), dim3(512), 0, 0, s, iters, d);
        if (mode == 3) hipLaunchKernelGGL((k_rd<3, false>), dim3(blocks), dim3(512), 0, 0, s, iters, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const double bytes = (double)blocks * 512 * iters * 4 * 16;
      printf("%s shift %d: %.3f ms  -> %.1f TB/s LDS aggregate (%.1f B/clk/CU at 2.4 GHz)\n", nm[mode], s, ms, bytes / ms * 1e-9, bytes / ms * 1e3 / 256 / 2.4e9);
    }
  const char* vn[6] = {"v_fmac_f32 sgpr*vgpr", "v_fmac_f32 vgpr*vgpr", "v_max_f32           ", "v_max3_f32          ", "v_cndmask vcc       ", "v_cmp -> vcc        "};
  for (int wpb = 256; wpb <= 512; wpb += 256)
  for (int mode = 0; mode < 6; ++mode) {
    float ms = 0;
    const int vit = 4000, vb = 256 * 3;   // 3 blocks per CU: 3 or 6 waves per SIMD
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL((k_valu<0>), dim3(vb), dim3(wpb), 0, 0, 1.0001f, vit, d);
      if (mode == 1) hipLaunchKernelGGL((k_valu<1>), dim3(vb), dim3(wpb), 0, 0, 1.0001f, vit, d);
      if (mode == 2) hipLaunchKernelGGL((k_valu<2>), dim3(vb), dim3(wpb), 0, 0, 1.0001f, vit, d);
      if (mode == 3) hipLaunchKernelGGL((k_valu<3>), dim3(vb), dim3(wpb), 0, 0, 1.0001f, vit, d);
      if (mode == 4) hipLaunchKernelGGL((k_valu<4>), dim3(vb), dim3(wpb), 0, 0, 1.0001f, vit, d);
      if (mode == 5) hipLaunchKernelGGL((k_valu<5>), dim3(vb), dim3(wpb), 0, 0, 1.0001f, vit, d);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double wave_instr_per_simd = (double)vit * 16 * (wpb / 64) * 3 / 4;
    printf("%s %d waves/SIMD: %.3f ms -> %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", vn[mode], wpb / 64 * 3 / 4, ms, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
  }
  return 0;
}
