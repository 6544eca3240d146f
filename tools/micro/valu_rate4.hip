// valu_rate4.hip — does the issue rate of plain float VALU instructions depend on WHICH registers they use (v0..v63 against
// v64+), on the number of registers the kernel allocates, on independence inside a wave, and on what is mixed in between?
// Background: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU is 0.5 in pz_trap_lean_kernel (40 VGPRs, 8 waves per SIMD) and 1.02 in
// icpc_lean3_kernel (78 VGPRs, 6 waves per SIMD) in every phase (profiles/r03_*).  Method as valu_rate3.hip.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate4.hip -o tools/micro/valu_rate4
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define STR2(x) #x
#define STR(x) STR2(x)
#define R32(X, B) X(B, 0) X(B, 1) X(B, 2) X(B, 3) X(B, 4) X(B, 5) X(B, 6) X(B, 7) X(B, 8) X(B, 9) X(B, 10) X(B, 11) X(B, 12) X(B, 13) X(B, 14) X(B, 15) \
  X(B, 16) X(B, 17) X(B, 18) X(B, 19) X(B, 20) X(B, 21) X(B, 22) X(B, 23) X(B, 24) X(B, 25) X(B, 26) X(B, 27) X(B, 28) X(B, 29) X(B, 30) X(B, 31)

// KIND 0: 32 independent v_fmac on v8..v39        1: the same on v80..v111        2: on v160..v191
//      3: 32 DEPENDENT v_fmac on v8               4: independent v_fmac (low) alternating with v_cmp vcc
//      5: v_add_f32 low, kernel allocates 200 VGPRs (touches v199 once)            6: v_pk_fma_f32 on low pairs
//      7: v_fma_f32 (VOP3) low                    8: v_mov_b32 low                 9: v_cndmask_b32_e32 (vcc) low
template <int KIND>
__global__ void __launch_bounds__(256) k(const float* in, float* out, long long* cyc, int iters) {
  const int tid = threadIdx.x;
  float b = in[tid & 255] * 1e-3f, c = in[(tid + 7) & 255] * 1e-3f;
  asm volatile("v_mov_b32 v1, %0\n\tv_mov_b32 v2, %1\n\tv_mov_b32 v3, %0\n\tv_mov_b32 v4, %1\n\tv_mov_b32 v5, %0" ::"v"(b), "v"(c) : "v1", "v2", "v3", "v4", "v5");
  if (KIND == 5) asm volatile("v_mov_b32 v199, 0" ::: "v199");
  if (KIND == 2) asm volatile("v_mov_b32 v191, 0" ::: "v191");
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define FM(B, i) "v_fmac_f32_e32 v[" STR(B) "+" STR(i) "], v1, v2\n\t"
#define DEP(B, i) "v_fmac_f32_e32 v8, v1, v2\n\t"
#define MIX(B, i) "v_fmac_f32_e32 v[" STR(B) "+" STR(i) "], v1, v2\n\tv_cmp_gt_f32_e32 vcc, v1, v2\n\t"
#define AD(B, i) "v_add_f32_e32 v[" STR(B) "+" STR(i) "], v1, v[" STR(B) "+" STR(i) "]\n\t"
#define F3(B, i) "v_fma_f32 v[" STR(B) "+" STR(i) "], v1, v2, v[" STR(B) "+" STR(i) "]\n\t"
#define MV(B, i) "v_mov_b32_e32 v[" STR(B) "+" STR(i) "], v1\n\t"
#define CN(B, i) "v_cndmask_b32_e32 v[" STR(B) "+" STR(i) "], v1, v2, vcc\n\t"
#define CLOB_LO "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39"
    if constexpr (KIND == 0) asm volatile(R32(FM, 8) ::: CLOB_LO);
    else if constexpr (KIND == 1) asm volatile(R32(FM, 80) ::: "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111");
    else if constexpr (KIND == 2) asm volatile(R32(FM, 160) ::: "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191");
    else if constexpr (KIND == 3) asm volatile(R32(DEP, 8) ::: "v8");
    else if constexpr (KIND == 4) asm volatile(R32(MIX, 8) ::: CLOB_LO, "vcc");
    else if constexpr (KIND == 5) asm volatile(R32(AD, 8) ::: CLOB_LO);
    else if constexpr (KIND == 7) asm volatile(R32(F3, 8) ::: CLOB_LO);
    else if constexpr (KIND == 8) asm volatile(R32(MV, 8) ::: CLOB_LO);
    else if constexpr (KIND == 9) asm volatile(R32(CN, 8) ::: CLOB_LO);
    else if constexpr (KIND >= 10 && KIND <= 14) {
      // one v_cmp -> vcc, K plain float instructions, one v_cndmask on that vcc (K = 0, 1, 3, 7), x8; KIND 14: vcc written by SALU first
#define F1 "v_fmac_f32_e32 v20, v1, v2\n\t"
#define GRP(FILL) "v_cmp_gt_f32_e32 vcc, v1, v2\n\t" FILL "v_cndmask_b32_e32 v8, v1, v2, vcc\n\t"
#define GRS "s_mov_b64 vcc, s[20:21]\n\tv_cndmask_b32_e32 v8, v1, v2, vcc\n\t"
      if constexpr (KIND == 10) asm volatile(GRP("") GRP("") GRP("") GRP("") GRP("") GRP("") GRP("") GRP("") ::: "v8", "v20", "vcc");
      else if constexpr (KIND == 11) asm volatile(GRP(F1) GRP(F1) GRP(F1) GRP(F1) GRP(F1) GRP(F1) GRP(F1) GRP(F1) ::: "v8", "v20", "vcc");
      else if constexpr (KIND == 12) asm volatile(GRP(F1 F1 F1) GRP(F1 F1 F1) GRP(F1 F1 F1) GRP(F1 F1 F1) GRP(F1 F1 F1) GRP(F1 F1 F1) GRP(F1 F1 F1) GRP(F1 F1 F1) ::: "v8", "v20", "vcc");
      else if constexpr (KIND == 13) asm volatile(GRP(F1 F1 F1 F1 F1 F1 F1) GRP(F1 F1 F1 F1 F1 F1 F1) GRP(F1 F1 F1 F1 F1 F1 F1) GRP(F1 F1 F1 F1 F1 F1 F1) GRP(F1 F1 F1 F1 F1 F1 F1) GRP(F1 F1 F1 F1 F1 F1 F1) GRP(F1 F1 F1 F1 F1 F1 F1) GRP(F1 F1 F1 F1 F1 F1 F1) ::: "v8", "v20", "vcc");
      else asm volatile(GRS GRS GRS GRS GRS GRS GRS GRS ::: "v8", "vcc", "s20", "s21");
    }
    else if constexpr (KIND == 15) {   // v_cmp into an SGPR pair, then v_cndmask_e64 on it directly
#define GRE "v_cmp_gt_f32_e64 s[20:21], v1, v2\n\tv_cndmask_b32_e64 v8, v1, v2, s[20:21]\n\t"
      asm volatile(GRE GRE GRE GRE GRE GRE GRE GRE ::: "v8", "s20", "s21");
    }
    else if constexpr (KIND == 16 || KIND == 17) {   // cross-lane swaps of gfx950 (the halving steps of a multi-value reduction)
#define SW32(B, i) "v_permlane32_swap_b32_e32 v[" STR(B) "+" STR(i) "], v1\n\t"
#define SW16(B, i) "v_permlane16_swap_b32_e32 v[" STR(B) "+" STR(i) "], v1\n\t"
      if constexpr (KIND == 16) asm volatile(R32(SW32, 8) ::: CLOB_LO, "v1");
      else asm volatile(R32(SW16, 8) ::: CLOB_LO, "v1");
    }
    else if constexpr (KIND == 18) {   // v_add_f32_dpp row_shr (in place, independent registers)
#define DP(B, i) "v_add_f32_dpp v[" STR(B) "+" STR(i) "], v[" STR(B) "+" STR(i) "], v[" STR(B) "+" STR(i) "] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      asm volatile(R32(DP, 8) ::: CLOB_LO);
    }
    else if constexpr (KIND == 6) {
#define PK(B, i) "v_pk_fma_f32 v[" STR(B) "+2*" STR(i) ":" STR(B) "+2*" STR(i) "+1], v[2:3], v[4:5], v[" STR(B) "+2*" STR(i) ":" STR(B) "+2*" STR(i) "+1]\n\t"
      asm volatile(PK(8, 0) PK(8, 1) PK(8, 2) PK(8, 3) PK(8, 4) PK(8, 5) PK(8, 6) PK(8, 7) PK(8, 8) PK(8, 9) PK(8, 10) PK(8, 11) PK(8, 12) PK(8, 13) PK(8, 14) PK(8, 15) ::: CLOB_LO);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s;
  asm volatile("v_mov_b32 %0, v8" : "=v"(s));
  out[blockIdx.x * 256 + tid] = s;
  if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int ninstr, const float* in, float* out, long long* cyc) {
  const int iters = 400;
  for (int wps : {1, 2, 4, 6, 8}) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, 5);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    printf("%-44s wps=%d  cyc/instr/SIMD = %6.2f   (per wave: %6.2f)\n", name, wps, med / ((double)iters * ninstr * wps), med / ((double)iters * ninstr));
    fflush(stdout);
  }
}

int main() {
  float *in, *out; long long* cyc;
  (void)hipMalloc(&in, 256 * 4); (void)hipMalloc(&out, 2048 * 256 * 4); (void)hipMalloc(&cyc, 2048 * 4 * 8);
  std::vector<float> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 1.f + 0.37f * (float)((i * 7919) % 101);
  (void)hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
  run<0>("v_fmac on v8..v39 (independent)", 32, in, out, cyc);
  run<1>("v_fmac on v80..v111", 32, in, out, cyc);
  run<2>("v_fmac on v160..v191", 32, in, out, cyc);
  run<3>("v_fmac on v8, 32 dependent", 32, in, out, cyc);
  run<4>("v_fmac + v_cmp vcc alternating (64)", 64, in, out, cyc);
  run<5>("v_add_f32 low, 200 VGPRs allocated", 32, in, out, cyc);
  run<6>("v_pk_fma_f32 x16 low pairs", 16, in, out, cyc);
  run<7>("v_fma_f32 (VOP3) low", 32, in, out, cyc);
  run<8>("v_mov_b32 low", 32, in, out, cyc);
  run<9>("v_cndmask_b32_e32 vcc low", 32, in, out, cyc);
  run<16>("v_permlane32_swap_b32", 32, in, out, cyc);
  run<17>("v_permlane16_swap_b32", 32, in, out, cyc);
  run<18>("v_add_f32_dpp row_shr:1 (independent)", 32, in, out, cyc);
  run<10>("[v_cmp vcc, v_cndmask vcc] x8 (16 instr)", 16, in, out, cyc);
  run<11>("[v_cmp, 1 fmac, v_cndmask] x8 (24)", 24, in, out, cyc);
  run<12>("[v_cmp, 3 fmac, v_cndmask] x8 (40)", 40, in, out, cyc);
  run<13>("[v_cmp, 7 fmac, v_cndmask] x8 (72)", 72, in, out, cyc);
  run<14>("[s_mov vcc, v_cndmask vcc] x8 (16)", 16, in, out, cyc);
  run<15>("[v_cmp_e64 sgpr, v_cndmask_e64 sgpr] x8 (16)", 16, in, out, cyc);
  return 0;
}
