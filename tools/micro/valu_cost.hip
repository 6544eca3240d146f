// valu_cost.hip — what ONE vector instruction of each kind costs a SIMD of gfx950 at the occupancy of icpc_lean3_kernel (512-thread
// workgroups, three per CU = 6 waves per SIMD), in the two currencies the kernel is accounted in:
//   * wall: cycles per instruction and SIMD from in-kernel s_memtime (all resident waves run the same unrolled block concurrently);
//   * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU of the same kernels under `rocprofv3 --pmc` (every kind is its own kernel name).
// Round 3 priced classes from stand-alone loops at 4 waves per SIMD, where a wave's own issue interval (~8 cycles) hides the difference
// between a 2-cycle and a 4-cycle instruction; here 6 waves per SIMD saturate the pipe and the counter tells the classes apart.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_cost.hip -o tools/micro/valu_cost
// run:   tools/micro/valu_cost            (table)      rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES -- tools/micro/valu_cost pmc
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#define STR2(x) #x
#define STR(x) STR2(x)
#define R32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
  X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define V(i) "v[8+" STR(i) "]"
#define VP(i) "v[8+2*" STR(i) ":8+2*" STR(i) "+1]"
#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", \
             "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "vcc", "s20", "s21", "s22", "s23"

// one entry per kind: name, instructions per block, asm of the block
#define KINDS(K) \
  K(0, "v_fmac_f32_e32", 32, R32(I0)) \
  K(1, "v_fma_f32 (VOP3)", 32, R32(I1)) \
  K(2, "v_add_f32_e32", 32, R32(I2)) \
  K(3, "v_mul_f32_e32", 32, R32(I3)) \
  K(4, "v_mov_b32_e32", 32, R32(I4)) \
  K(5, "v_max_f32_e32", 32, R32(I5)) \
  K(6, "v_max3_f32", 32, R32(I6)) \
  K(7, "v_med3_f32", 32, R32(I7)) \
  K(8, "v_cmp_gt_f32_e32 vcc (no reader)", 32, R32(I8)) \
  K(9, "[v_cmp_e32 vcc, v_cndmask_e32 vcc] x16", 32, R16(I9)) \
  K(10, "[v_cmp_e64 sgpr, v_cndmask_e64 sgpr] x16", 32, R16(I10)) \
  K(11, "v_add_f32_dpp row_shr:1", 32, R32(I11)) \
  K(12, "v_mov_b32_dpp row_shr:1", 32, R32(I12)) \
  K(13, "v_add_u32_e32", 32, R32(I13)) \
  K(14, "v_lshlrev_b32_e32", 32, R32(I14)) \
  K(15, "v_and_b32_e32", 32, R32(I15)) \
  K(16, "v_lshl_add_u32", 32, R32(I16)) \
  K(17, "v_add3_u32", 32, R32(I17)) \
  K(18, "v_cvt_f32_i32", 32, R32(I18)) \
  K(19, "v_cvt_f64_f32", 16, R16(I19)) \
  K(20, "v_add_f64", 16, R16(I20)) \
  K(21, "v_fma_f64", 16, R16(I21)) \
  K(22, "v_pk_fma_f32", 16, R16(I22)) \
  K(23, "v_pk_add_f32", 16, R16(I23)) \
  K(24, "v_pk_mul_f32", 16, R16(I24)) \
  K(25, "v_readlane_b32", 32, R32(I25)) \
  K(26, "v_writelane_b32", 32, R32(I26)) \
  K(27, "v_permlane32_swap_b32", 32, R32(I27)) \
  K(28, "v_log_f32", 32, R32(I28)) \
  K(29, "v_rcp_f32", 32, R32(I29)) \
  K(30, "v_fma_f32 (VOP3, SGPR operand)", 32, R32(I30)) \
  K(31, "v_fmac_f32_e32 (SGPR src0)", 32, R32(I31)) \
  K(32, "v_fmac_f32_e32 (literal src0)", 32, R32(I32)) \
  K(33, "[v_cmp_e32 vcc, v_addc_co_u32 vcc] x16", 32, R16(I33)) \
  K(34, "v_max_f32_dpp row_shr:1", 32, R32(I34)) \
  K(35, "v_or_b32_dpp row_shr:1", 32, R32(I35)) \
  K(36, "v_bfe_u32", 32, R32(I36)) \
  K(37, "v_mad_u32_u24", 32, R32(I37)) \
  K(38, "v_mul_lo_u32", 32, R32(I38)) \
  K(39, "[v_cmp_lt_u32_e32 vcc, v_cndmask vcc] x16", 32, R16(I39)) \
  K(40, "v_sub_f32_e32", 32, R32(I40)) \
  K(41, "v_cndmask_b32_e64 (SGPR pair nobody writes)", 32, R32(I41)) \
  K(42, "v_mov_b32 literal", 32, R32(I42)) \
  K(43, "v_cmp_gt_f32_e64 sgpr (no reader)", 32, R32(I43)) \
  K(44, "v_cvt_i32_f32", 32, R32(I44)) \
  K(45, "v_pk_mov_b32", 16, R16(I45)) \
  K(46, "v_bfi_b32", 32, R32(I46)) \
  K(47, "v_perm_b32", 32, R32(I47)) \
  K(48, "v_min_u32_e32", 32, R32(I48)) \
  K(49, "v_mul_f64", 16, R16(I49)) \
  K(50, "v_cvt_f32_f64", 16, R16(I50)) \
  K(51, "[v_fmac, v_max_f32] x16", 32, R16(I51)) \
  K(52, "[v_fmac, v_add_f32_dpp] x16", 32, R16(I52)) \
  K(53, "[v_fmac, v_add_u32] x16", 32, R16(I53)) \
  K(54, "[v_fmac, v_fmac, v_fmac, v_max3] x8", 32, R8(I54)) \
  K(55, "v_mov_b32_dpp wave_shr:1", 32, R32(I55)) \
  K(56, "v_add_f32_dpp row_bcast:15", 32, R32(I56)) \
  K(57, "v_fmac_f32_dpp row_shr:1", 32, R32(I57)) \
  K(58, "v_cmp_class / v_cmp_ge_f32 e32 + v_cndmask, 2 fmac between", 32, R8(I58)) \
  K(59, "v_xor_b32_e32", 32, R32(I59)) \
  K(60, "v_max_f32 VOP3 (neg modifier)", 32, R32(I60)) \
  K(61, "v_add_f32 VOP3 (abs modifier)", 32, R32(I61)) \
  K(62, "v_add_co_u32 + v_addc (64-bit add) x16", 32, R16(I62)) \
  K(63, "v_sqrt_f32", 32, R32(I63))

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define I0(i) "v_fmac_f32_e32 " V(i) ", v1, v2\n\t"
#define I1(i) "v_fma_f32 " V(i) ", v1, v2, " V(i) "\n\t"
#define I2(i) "v_add_f32_e32 " V(i) ", v1, " V(i) "\n\t"
#define I3(i) "v_mul_f32_e32 " V(i) ", v1, " V(i) "\n\t"
#define I4(i) "v_mov_b32_e32 " V(i) ", v1\n\t"
#define I5(i) "v_max_f32_e32 " V(i) ", v1, " V(i) "\n\t"
#define I6(i) "v_max3_f32 " V(i) ", v1, v2, " V(i) "\n\t"
#define I7(i) "v_med3_f32 " V(i) ", v1, v2, " V(i) "\n\t"
#define I8(i) "v_cmp_gt_f32_e32 vcc, v1, " V(i) "\n\t"
#define I9(i) "v_cmp_gt_f32_e32 vcc, v1, " V(i) "\n\tv_cndmask_b32_e32 " V(i) ", v1, v2, vcc\n\t"
#define I10(i) "v_cmp_gt_f32_e64 s[20:21], v1, " V(i) "\n\tv_cndmask_b32_e64 " V(i) ", v1, v2, s[20:21]\n\t"
#define I11(i) "v_add_f32_dpp " V(i) ", " V(i) ", " V(i) " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I12(i) "v_mov_b32_dpp " V(i) ", v1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I13(i) "v_add_u32_e32 " V(i) ", v1, " V(i) "\n\t"
#define I14(i) "v_lshlrev_b32_e32 " V(i) ", 1, " V(i) "\n\t"
#define I15(i) "v_and_b32_e32 " V(i) ", v1, " V(i) "\n\t"
#define I16(i) "v_lshl_add_u32 " V(i) ", v1, 2, " V(i) "\n\t"
#define I17(i) "v_add3_u32 " V(i) ", v1, v2, " V(i) "\n\t"
#define I18(i) "v_cvt_f32_i32_e32 " V(i) ", v1\n\t"
#define I19(i) "v_cvt_f64_f32_e32 " VP(i) ", v1\n\t"
#define I20(i) "v_add_f64 " VP(i) ", v[2:3], " VP(i) "\n\t"
#define I21(i) "v_fma_f64 " VP(i) ", v[2:3], v[4:5], " VP(i) "\n\t"
#define I22(i) "v_pk_fma_f32 " VP(i) ", v[2:3], v[4:5], " VP(i) "\n\t"
#define I23(i) "v_pk_add_f32 " VP(i) ", v[2:3], " VP(i) "\n\t"
#define I24(i) "v_pk_mul_f32 " VP(i) ", v[2:3], " VP(i) "\n\t"
#define I25(i) "v_readlane_b32 s20, " V(i) ", 63\n\t"
#define I26(i) "v_writelane_b32 " V(i) ", s22, 5\n\t"
#define I27(i) "v_permlane32_swap_b32_e32 " V(i) ", v1\n\t"
#define I28(i) "v_log_f32_e32 " V(i) ", v1\n\t"
#define I29(i) "v_rcp_f32_e32 " V(i) ", v1\n\t"
#define I30(i) "v_fma_f32 " V(i) ", s22, v2, " V(i) "\n\t"
#define I31(i) "v_fmac_f32_e32 " V(i) ", s22, v2\n\t"
#define I32(i) "v_fmac_f32_e32 " V(i) ", 0x3fc01234, v2\n\t"
#define I33(i) "v_cmp_ge_f32_e32 vcc, v1, " V(i) "\n\tv_addc_co_u32_e32 " V(i) ", vcc, " V(i) ", " V(i) ", vcc\n\t"
#define I34(i) "v_max_f32_dpp " V(i) ", " V(i) ", " V(i) " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I35(i) "v_or_b32_dpp " V(i) ", " V(i) ", " V(i) " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define I36(i) "v_bfe_u32 " V(i) ", v1, 3, 4\n\t"
#define I37(i) "v_mad_u32_u24 " V(i) ", v1, v2, " V(i) "\n\t"
#define I38(i) "v_mul_lo_u32 " V(i) ", v1, " V(i) "\n\t"
#define I39(i) "v_cmp_lt_u32_e32 vcc, v1, " V(i) "\n\tv_cndmask_b32_e32 " V(i) ", v1, v2, vcc\n\t"
#define I40(i) "v_sub_f32_e32 " V(i) ", v1, " V(i) "\n\t"
#define I41(i) "v_cndmask_b32_e64 " V(i) ", v1, v2, s[22:23]\n\t"
#define I42(i) "v_mov_b32_e32 " V(i) ", 0xff800000\n\t"
#define I43(i) "v_cmp_gt_f32_e64 s[20:21], v1, " V(i) "\n\t"
#define I44(i) "v_cvt_i32_f32_e32 " V(i) ", v1\n\t"
#define I45(i) "v_pk_mov_b32 " VP(i) ", v[2:3], v[4:5]\n\t"
#define I46(i) "v_bfi_b32 " V(i) ", v1, v2, " V(i) "\n\t"
#define I47(i) "v_perm_b32 " V(i) ", v1, v2, " V(i) "\n\t"
#define I48(i) "v_min_u32_e32 " V(i) ", v1, " V(i) "\n\t"
#define I49(i) "v_mul_f64 " VP(i) ", v[2:3], " VP(i) "\n\t"
#define I50(i) "v_cvt_f32_f64_e32 " V(i) ", v[2:3]\n\t"
#define I51(i) "v_fmac_f32_e32 " V(i) ", v1, v2\n\tv_max_f32_e32 v[24+" STR(i) "], v1, v[24+" STR(i) "]\n\t"
#define I52(i) "v_fmac_f32_e32 " V(i) ", v1, v2\n\tv_add_f32_dpp v[24+" STR(i) "], v[24+" STR(i) "], v[24+" STR(i) "] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I53(i) "v_fmac_f32_e32 " V(i) ", v1, v2\n\tv_add_u32_e32 v[24+" STR(i) "], v1, v[24+" STR(i) "]\n\t"
#define I54(i) "v_fmac_f32_e32 " V(i) ", v1, v2\n\tv_fmac_f32_e32 v[16+" STR(i) "], v1, v2\n\tv_fmac_f32_e32 v[24+" STR(i) "], v1, v2\n\tv_max3_f32 v[32+" STR(i) "], v1, v2, v[32+" STR(i) "]\n\t"
#define I55(i) "v_mov_b32_dpp " V(i) ", v1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I56(i) "v_add_f32_dpp " V(i) ", " V(i) ", " V(i) " row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
#define I57(i) "v_fmac_f32_dpp " V(i) ", " V(i) ", v1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I58(i) "v_cmp_ge_f32_e32 vcc, v1, " V(i) "\n\tv_fmac_f32_e32 v[16+" STR(i) "], v1, v2\n\tv_fmac_f32_e32 v[24+" STR(i) "], v1, v2\n\tv_cndmask_b32_e32 " V(i) ", v1, v2, vcc\n\t"
#define I59(i) "v_xor_b32_e32 " V(i) ", v1, " V(i) "\n\t"
#define I60(i) "v_max_f32_e64 " V(i) ", -v1, " V(i) "\n\t"
#define I61(i) "v_add_f32_e64 " V(i) ", |v1|, " V(i) "\n\t"
#define I62(i) "v_add_co_u32_e32 " VP(i) ", vcc, v1, " VP(i) "\n\t"   /* placeholder form: see K(62) below */
#define I63(i) "v_sqrt_f32_e32 " V(i) ", v1\n\t"
#undef I62
#define I62(i) "v_add_co_u32_e32 v[8+2*" STR(i) "], vcc, v1, v[8+2*" STR(i) "]\n\tv_addc_co_u32_e32 v[9+2*" STR(i) "], vcc, v2, v[9+2*" STR(i) "], vcc\n\t"

template <int KIND>
__global__ void __launch_bounds__(512) vc(const float* in, float* out, long long* cyc, int iters) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const float b = in[tid & 255] * 1e-3f, c = in[(tid + 7) & 255] * 1e-3f;
  if (iters < 0) lds[tid] = b;   // (keeps the dynamic LDS allocation, which sets the occupancy, alive)
  asm volatile("v_mov_b32 v1, %0\n\tv_mov_b32 v2, %1\n\tv_mov_b32 v3, %0\n\tv_mov_b32 v4, %1\n\tv_mov_b32 v5, %0\n\ts_mov_b32 s22, 0x3f800000\n\ts_mov_b32 s23, 0"
               :: "v"(b), "v"(c) : "v1", "v2", "v3", "v4", "v5", "s22", "s23");
  asm volatile(R32(I4) ::: CLOB);
  __syncthreads();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define K(id, name, n, body) if constexpr (KIND == id) asm volatile(body ::: CLOB);
    KINDS(K)
#undef K
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && tid == 0) cyc[1024 * 8] = r1 - r0;   // 100 MHz ticks of the same interval: s_memtime ticks per second
  float s;
  asm volatile("v_mov_b32 %0, v8" : "=v"(s));
  out[(size_t)blockIdx.x * 512 + tid] = s;
  if ((tid & 63) == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int ninstr, const float* in, float* out, long long* cyc, bool pmc, int long_iters) {
  const int iters = long_iters ? long_iters : pmc ? 200 : 400, wgs = 3, wps = 2 * wgs;
  const int blocks = 256 * wgs;
  const size_t lds = 53000;   // three workgroups per CU
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vc<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (!pmc) { hipLaunchKernelGGL(vc<KIND>, dim3(blocks), dim3(512), lds, 0, in, out, cyc, 5); (void)hipDeviceSynchronize(); }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(vc<KIND>, dim3(blocks), dim3(512), lds, 0, in, out, cyc, iters);
  (void)hipEventRecord(e1, 0);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * 8);
  long long rt = 0;
  (void)hipMemcpy(&rt, cyc + 1024 * 8, 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  const std::vector<long long>& hh = h;
  printf("K%-3d %-52s n=%2d  ticks/instr/SIMD = %6.2f  (per wave %6.2f)  wall %.3f ms  = %.3f ns/instr/SIMD; s_memtime at %.0f MHz\n", KIND, name, ninstr,
         med / ((double)iters * ninstr * wps), med / ((double)iters * ninstr), ms, 1e6 * ms / ((double)iters * ninstr * wps),
         rt > 0 ? (double)(hh[0] > 0 ? h[h.size() / 2] : 0) / (double)rt * 100.0 : 0.0);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");
  float *in, *out; long long* cyc;
  (void)hipMalloc(&in, 256 * 4); (void)hipMalloc(&out, (size_t)1024 * 512 * 4); (void)hipMalloc(&cyc, (1024 * 8 + 8) * 8);
  std::vector<float> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 1.f + 0.37f * (float)((i * 7919) % 101);
  (void)hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
  if (argc > 1 && !strcmp(argv[1], "long")) {   // long launches: the clock has ramped; wall time is the measure
    for (int rep = 0; rep < 2; ++rep) {
      run<0>("v_fmac_f32_e32", 32, in, out, cyc, false, 100000);
      run<6>("v_max3_f32", 32, in, out, cyc, false, 100000);
      run<11>("v_add_f32_dpp row_shr:1", 32, in, out, cyc, false, 100000);
      run<22>("v_pk_fma_f32", 16, in, out, cyc, false, 100000);
      run<54>("[v_fmac x3, v_max3] x8", 32, in, out, cyc, false, 100000);
      run<9>("[v_cmp vcc, v_cndmask vcc] x16", 32, in, out, cyc, false, 100000);
    }
    return 0;
  }
#define K(id, name, n, body) run<id>(name, n, in, out, cyc, pmc, 0);
  KINDS(K)
#undef K
  return 0;
}
