// Sanitizer driver of the host-side lowering (tests/test_sanitizers_cpu.py builds it with g++ -fsanitize=address,undefined
// together with csrc/ldsp_api.hip compiled as plain C++): reads raw ldsp_icpc_params blocks from a file and runs the complete
// lowering of each (ldsp_icpc_check_params), plus the coefficient entry points over a sweep of shapes.  Prints one return
// code per block; a sanitizer report makes the process fail.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>

#include "../../include/ldsp.h"
#include "../../legenddsp.jl_amd/csrc/icpc_dev.hpp"

// the launchers live in the kernel translation units; the driver never launches
namespace ldsp {
hipError_t launch_icpc(const float*, int64_t, int, int, bool, const IcpcDev*, float*, const IcpcOutDev&, const float*, float, bool, bool, bool, int, int,
                       hipStream_t, hipEvent_t, int*) { return hipErrorNotSupported; }
hipError_t launch_icpc_lean3(const float*, int64_t, int, int, bool, bool, const IcpcDev*, const IcpcOutDev&, const float*, float, int, bool, hipStream_t) { return hipErrorNotSupported; }
size_t icpc_lean3_smem_bytes(int NT, int Lf) { return (size_t)(NT * 16 + Lf) * 4; }
int g_dbg_lds_pad = 0;
hipError_t launch_pz_trap_lean(const float*, int64_t, int, bool, const IcpcDev*, float*, float*, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_pz_trap(const float*, int64_t, int, bool, const IcpcDev*, float*, float*, hipStream_t) { return hipErrorNotSupported; }
size_t icpc_smem_bytes(int NT) { return (size_t)NT * 16 * 8; }
hipError_t launch_trap_grid(const float*, int64_t, int, bool, const TrapGridDev*, float*, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_sg_grid(const float*, int64_t, int, bool, const SgGridDev*, float*, float*, float*, float*, float*, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_fir_grid(const float*, int64_t, int, bool, const FirGridDev*, float*, hipStream_t) { return hipErrorNotSupported; }
}  // namespace ldsp

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<unsigned char> buf(sizeof(ldsp_icpc_params));
  int n = 0;
  while (fread(buf.data(), 1, buf.size(), f) == buf.size()) {
    ldsp_icpc_params p;
    memcpy(&p, buf.data(), sizeof p);
    const int rc = ldsp_icpc_check_params(&p);
    printf("%d %d\n", n++, rc);
    // the coefficient entry points on this block's filters
    if (p.cusp.length >= 5 && p.cusp.length <= LDSP_MAX_FIR_TAPS) { std::vector<double> h((size_t)p.cusp.length); (void)ldsp_cusp_coeffs(&p.cusp, h.data()); }
    if (p.zac.length >= 5 && p.zac.length <= LDSP_MAX_FIR_TAPS) { std::vector<double> h((size_t)p.zac.length); (void)ldsp_zac_coeffs(&p.zac, h.data()); }
  }
  fclose(f);
  for (int npts = 1; npts <= 65; npts += 2)
    for (int deg = 0; deg <= 5; ++deg)
      for (int der = 0; der <= 2; ++der) {
        std::vector<double> c((size_t)npts);
        (void)ldsp_sg_coeffs(npts, deg, der, c.data());
      }
  printf("done %d\n", n);
  return 0;
}
