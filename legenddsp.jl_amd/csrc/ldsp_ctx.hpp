// ldsp_ctx.hpp — internals shared by the translation units that implement the C ABI.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <string>
#include "../../include/ldsp.h"
#include "icpc_dev.hpp"

int ldsp_fail(int code, const char* fmt, ...);
#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) return ldsp_fail(LDSP_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

// Makes `dev` the calling thread's current HIP device for the lifetime of the object and restores the caller's device
// afterwards (torch reads the current device through hipGetDevice: an entry point that left ITS device current would move
// later device='cuda' allocations of a multi-GPU process to the wrong GPU).  dev < 0: no-op.
struct ldsp_device_guard {
  int prev = -1;
  explicit ldsp_device_guard(int dev) {
    if (dev < 0) return;
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != dev) { prev = cur; (void)hipSetDevice(dev); }
    else if (cur < 0) (void)hipSetDevice(dev);
  }
  ~ldsp_device_guard() { if (prev >= 0) (void)hipSetDevice(prev); }
  ldsp_device_guard(const ldsp_device_guard&) = delete;
  ldsp_device_guard& operator=(const ldsp_device_guard&) = delete;
};

struct ldsp_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // dsp_icpc parameter staging
  ldsp_icpc_params icpc_last{};
  bool icpc_valid = false;
  int icpc_mode_built = -1;
  int icpc_u16_built = -1;   // the in_u16 flag the device copy of the parameter block was built with
  ldsp::IcpcDev icpc_host{};
  ldsp::IcpcDev* d_icpc = nullptr;
  float* d_hc = nullptr;
  float* d_hz = nullptr;
  // the parameter block used before the current one (a routine that alternates between two blocks — dsp_icpc_compressed: presummed and
  // windowed traces — lowers and uploads each of them once)
  ldsp_icpc_params icpc_last_b{};
  bool icpc_valid_b = false;
  int icpc_mode_built_b = -1;
  int icpc_u16_built_b = -1;
  ldsp::IcpcDev icpc_host_b{};
  ldsp::IcpcDev* d_icpc_b = nullptr;
  float* d_hc_b = nullptr;
  float* d_hz_b = nullptr;
  float* d_aux = nullptr;   // [aux_cap][4] kernel 1 -> kernel 2 hand-over (blmean, t50 position)
  int64_t aux_cap = 0;
  float* d_fir_grid = nullptr;   // [fir_grid_cap] taps of ldsp_fir_grid_run (grow-only)
  size_t fir_grid_cap = 0;
  void* d_sg_grid = nullptr;  // SgGridDev of ldsp_sg_grid_run
  void* d_sipm = nullptr;     // SipmDev of ldsp_sipm_run (k_sipm_s4 reads it through a pointer); re-uploaded when it changes
  std::vector<unsigned char> sipm_last;
  void* d_grid = nullptr;   // TrapGridDev of ldsp_trap_grid_run (allocated on first use)
  float* d_coef = nullptr;  // [LDSP_MAX_FIR_TAPS] staging for functor coefficients / small tables
  int cusp_direct = 0;
  int sipm_generic = 0; // option "sipm_generic": always the generic two-array dsp_sipm kernel
  int two_kernel = 0;   // option "two_kernel": never fuse the CUSP/ZAC stage into icpc_kernel
  int dbg_stop = 0;
  int icpc_generic = 0;  // option "icpc_generic": always icpc_kernel, never the lean kernel (comparator)
  int icpc_r2 = 0;      // option "icpc_r2": 4097..8192-sample traces on 1024 threads x 8 samples (8 waves per SIMD) instead of 512 x 16
  long long* dbg_stamps = nullptr;   // option "dbg_stamps": device buffer of a diagnostic (LDSP_STAMPS) build
  int multi_serial = 0;          // option "multi_serial": MultiIntersect by the reference's one-lane walk (comparator of the wave-parallel search)
  int fir_grid_per_point = 0;    // option "fir_grid_per_point": ldsp_fir_grid_run evaluates every grid point's filter outputs (comparator)
  // timing
  int timing = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evm = nullptr;  // evm: boundary between the two dsp_icpc kernels
  hipEvent_t evs = nullptr;                                 // recorded at the end of every run call on the stream that ran it: a newly set stream waits for it
  bool evs_recorded = false;
  int n_launches = 0, n_stages = 1;
  const char* last_kernel = "";   // dominant kernel of the last ldsp_*_run call (static string)
};

// Scope of one run call: the context's device is current, and at the end — however the call returns — an event is recorded on the
// stream the call used.  A later change of stream (ldsp_ctx_set_stream) makes the new stream wait for that event and never touches the
// previous stream again: its owner may have destroyed it in the meantime (hipEventRecord on a destroyed stream does not fail, it crashes).
struct ldsp_run_guard {
  ldsp_device_guard dev;
  ldsp_ctx* c;
  explicit ldsp_run_guard(ldsp_ctx* ctx) : dev(ctx ? ctx->device : -1), c(ctx) {}
  ~ldsp_run_guard() {
    if (c && c->evs) {
      if (hipEventRecord(c->evs, c->stream) == hipSuccess) c->evs_recorded = true;
      else (void)hipGetLastError();
    }
  }
  ldsp_run_guard(const ldsp_run_guard&) = delete;
  ldsp_run_guard& operator=(const ldsp_run_guard&) = delete;
};

// common argument checks of the per-trace entry points
int ldsp_check_batch(ldsp_ctx* c, const void* x, int64_t n, int32_t L, const char* who);

// Argument checks common to the batch entry points, then the context's device made current for the rest of the CALLER's scope
// (restored at its exit).  Used as a statement:  int rc = ldsp_check_batch(c, x, n, L, "name");
int ldsp_check_batch_impl(ldsp_ctx* c, const void* x, int64_t n, int32_t L, const char* who);
#define ldsp_check_batch(c, x, n, L, who) ldsp_check_batch_impl((c), (x), (n), (L), (who)); ldsp_run_guard ldsp_guard_((c))
