// Shared host/device definitions for libsrslte_phy_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define SRSLTE_SUCCESS 0
#define SRSLTE_ERROR -1
#define SRSLTE_ERROR_INVALID_INPUTS -2

// Diagnostics. The reference's ERROR() macro (lib/include/srslte/phy/utils/debug.h:75-89) prints to stderr unless the application has
// registered a handler with srslte_phy_log_register_handler (utils/phy_logger.c:37-52), in which case the text goes to that callback.
// hip_log does the same: when this library is linked into a program that also carries the reference's phy_logger.c (the link-time drop-in
// of INTEGRATION.md §1), `handler_registered` and `srslte_phy_log_print` resolve to the reference's and a registered handler receives
// every message; on its own the library has neither symbol (weak, null) and prints to stderr.
void hip_log(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// Every HIP failure surfaces as SRSLTE_ERROR with a diagnostic through hip_log (config.h:58-66 convention).
#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e__ = (expr);                                                                            \
    if (e__ != hipSuccess) {                                                                            \
      hip_log("[srslte_hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__);    \
      return SRSLTE_ERROR;                                                                              \
    }                                                                                                   \
  } while (0)

#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

#ifndef SRSLTE_HIP_GRANTS_TB_DIRECT
#define SRSLTE_HIP_GRANTS_TB_DIRECT 1 // 0: the DL grants mode always assembles its transport blocks with tb_crc_bytes_kernel (A/B builds)
#endif

#ifndef SRSLTE_HIP_GRANTS_DESC_BY_KERNEL
#define SRSLTE_HIP_GRANTS_DESC_BY_KERNEL 1 // 0: the DL grants mode's descriptors reach the device through hipMemcpyAsync (A/B builds)
#endif

typedef float2 cf32; // layout-compatible with C99 float _Complex (cf_t, config.h:68)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- LTE numerology (restates phy_common.c:322-345, phy_common.h:93-116) ----
static inline int lte_symbol_sz(int nof_prb)
{
  if (nof_prb <= 0) return -1;
  if (nof_prb <= 6) return 128;
  if (nof_prb <= 15) return 256;
  if (nof_prb <= 25) return 384;
  if (nof_prb <= 50) return 768;
  if (nof_prb <= 75) return 1024;
  if (nof_prb <= 110) return 1536;
  return -1;
}
static inline int lte_cp_len(int N, int c) { return (c * N + 2047) / 2048; }
static inline int lte_cp_len_norm(int l, int N) { return l == 0 ? lte_cp_len(N, 160) : lte_cp_len(N, 144); }
static inline int lte_cp_len_ext(int N) { return lte_cp_len(N, 512); }
