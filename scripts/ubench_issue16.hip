// Development probe (round 3): issue rate of the 16-bit VALU forms a turbo-decoder mapping could be built from, next to the packed
// forms of scripts/ubench_issue.hip.  hipcc --offload-arch=gfx950 -O3 scripts/ubench_issue16.hip -o /tmp/ub16 && /tmp/ub16 [out.json]
// Workgroups of 256 threads (one wavefront per SIMD of the CU they land on), grid = 256 x k: k wavefronts per SIMD when the
// dispatcher spreads them evenly (ubench_issue.hip's 64-thread groups did not guarantee that at k = 1).
// Each kernel: 8 independent chains, one instruction per chain per body.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#define REP4(x) x x x x
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define IND8(op, tail)                                                                             \
  op " %0, %0, %8" tail "\n" op " %1, %1, %8" tail "\n" op " %2, %2, %8" tail "\n" op " %3, %3, %8" tail "\n" \
  op " %4, %4, %8" tail "\n" op " %5, %5, %8" tail "\n" op " %6, %6, %8" tail "\n" op " %7, %7, %8" tail "\n"
#define IND8_3(op, tail)                                                                             \
  op " %0, %0, %8, %1" tail "\n" op " %1, %1, %8, %2" tail "\n" op " %2, %2, %8, %3" tail "\n" op " %3, %3, %8, %4" tail "\n" \
  op " %4, %4, %8, %5" tail "\n" op " %5, %5, %8, %6" tail "\n" op " %6, %6, %8, %7" tail "\n" op " %7, %7, %8, %0" tail "\n"

#define KERNEL8(name, body)                                                                        \
  __global__ __launch_bounds__(256) void name(int* out, int n)                                     \
  {                                                                                                \
    int v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7, c = 0x00010001; \
    for (int i = 0; i < n; i++) {                                                                  \
      asm volatile(REP4(body) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c)); \
    }                                                                                              \
    if ((v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7) == 0x12345678) out[0] = v0;                        \
  }
KERNEL8(k_pk_add, IND8("v_pk_add_i16", " clamp"))
KERNEL8(k_pk_max, IND8("v_pk_max_i16", ""))
KERNEL8(k_add32, IND8("v_add_u32", ""))
KERNEL8(k_add_i16c, IND8("v_add_i16", " clamp"))
KERNEL8(k_add_i16c_hi, IND8("v_add_i16", " op_sel:[1,0,1] clamp"))
KERNEL8(k_sub_i16c, IND8("v_sub_i16", " clamp"))
KERNEL8(k_max_i16, IND8("v_max_i16", ""))
KERNEL8(k_max_i16_dpp, IND8("v_max_i16_dpp", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
KERNEL8(k_max_i16_ror, IND8("v_max_i16_dpp", " row_ror:8 row_mask:0xf bank_mask:0xf"))
KERNEL8(k_max3_i16, IND8_3("v_max3_i16", ""))
KERNEL8(k_med3_i16, IND8_3("v_med3_i16", ""))
KERNEL8(k_add_u16, IND8("v_add_u16", ""))
KERNEL8(k_add_u16_dpp, IND8("v_add_u16_dpp", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
KERNEL8(k_bfi, IND8_3("v_bfi_b32", ""))
KERNEL8(k_perm, IND8_3("v_perm_b32", ""))
KERNEL8(k_and, IND8("v_and_b32", ""))
KERNEL8(k_max_i32, IND8("v_max_i32", ""))
KERNEL8(k_max_i32_dpp, IND8("v_max_i32_dpp", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
KERNEL8(k_add_i32c, IND8("v_add_i32", " clamp"))
KERNEL8(k_med3_i32, IND8_3("v_med3_i32", ""))
KERNEL8(k_max3_i32, IND8_3("v_max3_i32", ""))
KERNEL8(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n")
// the 8-states-in-a-lane trellis step on packed pairs: 12 adds + 8 max, independent within the step (two interleaved chains of 4)
KERNEL8(k_mix_pk, "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_pk_add_i16 %2, %2, %8 clamp\n v_pk_max_i16 %3, %3, %8\n"
                  "v_pk_add_i16 %4, %4, %8 clamp\n v_pk_add_i16 %5, %5, %8 clamp\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8\n")
KERNEL8(k_mix_16, "v_add_i16 %0, %0, %8 clamp\n v_add_i16 %1, %1, %8 clamp\n v_add_i16 %2, %2, %8 clamp\n v_max_i16 %3, %3, %8\n"
                  "v_add_i16 %4, %4, %8 clamp\n v_add_i16 %5, %5, %8 clamp\n v_max_i16 %6, %6, %8\n v_max_i16 %7, %7, %8\n")

static std::string g_json;
static double      g_ghz = 2.4;

template <typename K>
int run(const char* name, K k, int* d)
{
  const int n = 2048;
  for (int waves : {1, 2, 4, 8}) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(256 * waves), dim3(256), 0, 0, d, n);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k, dim3(256 * waves), dim3(256), 0, 0, d, n);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms;
    CHK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    const double instr_per_simd = (double)waves * n * 32; // wave-instructions issued by one SIMD
    const double cyc            = ms * 1e6 * g_ghz / instr_per_simd;
    printf("%-14s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD\n", name, waves, ms, cyc);
    char buf[256];
    snprintf(buf, sizeof buf, "%s{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_wave_instr\": %.3f}", g_json.empty() ? "" : ", ", name,
             waves, ms, cyc);
    g_json += buf;
    CHK(hipEventDestroy(a));
    CHK(hipEventDestroy(b));
  }
  return 0;
}
#define RUN(nm, k) if (run(nm, k, d)) return 1;
int main(int argc, char** argv)
{
  int* d;
  CHK(hipMalloc(&d, 4));
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  g_ghz = p.clockRate / 1e6;
  printf("device %s, %d CUs, clockRate %.2f GHz\n", p.name, p.multiProcessorCount, g_ghz);
  RUN("pk_add_i16c", k_pk_add) RUN("pk_max_i16", k_pk_max) RUN("add_u32", k_add32) RUN("add_i16c", k_add_i16c) RUN("add_i16c_hi", k_add_i16c_hi)
  RUN("sub_i16c", k_sub_i16c) RUN("max_i16", k_max_i16) RUN("max_i16_dpp", k_max_i16_dpp) RUN("max_i16_ror8", k_max_i16_ror) RUN("max3_i16", k_max3_i16)
  RUN("med3_i16", k_med3_i16) RUN("add_u16", k_add_u16) RUN("add_u16_dpp", k_add_u16_dpp) RUN("bfi_b32", k_bfi) RUN("perm_b32", k_perm) RUN("and_b32", k_and)
  RUN("max_i32", k_max_i32) RUN("max_i32_dpp", k_max_i32_dpp) RUN("add_i32c", k_add_i32c) RUN("med3_i32", k_med3_i32) RUN("max3_i32", k_max3_i32)
  RUN("mov_dpp", k_mov_dpp) RUN("mix_pk", k_mix_pk) RUN("mix_16", k_mix_16)
  if (argc > 1) {
    FILE* f = fopen(argv[1], "w");
    fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f, \"results\": [%s]}\n", p.name, p.multiProcessorCount, g_ghz, g_json.c_str());
    fclose(f);
  }
  return 0;
}
