// Development probe: VALU issue rate of gfx950 for the instructions the turbo decoder is made of.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_issue.hip -o /tmp/ubench && /tmp/ubench [out.json]
// Part 1 (throughput): 8 INDEPENDENT chains per wave of v_pk_add_i16 clamp / v_pk_max_i16 / v_mov_b32_dpp at 1, 2, 4, 8 waves per
// SIMD -> cycles per wave-instruction per SIMD. MI355X_MICROARCH.md:53-54,473,489 says SIMD-32: a wave64 VALU op takes 2 cycles,
// reachable once >= 2 waves share a SIMD. The measured figure is what bench.py prices the decoder's VALU count against.
// Part 2 (latency): dependent chains (what a single trellis recursion is) with and without the s_nop a DPP consumer needs.
// The device clock during the run is measured with s_memrealtime (100 MHz constant) against s_memtime (shader clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// 8 independent chains (registers %0..%7), one instruction per chain per body: no dependent-issue stall inside a wave
#define IND8(op, tail)                                                                             \
  op " %0, %0, %8" tail "\n" op " %1, %1, %8" tail "\n" op " %2, %2, %8" tail "\n" op " %3, %3, %8" tail "\n" \
  op " %4, %4, %8" tail "\n" op " %5, %5, %8" tail "\n" op " %6, %6, %8" tail "\n" op " %7, %7, %8" tail "\n"
#define DPP8                                                                                       \
  "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" \
  "v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" \
  "v_mov_b32_dpp %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n"                     \
  "v_mov_b32_dpp %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"

#define KERNEL8(name, body)                                                                        \
  __global__ __launch_bounds__(64) void name(int* out, int n)                                      \
  {                                                                                                \
    int v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7, c = 0x00010001; \
    for (int i = 0; i < n; i++) {                                                                  \
      asm volatile(REP4(body) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c)); \
    }                                                                                              \
    if ((v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7) == 0x12345678) out[0] = v0;                        \
  }
KERNEL8(k_ind_add, IND8("v_pk_add_i16", " clamp"))
KERNEL8(k_ind_max, IND8("v_pk_max_i16", ""))
KERNEL8(k_ind_dpp, DPP8)
KERNEL8(k_ind_add32, IND8("v_add_u32", ""))
// the decoder's mix: per 8 instructions 4 adds, 2 max, 2 dpp (independent)
KERNEL8(k_ind_mix, "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_pk_max_i16 %2, %2, %8\n"
                   "v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_pk_add_i16 %4, %4, %8 clamp\n"
                   "v_pk_add_i16 %5, %5, %8 clamp\n v_pk_max_i16 %6, %6, %8\n v_mov_b32_dpp %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n")

#define KERNEL(name, body)                                                         \
  __global__ __launch_bounds__(64) void name(int* out, int n)                      \
  {                                                                                \
    int v = threadIdx.x, c = 0x00010001, w = threadIdx.x * 3;                      \
    for (int i = 0; i < n; i++) {                                                  \
      asm volatile(REP16(body) : "+v"(v), "+v"(w) : "v"(c));                       \
    }                                                                              \
    if (v == 0x12345678 && w == 77) out[0] = v;                                    \
  }
KERNEL(k_dep, "v_pk_add_i16 %0, %0, %2 clamp\n")
KERNEL(k_two, "v_pk_add_i16 %0, %0, %2 clamp\n v_pk_add_i16 %1, %1, %2 clamp\n")
KERNEL(k_dpp, "v_pk_add_i16 %0, %0, %2 clamp\n s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_max, "v_pk_add_i16 %0, %0, %2 clamp\n v_pk_max_i16 %0, %0, %2\n")

__global__ void k_clock(unsigned long long* out)
{ // shader clock (s_memtime) against the 100 MHz constant clock (s_memrealtime) over a busy loop
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  int v = threadIdx.x;
  for (int i = 0; i < 200000; i++) asm volatile(REP16("v_add_u32 %0, %0, %0\n") : "+v"(v));
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = v; }
}

static std::string g_json;
static double g_ghz = 2.4;

template <typename K>
int run(const char* name, K k, int instr_per_body, int bodies_per_iter, int* d)
{
  const int n = 2048;
  for (int waves : {1, 2, 4, 8}) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(1024 * waves), dim3(64), 0, 0, d, n); // 256 CUs x 4 SIMDs x waves
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k, dim3(1024 * waves), dim3(64), 0, 0, d, n);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms;
    CHK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    double instr_per_simd = (double)waves * n * bodies_per_iter * instr_per_body; // wave-instructions issued by one SIMD
    double cyc = ms * 1e6 * g_ghz / instr_per_simd;
    printf("%-10s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD (@%.2f GHz)\n", name, waves, ms, cyc, g_ghz);
    char buf[256];
    snprintf(buf, sizeof buf, "%s{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_wave_instr\": %.3f}", g_json.empty() ? "" : ", ", name,
             waves, ms, cyc);
    g_json += buf;
  }
  return 0;
}
int main(int argc, char** argv)
{
  int* d;
  unsigned long long* dc;
  CHK(hipMalloc(&d, 4));
  CHK(hipMalloc(&dc, 24));
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, dc);
  unsigned long long hc[3];
  CHK(hipMemcpy(hc, dc, 24, hipMemcpyDeviceToHost));
  // s_memtime counts at the shader clock? On gfx9 it is the constant 100 MHz counter too on some parts: report both, and also derive
  // the clock from the known cost of the busy loop below (the independent-add kernel at 1 wave is latency-bound, so not that).
  printf("clock probe: memtime delta %llu, memrealtime delta %llu (100 MHz) -> memtime runs at %.1f MHz\n", hc[0], hc[1], 100.0 * hc[0] / hc[1]);
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  g_ghz = p.clockRate / 1e6;
  printf("device %s, %d CUs, clockRate %.2f GHz\n", p.name, p.multiProcessorCount, g_ghz);
  if (run("ind_add", k_ind_add, 8, 4, d)) return 1;
  if (run("ind_max", k_ind_max, 8, 4, d)) return 1;
  if (run("ind_dpp", k_ind_dpp, 8, 4, d)) return 1;
  if (run("ind_add32", k_ind_add32, 8, 4, d)) return 1;
  if (run("ind_mix", k_ind_mix, 8, 4, d)) return 1;
  if (run("dep_add", k_dep, 1, 16, d)) return 1;
  if (run("two_add", k_two, 2, 16, d)) return 1;
  if (run("dep_dpp", k_dpp, 2, 16, d)) return 1; // s_nop not counted as an instruction
  if (run("dep_addmax", k_max, 2, 16, d)) return 1;
  if (argc > 1) {
    FILE* f = fopen(argv[1], "w");
    fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f, \"memtime_mhz\": %.1f, \"results\": [%s]}\n", p.name, p.multiProcessorCount, g_ghz,
            100.0 * hc[0] / hc[1], g_json.c_str());
    fclose(f);
  }
  return 0;
}
