// Development probe (round 3): what ONE wavefront (and 2 / 4 wavefronts of one workgroup on the same SIMD) issues per cycle, timed
// inside the kernel with s_memtime so that neither workgroup placement nor clock ramps enter.  One workgroup on the whole device.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_single.hip -o scripts/ubench_single && scripts/ubench_single
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define IND8(op, tail)                                                                             \
  op " %0, %0, %8" tail "\n" op " %1, %1, %8" tail "\n" op " %2, %2, %8" tail "\n" op " %3, %3, %8" tail "\n" \
  op " %4, %4, %8" tail "\n" op " %5, %5, %8" tail "\n" op " %6, %6, %8" tail "\n" op " %7, %7, %8" tail "\n"
#define KERNEL8(name, body)                                                                        \
  __global__ void name(unsigned long long* out, int n)                                             \
  {                                                                                                \
    int v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7, c = 0x00010001; \
    __syncthreads();                                                                               \
    const unsigned long long t0 = __builtin_readcyclecounter();                                    \
    for (int i = 0; i < n; i++) {                                                                  \
      asm volatile(REP16(body) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c)); \
    }                                                                                              \
    const unsigned long long t1 = __builtin_readcyclecounter();                                    \
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                                  \
    if ((v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7) == 0x12345678) out[63] = v0;                       \
  }
KERNEL8(k_pk_add, IND8("v_pk_add_i16", " clamp"))
KERNEL8(k_add32, IND8("v_add_u32", ""))
KERNEL8(k_max16, IND8("v_max_i16", ""))
KERNEL8(k_mix, "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_pk_add_i16 %2, %2, %8 clamp\n v_pk_max_i16 %3, %3, %8\n"
               "v_pk_add_i16 %4, %4, %8 clamp\n v_pk_add_i16 %5, %5, %8 clamp\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8\n")
// dependent chains: N interleaved chains of v_pk_add_i16 (latency of a dependent packed op), and the decoder's own chain
// add -> dpp -> max -> add ... with 1, 2 or 4 chains interleaved
#define DEP1 "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n" \
             "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n"
#define DEP2 "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n" \
             "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n"
#define DEP4 "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_pk_add_i16 %2, %2, %8 clamp\n v_pk_add_i16 %3, %3, %8 clamp\n" \
             "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_pk_add_i16 %2, %2, %8 clamp\n v_pk_add_i16 %3, %3, %8 clamp\n"
#define DPPC(r) "v_pk_add_i16 " r ", " r ", %8 clamp\n s_nop 1\n v_mov_b32_dpp " r ", " r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_pk_max_i16 " r ", " r ", %8\n"
#define DPPC2 "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" \
              "v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n nop_pad"
KERNEL8(k_dep1, DEP1)
KERNEL8(k_dep2, DEP2)
KERNEL8(k_dep4, DEP4)
KERNEL8(k_dppc1, DPPC("%0") DPPC("%0"))
KERNEL8(k_dppc2, "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n"
                 "v_pk_add_i16 %2, %2, %8 clamp\n v_pk_add_i16 %3, %3, %8 clamp\n")
template <typename K>
int run(const char* name, K k, unsigned long long* d)
{
  const int n = 256; // x 128 instructions
  for (int threads : {64, 256, 512, 1024}) {
    unsigned long long h[16];
    hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, n);
    hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, n);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    double mx = 0;
    for (int w = 0; w < threads / 64; w++) mx = h[w] > mx ? h[w] : mx;
    const int wps = threads <= 256 ? 1 : threads / 256;
    printf("%-8s %4d threads (%d wave(s) per SIMD): %.2f cycles per instruction per wave, %.2f per SIMD\n", name, threads, wps, mx / (n * 128.0),
           mx / (n * 128.0) / wps);
  }
  return 0;
}
int main()
{
  unsigned long long* d;
  CHK(hipMalloc(&d, 64 * 8));
  // (dppc1: 6 instructions + 2 s_nop per body of 8 slots; cycles are per 8-instruction body slot)
  if (run("dep1", k_dep1, d) || run("dep2", k_dep2, d) || run("dep4", k_dep4, d) || run("dppc1", k_dppc1, d) || run("dppc2", k_dppc2, d)) return 1;
  if (run("pk_add", k_pk_add, d) || run("add_u32", k_add32, d) || run("max_i16", k_max16, d) || run("mix_pk", k_mix, d)) return 1;
  return 0;
}
