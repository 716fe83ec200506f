// Development probe (round 3): how a grid of one-wavefront workgroups is spread over the chip's SIMDs and what each wavefront then
// issues per cycle, timed INSIDE the kernel (s_memtime = shader cycles, s_memrealtime = 100 MHz) so that placement, issue rate and the
// clock the chip holds under this load can be told apart.  Grids: 832 x 64 threads (the decoder's launch for 128 subframes), 1024, 2048,
// 3328, each with 13 KB of LDS per workgroup as the decoder.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_placement.hip -o scripts/ubench_placement && scripts/ubench_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define BODY_PK "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_pk_max_i16 %2, %2, %8\n v_pk_add_i16 %3, %3, %8 clamp\n" \
                "v_pk_add_i16 %4, %4, %8 clamp\n v_pk_max_i16 %5, %5, %8\n v_pk_add_i16 %6, %6, %8 clamp\n v_pk_max_i16 %7, %7, %8\n"
#define BODY_U32 "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n" \
                 "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
struct Rec { unsigned long long cyc, rt; unsigned hw, xcc; };
#define KERNEL(name, body)                                                                                                        \
  __global__ __launch_bounds__(64) void name(Rec* out, int n)                                                                     \
  {                                                                                                                               \
    extern __shared__ int lds[];                                                                                                  \
    int v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7, c = 0x00010001; \
    lds[threadIdx.x] = v0;                                                                                                        \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();                            \
    for (int i = 0; i < n; i++) {                                                                                                 \
      asm volatile(REP16(body) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c));        \
    }                                                                                                                             \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                            \
    unsigned hw, xcc;                                                                                                             \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));           \
    if (threadIdx.x == 0) out[blockIdx.x] = Rec{t1 - t0, r1 - r0, hw, xcc};                                                       \
    if ((v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7) == 0x12345678) out[0].cyc = lds[v0 & 63];                                         \
  }
KERNEL(k_pk, BODY_PK)
KERNEL(k_u32, BODY_U32)
template <typename K>
int run(const char* name, K k, Rec* d, int grid)
{
  const int n = 1024; // x 128 instructions: ~0.25-0.5 ms
  std::vector<Rec> h(grid);
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(grid), dim3(64), 13 * 1024, 0, d, n);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(grid), dim3(64), 13 * 1024, 0, d, n);
  CHK(hipEventRecord(b));
  CHK(hipEventSynchronize(b));
  float ms;
  CHK(hipEventElapsedTime(&ms, a, b));
  CHK(hipMemcpy(h.data(), d, grid * sizeof(Rec), hipMemcpyDeviceToHost));
  std::map<unsigned long long, int> per_simd, per_cu;
  double cyc = 0, clk = 0, cmax = 0;
  for (auto& r : h) {
    const unsigned simd = (r.hw >> 4) & 3, cu = (r.hw >> 8) & 15, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7, xcc = r.xcc & 15;
    const unsigned long long cuid = ((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu;
    per_simd[(cuid << 4) | simd]++;
    per_cu[cuid]++;
    cyc += (double)r.cyc;
    cmax = std::max(cmax, (double)r.cyc);
    clk += (double)r.cyc / (double)r.rt * 0.1; // GHz
  }
  std::map<int, int> hist_simd, hist_cu;
  for (auto& kv : per_simd) hist_simd[kv.second]++;
  for (auto& kv : per_cu) hist_cu[kv.second]++;
  printf("%-6s grid %5d: %.3f ms by events; in-kernel mean %.2f cycles per instruction (max %.2f), clock %.2f GHz; SIMDs used %zu, CUs used %zu\n", name, grid, ms,
         cyc / grid / (n * 128.0), cmax / (n * 128.0), clk / grid, per_simd.size(), per_cu.size());
  printf("        wavefronts per SIMD -> number of SIMDs:");
  for (auto& kv : hist_simd) printf("  %d: %d", kv.first, kv.second);
  printf("\n        wavefronts per CU -> number of CUs:");
  for (auto& kv : hist_cu) printf("  %d: %d", kv.first, kv.second);
  printf("\n");
  return 0;
}
int main()
{
  Rec* d;
  CHK(hipMalloc(&d, 8192 * sizeof(Rec)));
  for (int grid : {256, 832, 1024, 2048, 3328}) {
    if (run("pk", k_pk, d, grid) || run("u32", k_u32, d, grid)) return 1;
  }
  return 0;
}
