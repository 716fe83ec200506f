// Development probe: issue cost of dependent packed-int16 / DPP chains and of s_nop with 1, 2 and 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_issue.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(name, body)                                                         \
  __global__ __launch_bounds__(64) void name(int* out, int n)                      \
  {                                                                                \
    int v = threadIdx.x, c = 0x00010001, w = threadIdx.x * 3;                      \
    for (int i = 0; i < n; i++) {                                                  \
      asm volatile(REP16(body) : "+v"(v), "+v"(w) : "v"(c));                       \
    }                                                                              \
    if (v == 0x12345678 && w == 77) out[0] = v;                                    \
  }
// 1: dependent pk adds, no nops.  2: dependent pk adds with s_nop 0.  3: two independent chains interleaved.
// 4: dependent add -> dpp -> add (nop 1 before the dpp as the hazard requires).  5: same, two chains interleaved, no nops needed
KERNEL(k_dep, "v_pk_add_i16 %0, %0, %2 clamp\n")
KERNEL(k_dep_nop, "v_pk_add_i16 %0, %0, %2 clamp\n s_nop 0\n")
KERNEL(k_two, "v_pk_add_i16 %0, %0, %2 clamp\n v_pk_add_i16 %1, %1, %2 clamp\n")
KERNEL(k_dpp, "v_pk_add_i16 %0, %0, %2 clamp\n s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dpp2, "v_pk_add_i16 %0, %0, %2 clamp\n v_pk_add_i16 %1, %1, %2 clamp\n s_nop 0\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_max, "v_pk_add_i16 %0, %0, %2 clamp\n v_pk_max_i16 %0, %0, %2\n")

template <typename K>
void run(const char* name, K k, int instr_per_body, int* d)
{
  const int n = 4096;
  for (int waves : {1, 2, 4}) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(1024 * waves), dim3(64), 0, 0, d, n);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k, dim3(1024 * waves), dim3(64), 0, 0, d, n);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    // per SIMD: waves * n * 16 bodies
    double bodies = (double)waves * n * 16;
    printf("%-10s waves/SIMD %d: %.3f ms, %.2f ns per body per SIMD (%d instr/body) -> %.2f cycles@2.4GHz per body\n", name, waves, ms,
           ms * 1e6 / bodies, instr_per_body, ms * 1e6 / bodies * 2.4);
  }
}
int main()
{
  int* d;
  hipMalloc(&d, 4);
  run("dep", k_dep, 1, d);
  run("dep_nop0", k_dep_nop, 2, d);
  run("two", k_two, 2, d);
  run("dpp", k_dpp, 3, d);
  run("dpp2", k_dpp2, 5, d);
  run("add_max", k_max, 2, d);
  return 0;
}
