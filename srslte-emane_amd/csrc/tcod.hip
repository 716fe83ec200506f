// LTE turbo encoder for gfx950: srslte_tcod_encode (turbocoder.c:76-186), batched over code blocks.
//
// Rate-1/3 PCCC: two 8-state RSC encoders (feedback 1+D^2+D^3, feed-forward 1+D+D^3), the second fed through the
// QPP interleaver, 12 tail bits. The recursion is serial in the state, but it is linear over GF(2): the state after a
// chunk is Z(start) ^ E, with Z the zero-input evolution of the chunk and E the zero-state response. One wavefront
// per code block: every lane (1) runs its chunk of K/64 bits from state 0 to get E, (2) the 64 chunk start states are
// chained through LDS, (3) every lane re-runs its chunk from its true start state and emits parity. 2x the bit work,
// 64x the parallelism; reads and writes are byte-per-bit streams as in the reference's unpacked API.
#include "common.hpp"
#include "phy_hip_internal.hpp"
#include <map>
#include <mutex>

namespace {

__device__ __forceinline__ int rsc_step(int& s, int bit)
{ // state s = r0<<2 | r1<<1 | r2 (turbocoder.c:120-125)
  const int r0 = (s >> 2) & 1, r1 = (s >> 1) & 1, r2 = s & 1;
  const int in = bit ^ r2 ^ r1, out = r2 ^ r0 ^ in;
  s            = (in << 2) | (r0 << 1) | r1;
  return out;
}

__global__ __launch_bounds__(64) void tcod_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const uint16_t* __restrict__ perm,
                                                  int K)
{
  __shared__ int zmap[2][8];   // zero-input evolution over one full chunk, per start state
  __shared__ int resp[2][64];  // zero-state response of each chunk, per encoder
  __shared__ int start[2][65]; // chunk start states
  const int      cb = blockIdx.x, lane = threadIdx.x;
  const uint8_t* x = in + (size_t)cb * K;
  uint8_t*       y = out + (size_t)cb * (3 * K + 12);
  const int      chunk = (K + 63) / 64, lo = lane * chunk, hi = min(K, lo + chunk);

  if (lane < 8) {
    int s = lane;
    for (int i = 0; i < chunk; i++) rsc_step(s, 0);
    zmap[0][lane] = s;
    zmap[1][lane] = s;
  }
  int sa = 0, sb = 0;
  for (int i = lo; i < hi; i++) {
    rsc_step(sa, x[i] & 1);
    rsc_step(sb, x[perm[i]] & 1);
  }
  resp[0][lane] = sa;
  resp[1][lane] = sb;
  __syncthreads();
  if (lane < 2) {
    int s = 0;
    for (int l = 0; l < 64; l++) {
      start[lane][l] = s;
      const int n    = min(K, (l + 1) * chunk) - min(K, l * chunk);
      if (n == chunk) {
        s = zmap[lane][s] ^ resp[lane][l];
      } else if (n > 0) { // short last chunk: evolve explicitly
        int z = s;
        for (int i = 0; i < n; i++) rsc_step(z, 0);
        s = z ^ resp[lane][l];
      }
    }
    start[lane][64] = s; // state after K bits
  }
  __syncthreads();
  sa = start[0][lane];
  sb = start[1][lane];
  for (int i = lo; i < hi; i++) {
    const int b = x[i];
    y[3 * i]     = (uint8_t)b;
    y[3 * i + 1] = (uint8_t)rsc_step(sa, b & 1);
    y[3 * i + 2] = (uint8_t)rsc_step(sb, x[perm[i]] & 1);
  }
  if (lane < 2) { // termination (turbocoder.c:146-184): x = r2^r1 drives the register to zero, z is the parity
    int      s = start[lane][64];
    uint8_t* t = y + 3 * K + 6 * lane;
    for (int j = 0; j < 3; j++) {
      const int bit = (s & 1) ^ ((s >> 1) & 1);
      t[2 * j]      = (uint8_t)bit;
      t[2 * j + 1]  = (uint8_t)rsc_step(s, bit);
    }
  }
}

std::mutex                     g_mtx;
std::map<long, uint16_t*>      g_perm; // device*8192 + K -> device QPP table

} // namespace

extern "C" int srslte_hip_tcod_encode_batch(const uint8_t* d_input, uint8_t* d_output, uint32_t long_cb, uint32_t nof_cb, void* stream)
{
  if (!d_input || !d_output) return SRSLTE_ERROR_INVALID_INPUTS;
  const int idx = lte_cb_index(long_cb);
  if (idx < 0 || lte_qpp_table[idx].K != long_cb) {
    fprintf(stderr, "[srslte_hip] Invalid CB size %u\n", long_cb); // turbocoder.c:89-93
    return SRSLTE_ERROR;
  }
  if (nof_cb == 0) return SRSLTE_SUCCESS;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  uint16_t* d_perm = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mtx);
    auto it = g_perm.find((long)dev * 8192 + long_cb);
    if (it == g_perm.end()) {
      std::vector<uint16_t> f, r;
      lte_qpp_tables(long_cb, 1, f, r);
      HIP_TRY(hipMalloc((void**)&d_perm, long_cb * 2));
      HIP_TRY(hipMemcpy(d_perm, f.data(), long_cb * 2, hipMemcpyHostToDevice));
      g_perm[(long)dev * 8192 + long_cb] = d_perm;
    } else {
      d_perm = it->second;
    }
  }
  hipLaunchKernelGGL(tcod_kernel, dim3(nof_cb), dim3(64), 0, (hipStream_t)stream, d_input, d_output, (const uint16_t*)d_perm, (int)long_cb);
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}
