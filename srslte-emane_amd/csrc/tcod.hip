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

// Byte-packed variant (srslte_tcod_encode_lut, turbocoder.c:189-367, without its CRC bookkeeping): K/8 input bytes MSB first;
// parity = p1[K] | t1[4] | p2[K] | t2[4] as one MSB-first bit stream (K/4+1 bytes), sys_tail = the nibble the reference
// stores in input[K/8]. Same chunked GF(2) scheme with whole bytes per lane.
__global__ __launch_bounds__(64) void tcod_bytes_kernel(const uint8_t* __restrict__ in, uint32_t in_stride, uint8_t* __restrict__ parity,
                                                        uint32_t par_stride, uint8_t* __restrict__ sys_tail,
                                                        const uint16_t* __restrict__ perm, int K)
{
  __shared__ uint8_t xs[768], p2[768];
  __shared__ int     zmap[8], resp[2][64], start[2][65], tails[12];
  const int          cb = blockIdx.x, lane = threadIdx.x, nbytes = K / 8;
  const uint8_t*     x = in + (size_t)cb * in_stride;
  uint8_t*           par = parity + (size_t)cb * par_stride;
  const int          cB = (nbytes + 63) / 64, lo = min(nbytes, lane * cB), hi = min(nbytes, lo + cB);
  for (int i = lane; i < nbytes; i += 64) xs[i] = x[i];
  if (lane < 8) {
    int s = lane;
    for (int i = 0; i < 8 * cB; i++) rsc_step(s, 0);
    zmap[lane] = s;
  }
  __syncthreads();
  auto bit_at = [&](int i) { return (xs[i >> 3] >> (7 - (i & 7))) & 1; };
  int  sa = 0, sb = 0;
  for (int i = 8 * lo; i < 8 * hi; i++) {
    rsc_step(sa, bit_at(i));
    rsc_step(sb, bit_at(perm[i]));
  }
  resp[0][lane] = sa;
  resp[1][lane] = sb;
  __syncthreads();
  if (lane < 2) {
    int s = 0;
    for (int l = 0; l < 64; l++) {
      start[lane][l] = s;
      const int n    = min(nbytes, (l + 1) * cB) - min(nbytes, l * cB);
      if (n == cB) {
        s = zmap[s] ^ resp[lane][l];
      } else if (n > 0) {
        int z = s;
        for (int i = 0; i < 8 * n; i++) rsc_step(z, 0);
        s = z ^ resp[lane][l];
      }
    }
    // termination (turbocoder.c:294-335): tails[6*enc + 2j] = x, [.. + 2j + 1] = z
    for (int j = 0; j < 3; j++) {
      const int bit          = (s & 1) ^ ((s >> 1) & 1);
      tails[6 * lane + 2 * j]     = bit;
      tails[6 * lane + 2 * j + 1] = rsc_step(s, bit);
    }
  }
  __syncthreads();
  sa = start[0][lane];
  sb = start[1][lane];
  for (int b = lo; b < hi; b++) {
    int o1 = 0, o2 = 0;
    for (int j = 0; j < 8; j++) {
      o1 = (o1 << 1) | rsc_step(sa, bit_at(8 * b + j));
      o2 = (o2 << 1) | rsc_step(sb, bit_at(perm[8 * b + j]));
    }
    par[b] = (uint8_t)o1;
    p2[b]  = (uint8_t)o2;
  }
  __syncthreads();
  auto nib = [&](int j) { return (tails[j] << 3) | (tails[j + 3] << 2) | (tails[j + 6] << 1) | tails[j + 9]; }; // turbocoder.c:337-349
  for (int j = lane; j <= nbytes; j += 64) {
    const int hi4 = j == 0 ? nib(1) : (p2[j - 1] & 0xf);
    const int lo4 = j == nbytes ? nib(2) : (p2[j] >> 4);
    par[nbytes + j] = (uint8_t)((hi4 << 4) | lo4);
  }
  if (lane == 0) sys_tail[cb] = (uint8_t)(nib(0) << 4);
}

std::mutex                     g_mtx;
std::map<long, uint16_t*>      g_perm; // device*8192 + K -> device QPP table

} // namespace

static int tcod_perm(uint32_t long_cb, uint16_t** d_perm_out)
{
  const int idx = lte_cb_index(long_cb);
  if (idx < 0 || lte_qpp_table[idx].K != long_cb) {
    hip_log("[srslte_hip] Invalid CB size %u\n", long_cb); // turbocoder.c:89-93
    return SRSLTE_ERROR;
  }
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mtx);
  auto it = g_perm.find((long)dev * 8192 + long_cb);
  if (it == g_perm.end()) {
    std::vector<uint16_t> f, r;
    lte_qpp_tables(long_cb, 1, f, r);
    uint16_t* d_perm = nullptr;
    HIP_TRY(hipMalloc((void**)&d_perm, long_cb * 2));
    HIP_TRY(hipMemcpy(d_perm, f.data(), long_cb * 2, hipMemcpyHostToDevice));
    it = g_perm.emplace((long)dev * 8192 + long_cb, d_perm).first;
  }
  *d_perm_out = it->second;
  return SRSLTE_SUCCESS;
}

extern "C" int srslte_hip_tcod_encode_batch(const uint8_t* d_input, uint8_t* d_output, uint32_t long_cb, uint32_t nof_cb, void* stream)
{
  if (!d_input || !d_output) return SRSLTE_ERROR_INVALID_INPUTS;
  uint16_t* d_perm = nullptr;
  if (int r = tcod_perm(long_cb, &d_perm)) return r;
  if (nof_cb == 0) return SRSLTE_SUCCESS;
  hipLaunchKernelGGL(tcod_kernel, dim3(nof_cb), dim3(64), 0, (hipStream_t)stream, d_input, d_output, (const uint16_t*)d_perm, (int)long_cb);
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}

extern "C" int srslte_hip_tcod_encode_bytes_batch(const uint8_t* d_input, uint32_t in_stride, uint8_t* d_parity, uint32_t par_stride,
                                                  uint8_t* d_sys_tail, uint32_t long_cb, uint32_t nof_cb, void* stream)
{
  if (!d_input || !d_parity || !d_sys_tail || in_stride < long_cb / 8 || par_stride < long_cb / 4 + 1) return SRSLTE_ERROR_INVALID_INPUTS;
  uint16_t* d_perm = nullptr;
  if (int r = tcod_perm(long_cb, &d_perm)) return r;
  if (nof_cb == 0) return SRSLTE_SUCCESS;
  hipLaunchKernelGGL(tcod_bytes_kernel, dim3(nof_cb), dim3(64), 0, (hipStream_t)stream, d_input, in_stride, d_parity, par_stride, d_sys_tail,
                     (const uint16_t*)d_perm, (int)long_cb);
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}
