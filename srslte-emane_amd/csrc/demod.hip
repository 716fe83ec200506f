// Soft demapper for gfx950: srslte_demod_soft_demodulate / _s / _b (demod_soft.c:479-549), batched.
//
// Max-log LLRs by |x|-offset folding. The int16/int8 variants reproduce the reference's per-modulation rounding
// (SURVEY §2.3 K9): QPSK = x*scale truncated toward zero + saturating pack (vector_simd.c:392-427); 16/64QAM SSE
// bodies = round-to-nearest-even (_mm_cvtps_epi32) + saturating pack + integer offsets 252 / 432,216 (demod_soft.c:95,
// :246-247), their scalar tails (nsymbols % 4, % 8 for int8) truncate and subtract a double offset (demod_soft.c:124-132,
// :290-300); 256QAM = float fold, then C cast (demod_soft.c:435-477). Sign convention: LLR > 0 <=> bit 1.
// Out-of-int32-range products (|symbol| > ~3e6) are undefined upstream and not reproduced.
//
// HBM-bound streaming kernel: one thread per symbol, 8 B read and Qm*sizeof(llr) written; optional fused LLR
// descrambling (scrambling.c:45-48) so that the PDSCH pipeline saves one read+write pass over the LLRs.
#include "common.hpp"
#include "phy_hip_internal.hpp"

#include "demod_dev.hpp"

namespace {
using namespace demod_dev;

// TYPE: 0 float, 1 int16, 2 int8. Symbols [ncalls][nsym]; LLRs [ncalls][nsym*Qm].
// scr: optional packed scrambling bits [10][scr_words] (bit i of call c = word[i/32] >> (i%32)), selected by (tti0+call)%10.
template <int TYPE>
__global__ __launch_bounds__(256) void demod_kernel(int mod, const cf32* __restrict__ sym, void* __restrict__ llr, int nsym, int ncalls,
                                                    const uint32_t* __restrict__ scr, int scr_words, int tti0)
{
  const int Qm = mod_bits(mod);
  const long total = (long)nsym * ncalls;
  for (long gi = (long)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (long)gridDim.x * blockDim.x) {
    const int  call = (int)(gi / nsym), i = (int)(gi - (long)call * nsym);
    const cf32 s = sym[gi];
    if constexpr (TYPE == 0) {
      float o[8];
      demod_f(mod, s, o);
      float* dst = (float*)llr + gi * Qm;
      for (int j = 0; j < Qm; j++) dst[j] = o[j];
    } else if constexpr (TYPE == 1) {
      short o[8];
      demod_s(mod, s, i, nsym, o);
      short* dst = (short*)llr + gi * Qm;
      if (scr) {
        const uint32_t* c = scr + (size_t)((tti0 + call) % 10) * scr_words;
        for (int j = 0; j < Qm; j++) {
          const int bit = i * Qm + j;
          if ((c[bit >> 5] >> (bit & 31)) & 1) o[j] = (short)-o[j]; // srslte_vec_neg_sss
        }
      }
      for (int j = 0; j < Qm; j++) dst[j] = o[j];
    } else {
      signed char o[8];
      demod_b(mod, s, i, nsym, o);
      signed char* dst = (signed char*)llr + gi * Qm;
      if (scr) {
        const uint32_t* c = scr + (size_t)((tti0 + call) % 10) * scr_words;
        for (int j = 0; j < Qm; j++) {
          const int bit = i * Qm + j;
          if ((c[bit >> 5] >> (bit & 31)) & 1) o[j] = (signed char)-o[j];
        }
      }
      for (int j = 0; j < Qm; j++) dst[j] = o[j];
    }
  }
}


// int16 fast path: 4 symbols per thread (two 16-B loads, 4*Qm int16 = 1..4 16-B stores): same per-symbol arithmetic,
// wider memory instructions. Needs nsym % 4 == 0 so that a group never straddles two calls.
template <int QM>
__global__ __launch_bounds__(256) void demod_s_x4_kernel(int mod, const float4* __restrict__ sym, uint4* __restrict__ llr, int nsym, long groups,
                                                         const uint32_t* __restrict__ scr, int scr_words, int tti0)
{
  for (long gi = (long)blockIdx.x * blockDim.x + threadIdx.x; gi < groups; gi += (long)gridDim.x * blockDim.x) {
    const long   s0   = gi * 4;
    const int    call = (int)(s0 / nsym), i0 = (int)(s0 - (long)call * nsym);
    const float4 a = sym[2 * gi], b = sym[2 * gi + 1];
    const cf32   sv[4] = {make_float2(a.x, a.y), make_float2(a.z, a.w), make_float2(b.x, b.y), make_float2(b.z, b.w)};
    union {
      short h[4 * QM];
      uint4 v[QM / 2];
    } out;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      short o[8];
      demod_s(mod, sv[t], i0 + t, nsym, o);
#pragma unroll
      for (int j = 0; j < QM; j++) out.h[t * QM + j] = o[j];
    }
    if (scr) {
      const uint32_t* c = scr + (size_t)((tti0 + call) % 10) * scr_words;
#pragma unroll
      for (int j = 0; j < 4 * QM; j++) {
        const int bit = i0 * QM + j;
        if ((c[bit >> 5] >> (bit & 31)) & 1) out.h[j] = (short)-out.h[j];
      }
    }
#pragma unroll
    for (int j = 0; j < QM / 2; j++) llr[gi * (QM / 2) + j] = out.v[j];
  }
}

} // namespace

int demod_launch(int type, int mod, const void* d_sym, void* d_llr, int nsym, int ncalls, const uint32_t* d_scr, int scr_words, int tti0,
                 hipStream_t st)
{
  if (mod < 0 || mod > 4) {
    hip_log("[srslte_hip] Invalid modulation %d\n", mod);
    return SRSLTE_ERROR;
  }
  if (!d_sym || !d_llr || nsym < 0 || ncalls < 0) return SRSLTE_ERROR_INVALID_INPUTS;
  const long total = (long)nsym * ncalls;
  if (total == 0) return SRSLTE_SUCCESS;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (type == 1 && mod >= MOD_QPSK && nsym % 4 == 0 && ((uintptr_t)d_sym % 16) == 0 && ((uintptr_t)d_llr % 16) == 0) {
    const long groups = total / 4;
    long       gb     = (groups + 255) / 256;
    if (gb > 16384) gb = 16384;
    const float4* s4 = (const float4*)d_sym;
    uint4*        l4 = (uint4*)d_llr;
    switch (mod) {
      case MOD_QPSK: hipLaunchKernelGGL(demod_s_x4_kernel<2>, dim3((unsigned)gb), dim3(256), 0, st, mod, s4, l4, nsym, groups, d_scr, scr_words, tti0); break;
      case MOD_16QAM: hipLaunchKernelGGL(demod_s_x4_kernel<4>, dim3((unsigned)gb), dim3(256), 0, st, mod, s4, l4, nsym, groups, d_scr, scr_words, tti0); break;
      case MOD_64QAM: hipLaunchKernelGGL(demod_s_x4_kernel<6>, dim3((unsigned)gb), dim3(256), 0, st, mod, s4, l4, nsym, groups, d_scr, scr_words, tti0); break;
      default: hipLaunchKernelGGL(demod_s_x4_kernel<8>, dim3((unsigned)gb), dim3(256), 0, st, mod, s4, l4, nsym, groups, d_scr, scr_words, tti0); break;
    }
    LAUNCH_CHECK();
    return SRSLTE_SUCCESS;
  }
  switch (type) {
    case 0: hipLaunchKernelGGL(demod_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, mod, (const cf32*)d_sym, d_llr, nsym, ncalls, d_scr, scr_words, tti0); break;
    case 1: hipLaunchKernelGGL(demod_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, mod, (const cf32*)d_sym, d_llr, nsym, ncalls, d_scr, scr_words, tti0); break;
    default: hipLaunchKernelGGL(demod_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, mod, (const cf32*)d_sym, d_llr, nsym, ncalls, d_scr, scr_words, tti0); break;
  }
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}

// C ABI: batched forms of srslte_demod_soft_demodulate{,_s,_b} (demod_soft.h:39-53); device pointers.
extern "C" int srslte_hip_demod_soft_demodulate_batch(int mod, const void* d_symbols, float* d_llr, int nsymbols, int ncalls, void* stream)
{
  return demod_launch(0, mod, d_symbols, d_llr, nsymbols, ncalls, nullptr, 0, 0, (hipStream_t)stream);
}
extern "C" int srslte_hip_demod_soft_demodulate_s_batch(int mod, const void* d_symbols, short* d_llr, int nsymbols, int ncalls, void* stream)
{
  return demod_launch(1, mod, d_symbols, d_llr, nsymbols, ncalls, nullptr, 0, 0, (hipStream_t)stream);
}
extern "C" int srslte_hip_demod_soft_demodulate_b_batch(int mod, const void* d_symbols, int8_t* d_llr, int nsymbols, int ncalls, void* stream)
{
  return demod_launch(2, mod, d_symbols, d_llr, nsymbols, ncalls, nullptr, 0, 0, (hipStream_t)stream);
}
