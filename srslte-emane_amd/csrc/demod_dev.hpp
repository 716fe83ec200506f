// Device-side soft-demapper primitives shared by demod.hip and the fused PDSCH kernel (pdsch.hip).
// See demod.hip for the rounding rules each variant reproduces (demod_soft.c).
#pragma once
#include "common.hpp"

namespace demod_dev {


enum { MOD_BPSK = 0, MOD_QPSK, MOD_16QAM, MOD_64QAM, MOD_256QAM };

__device__ __forceinline__ int   sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
__device__ __forceinline__ int   sat8(int v) { return v > 127 ? 127 : (v < -128 ? -128 : v); }
__device__ __forceinline__ short abs16(short v) { return (short)(v < 0 ? -v : v); }          // _mm_abs_epi16: -32768 stays
__device__ __forceinline__ signed char abs8(signed char v) { return (signed char)(v < 0 ? -v : v); }
__device__ __forceinline__ int   rne(float v) { return __float2int_rn(v); }                    // _mm_cvtps_epi32
__device__ __forceinline__ int   trunc_i(float v) { return (int)v; }                           // _mm_cvttps_epi32 / C cast

// ---- float ----------------------------------------------------------------
__device__ __forceinline__ void demod_f(int mod, cf32 s, float* o)
{
  switch (mod) {
    case MOD_BPSK: o[0] = (float)(-(double)(s.x + s.y) / 1.4142135623730951); break;
    case MOD_QPSK: {
      const float g = (float)-1.4142135623730951;
      o[0] = s.x * g;
      o[1] = s.y * g;
    } break;
    case MOD_16QAM:
      o[0] = -s.x;
      o[1] = -s.y;
      o[2] = (float)((double)fabsf(s.x) - 2 / 3.1622776601683795);
      o[3] = (float)((double)fabsf(s.y) - 2 / 3.1622776601683795);
      break;
    case MOD_64QAM:
      o[0] = -s.x;
      o[1] = -s.y;
      o[2] = (float)((double)fabsf(s.x) - 4 / 6.48074069840786);
      o[3] = (float)((double)fabsf(s.y) - 4 / 6.48074069840786);
      o[4] = (float)((double)fabsf(o[2]) - 2 / 6.48074069840786);
      o[5] = (float)((double)fabsf(o[3]) - 2 / 6.48074069840786);
      break;
    default: {
      float       re = -s.x, im = -s.y;
      const float o1 = 8.0f / 13.038404810405298f, o2 = 4.0f / 13.038404810405298f, o3 = 2.0f / 13.038404810405298f;
      o[0] = re; o[1] = im;
      re = fabsf(re) - o1; im = fabsf(im) - o1; o[2] = re; o[3] = im;
      re = fabsf(re) - o2; im = fabsf(im) - o2; o[4] = re; o[5] = im;
      re = fabsf(re) - o3; im = fabsf(im) - o3; o[6] = re; o[7] = im;
    }
  }
}

// ---- int16 ----------------------------------------------------------------
__device__ __forceinline__ void demod_s(int mod, cf32 s, int i, int nsym, short* o)
{
  const bool body = i < 4 * (nsym / 4); // SSE bodies of 16/64QAM take 4 symbols per step
  switch (mod) {
    case MOD_BPSK: o[0] = (short)((double)(-100.0f * (s.x + s.y)) / 1.4142135623730951); break;
    case MOD_QPSK: {
      const float g = (float)(-100 * 1.4142135623730951);
      // vector_simd.c:392-427: 16 floats per AVX2 step (cvtt + saturating pack), scalar C cast for the rest (wraps on x86)
      const bool vec = i < 8 * (nsym / 8);
      o[0] = vec ? (short)sat16(trunc_i(s.x * g)) : (short)trunc_i(s.x * g);
      o[1] = vec ? (short)sat16(trunc_i(s.y * g)) : (short)trunc_i(s.y * g);
    } break;
    case MOD_16QAM:
      if (body) {
        short re = (short)sat16(rne(s.x * -400.0f)), im = (short)sat16(rne(s.y * -400.0f));
        o[0] = re; o[1] = im;
        o[2] = (short)(abs16(re) - 252);
        o[3] = (short)(abs16(im) - 252);
      } else {
        short yre = (short)trunc_i(400 * s.x), yim = (short)trunc_i(400 * s.y);
        o[0] = (short)-yre; o[1] = (short)-yim;
        o[2] = (short)((double)abs((int)yre) - 2 * 400 / 3.1622776601683795);
        o[3] = (short)((double)abs((int)yim) - 2 * 400 / 3.1622776601683795);
      }
      break;
    case MOD_64QAM:
      if (body) {
        short re = (short)sat16(rne(s.x * -700.0f)), im = (short)sat16(rne(s.y * -700.0f));
        short a1 = (short)(abs16(re) - 432), b1 = (short)(abs16(im) - 432);
        o[0] = re; o[1] = im; o[2] = a1; o[3] = b1;
        o[4] = (short)(abs16(a1) - 216);
        o[5] = (short)(abs16(b1) - 216);
      } else {
        float yre = (float)(short)trunc_i(700 * s.x), yim = (float)(short)trunc_i(700 * s.y);
        o[0] = (short)-yre; o[1] = (short)-yim;
        o[2] = (short)((double)abs((int)yre) - 4 * 700 / 6.48074069840786);
        o[3] = (short)((double)abs((int)yim) - 4 * 700 / 6.48074069840786);
        o[4] = (short)((double)abs((int)o[2]) - 2 * 700 / 6.48074069840786);
        o[5] = (short)((double)abs((int)o[3]) - 2 * 700 / 6.48074069840786);
      }
      break;
    default: {
      float       re = -s.x, im = -s.y;
      const float o1 = 8.0f / 13.038404810405298f, o2 = 4.0f / 13.038404810405298f, o3 = 2.0f / 13.038404810405298f;
      o[0] = (short)(1000 * re); o[1] = (short)(1000 * im);
      re = fabsf(re) - o1; im = fabsf(im) - o1; o[2] = (short)(1000 * re); o[3] = (short)(1000 * im);
      re = fabsf(re) - o2; im = fabsf(im) - o2; o[4] = (short)(1000 * re); o[5] = (short)(1000 * im);
      re = fabsf(re) - o3; im = fabsf(im) - o3; o[6] = (short)(1000 * re); o[7] = (short)(1000 * im);
    }
  }
}

// ---- int8 -----------------------------------------------------------------
__device__ __forceinline__ void demod_b(int mod, cf32 s, int i, int nsym, signed char* o)
{
  const bool body = i < 8 * (nsym / 8); // SSE bodies take 8 symbols per step
  switch (mod) {
    case MOD_BPSK: o[0] = (signed char)((double)(-20.0f * (s.x + s.y)) / 1.4142135623730951); break;
    case MOD_QPSK: {
      const float g = (float)(-20 * 1.4142135623730951);
      // vector_simd.c:431-497: 16 floats per SSE step, scalar C cast for the rest
      o[0] = body ? (signed char)sat8(sat16(trunc_i(s.x * g))) : (signed char)trunc_i(s.x * g);
      o[1] = body ? (signed char)sat8(sat16(trunc_i(s.y * g))) : (signed char)trunc_i(s.y * g);
    } break;
    case MOD_16QAM:
      if (body) {
        signed char re = (signed char)sat8(sat16(rne(s.x * -30.0f))), im = (signed char)sat8(sat16(rne(s.y * -30.0f)));
        o[0] = re; o[1] = im;
        o[2] = (signed char)(abs8(re) - 18);
        o[3] = (signed char)(abs8(im) - 18);
      } else {
        short yre = (signed char)trunc_i(30 * s.x), yim = (signed char)trunc_i(30 * s.y);
        o[0] = (signed char)-yre; o[1] = (signed char)-yim;
        o[2] = (signed char)((double)abs((int)yre) - 2 * 30 / 3.1622776601683795);
        o[3] = (signed char)((double)abs((int)yim) - 2 * 30 / 3.1622776601683795);
      }
      break;
    case MOD_64QAM:
      if (body) {
        signed char re = (signed char)sat8(sat16(rne(s.x * -40.0f))), im = (signed char)sat8(sat16(rne(s.y * -40.0f)));
        signed char a1 = (signed char)(abs8(re) - 24), b1 = (signed char)(abs8(im) - 24);
        o[0] = re; o[1] = im; o[2] = a1; o[3] = b1;
        o[4] = (signed char)(abs8(a1) - 12);
        o[5] = (signed char)(abs8(b1) - 12);
      } else {
        float yre = (float)(signed char)trunc_i(40 * s.x), yim = (float)(signed char)trunc_i(40 * s.y);
        o[0] = (signed char)-yre; o[1] = (signed char)-yim;
        o[2] = (signed char)((double)abs((int)yre) - 4 * 40 / 6.48074069840786);
        o[3] = (signed char)((double)abs((int)yim) - 4 * 40 / 6.48074069840786);
        o[4] = (signed char)((double)abs((int)o[2]) - 2 * 40 / 6.48074069840786);
        o[5] = (signed char)((double)abs((int)o[3]) - 2 * 40 / 6.48074069840786);
      }
      break;
    default: {
      float       re = -s.x, im = -s.y;
      const float o1 = 8.0f / 13.038404810405298f, o2 = 4.0f / 13.038404810405298f, o3 = 2.0f / 13.038404810405298f;
      o[0] = (signed char)(50 * re); o[1] = (signed char)(50 * im);
      re = fabsf(re) - o1; im = fabsf(im) - o1; o[2] = (signed char)(50 * re); o[3] = (signed char)(50 * im);
      re = fabsf(re) - o2; im = fabsf(im) - o2; o[4] = (signed char)(50 * re); o[5] = (signed char)(50 * im);
      re = fabsf(re) - o3; im = fabsf(im) - o3; o[6] = (signed char)(50 * re); o[7] = (signed char)(50 * im);
    }
  }
}

__device__ __forceinline__ int mod_bits(int mod) { return mod == MOD_BPSK ? 1 : 2 * mod; }


} // namespace demod_dev
