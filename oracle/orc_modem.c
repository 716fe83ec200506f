/*
 * oracle/orc_modem.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 * CPU restatement of the LTE modulator tables (modem/lte_tables.c, mod.c; 36.211 7.1) and of the
 * max-log soft demapper srslte_demod_soft_demodulate{,_s,_b} (modem/demod_soft.c) including each
 * variant's rounding: SSE bodies use round-to-nearest-even + saturating packs and integer offsets,
 * their scalar tails truncate; QPSK uses cvtt (truncate) + saturating pack; 256QAM is float fold + C cast.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>

int orc_mod_bits(int mod)
{
  switch (mod) {
    case ORC_MOD_BPSK: return 1;
    case ORC_MOD_QPSK: return 2;
    case ORC_MOD_16QAM: return 4;
    case ORC_MOD_64QAM: return 6;
    case ORC_MOD_256QAM: return 8;
  }
  return -1;
}

static float pam_level(const uint8_t* b, int nb, double norm)
{ /* 36.211 7.1.x recursive form: (1-2b0)(2^(n-1) - (1-2b1)(2^(n-2) - ...)) / norm, bits b[0], b[2], ... of one axis */
  double v = 1.0;
  for (int i = nb - 1; i >= 1; i--) {
    v = (double)(1 << (nb - i)) - (1 - 2 * b[2 * i]) * v;
  }
  return (float)((1 - 2 * b[0]) * v / norm);
}

int orc_modulate(int mod, const uint8_t* bits, orc_cf_t* symbols, int nbits)
{ /* mod.c:33-60 + lte_tables.c:31-182 */
  int Qm = orc_mod_bits(mod);
  if (Qm < 0) return -1;
  int nsym = nbits / Qm;
  for (int i = 0; i < nsym; i++) {
    const uint8_t* b = &bits[i * Qm];
    switch (mod) {
      case ORC_MOD_BPSK: {
        float v = (float)((1 - 2 * b[0]) / sqrt(2));
        symbols[i] = (orc_cf_t){v, v};
      } break;
      case ORC_MOD_QPSK: symbols[i] = (orc_cf_t){pam_level(b, 1, sqrt(2)), pam_level(b + 1, 1, sqrt(2))}; break;
      case ORC_MOD_16QAM: symbols[i] = (orc_cf_t){pam_level(b, 2, sqrt(10)), pam_level(b + 1, 2, sqrt(10))}; break;
      case ORC_MOD_64QAM: symbols[i] = (orc_cf_t){pam_level(b, 3, sqrt(42)), pam_level(b + 1, 3, sqrt(42))}; break;
      default: symbols[i] = (orc_cf_t){pam_level(b, 4, sqrt(170)), pam_level(b + 1, 4, sqrt(170))}; break;
    }
  }
  return nsym;
}

/* ------------------------------------------------------------------ float demapper (demod_soft.c:58-62,79-88,228-238,414-436) */

int orc_demod_soft_f(int mod, const orc_cf_t* s, float* llr, int n)
{
  const float* x = (const float*)s;
  switch (mod) {
    case ORC_MOD_BPSK:
      for (int i = 0; i < n; i++) llr[i] = (float)(-(s[i].re + s[i].im) / sqrt(2));
      return 0;
    case ORC_MOD_QPSK: {
      float g = (float)-sqrt(2);
      for (int i = 0; i < 2 * n; i++) llr[i] = x[i] * g;
      return 0;
    }
    case ORC_MOD_16QAM:
      for (int i = 0; i < n; i++) {
        llr[4 * i + 0] = -s[i].re;
        llr[4 * i + 1] = -s[i].im;
        llr[4 * i + 2] = (float)(fabsf(s[i].re) - 2 / sqrt(10));
        llr[4 * i + 3] = (float)(fabsf(s[i].im) - 2 / sqrt(10));
      }
      return 0;
    case ORC_MOD_64QAM:
      for (int i = 0; i < n; i++) {
        llr[6 * i + 0] = -s[i].re;
        llr[6 * i + 1] = -s[i].im;
        llr[6 * i + 2] = (float)(fabsf(s[i].re) - 4 / sqrt(42));
        llr[6 * i + 3] = (float)(fabsf(s[i].im) - 4 / sqrt(42));
        llr[6 * i + 4] = (float)(fabsf(llr[6 * i + 2]) - 2 / sqrt(42));
        llr[6 * i + 5] = (float)(fabsf(llr[6 * i + 3]) - 2 / sqrt(42));
      }
      return 0;
    case ORC_MOD_256QAM:
      for (int i = 0; i < n; i++) {
        float re = -s[i].re, im = -s[i].im;
        const float o[3] = {8.0f / sqrtf(170.0f), 4.0f / sqrtf(170.0f), 2.0f / sqrtf(170.0f)};
        llr[8 * i] = re; llr[8 * i + 1] = im;
        for (int j = 0; j < 3; j++) {
          re = fabsf(re) - o[j]; im = fabsf(im) - o[j];
          llr[8 * i + 2 + 2 * j] = re; llr[8 * i + 3 + 2 * j] = im;
        }
      }
      return 0;
  }
  return -1;
}

/* ------------------------------------------------------------------ helpers mimicking the SSE conversions */

static inline int32_t cvt_rne(float v)
{ /* _mm_cvtps_epi32: round to nearest even, 0x80000000 when out of range / NaN */
  if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT32_MIN;
  return (int32_t)nearbyintf(v);
}
static inline int32_t cvt_trunc(float v)
{ /* _mm_cvttps_epi32 */
  if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT32_MIN;
  return (int32_t)v;
}
static inline int16_t pack16(int32_t v) { return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); }
static inline int8_t  pack8(int32_t v) { return (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v)); }
static inline int16_t abs16(int16_t v) { return (int16_t)(v < 0 ? -v : v); } /* _mm_abs_epi16: -32768 stays */
static inline int8_t  abs8(int8_t v) { return (int8_t)(v < 0 ? -v : v); }

/* ------------------------------------------------------------------ int16 demapper */

int orc_demod_soft_s(int mod, const orc_cf_t* s, int16_t* llr, int n)
{
  const float* x = (const float*)s;
  switch (mod) {
    case ORC_MOD_BPSK: /* demod_soft.c:52-56: (short) of a double expression */
      for (int i = 0; i < n; i++) llr[i] = (int16_t)(-100 * (s[i].re + s[i].im) / sqrt(2));
      return 0;
    case ORC_MOD_QPSK: { /* demod_soft.c:68-70 -> vector_simd.c:392-427: x*scale, cvtt, packs (body) / C cast (tail < 16) */
      float g    = (float)(-100 * sqrt(2));
      int   len  = 2 * n, body = len - (len % 16);
      for (int i = 0; i < body; i++) llr[i] = pack16(cvt_trunc(x[i] * g));
      for (int i = body; i < len; i++) llr[i] = (int16_t)(x[i] * g);
      return 0;
    }
    case ORC_MOD_16QAM: { /* demod_soft.c:90-133 */
      int16_t off = (int16_t)(2 * 400 / sqrt(10));
      int     i   = 0;
      for (; i < 4 * (n / 4); i++) {
        int16_t re = pack16(cvt_rne(s[i].re * -400.0f)), im = pack16(cvt_rne(s[i].im * -400.0f));
        llr[4 * i + 0] = re;
        llr[4 * i + 1] = im;
        llr[4 * i + 2] = (int16_t)(abs16(re) - off);
        llr[4 * i + 3] = (int16_t)(abs16(im) - off);
      }
      for (; i < n; i++) {
        short yre = (short)(400 * s[i].re), yim = (short)(400 * s[i].im);
        llr[4 * i + 0] = (int16_t)-yre;
        llr[4 * i + 1] = (int16_t)-yim;
        llr[4 * i + 2] = (int16_t)(abs(yre) - 2 * 400 / sqrt(10));
        llr[4 * i + 3] = (int16_t)(abs(yim) - 2 * 400 / sqrt(10));
      }
      return 0;
    }
    case ORC_MOD_64QAM: { /* demod_soft.c:240-301 */
      int16_t o1 = (int16_t)(4 * 700 / sqrt(42)), o2 = (int16_t)(2 * 700 / sqrt(42));
      int     i  = 0;
      for (; i < 4 * (n / 4); i++) {
        int16_t re = pack16(cvt_rne(s[i].re * -700.0f)), im = pack16(cvt_rne(s[i].im * -700.0f));
        int16_t a1 = (int16_t)(abs16(re) - o1), b1 = (int16_t)(abs16(im) - o1);
        llr[6 * i + 0] = re; llr[6 * i + 1] = im; llr[6 * i + 2] = a1; llr[6 * i + 3] = b1;
        llr[6 * i + 4] = (int16_t)(abs16(a1) - o2);
        llr[6 * i + 5] = (int16_t)(abs16(b1) - o2);
      }
      for (; i < n; i++) {
        float yre = (short)(700 * s[i].re), yim = (short)(700 * s[i].im);
        llr[6 * i + 0] = (int16_t)-yre;
        llr[6 * i + 1] = (int16_t)-yim;
        llr[6 * i + 2] = (int16_t)(abs((int)yre) - 4 * 700 / sqrt(42));
        llr[6 * i + 3] = (int16_t)(abs((int)yim) - 4 * 700 / sqrt(42));
        llr[6 * i + 4] = (int16_t)(abs(llr[6 * i + 2]) - 2 * 700 / sqrt(42));
        llr[6 * i + 5] = (int16_t)(abs(llr[6 * i + 3]) - 2 * 700 / sqrt(42));
      }
      return 0;
    }
    case ORC_MOD_256QAM: /* demod_soft.c:457-477 */
      for (int i = 0; i < n; i++) {
        float       re = -s[i].re, im = -s[i].im;
        const float o[3] = {8.0f / sqrtf(170.0f), 4.0f / sqrtf(170.0f), 2.0f / sqrtf(170.0f)};
        llr[8 * i] = (int16_t)(1000 * re); llr[8 * i + 1] = (int16_t)(1000 * im);
        for (int j = 0; j < 3; j++) {
          re = fabsf(re) - o[j]; im = fabsf(im) - o[j];
          llr[8 * i + 2 + 2 * j] = (int16_t)(1000 * re); llr[8 * i + 3 + 2 * j] = (int16_t)(1000 * im);
        }
      }
      return 0;
  }
  return -1;
}

/* ------------------------------------------------------------------ int8 demapper */

int orc_demod_soft_b(int mod, const orc_cf_t* s, int8_t* llr, int n)
{
  const float* x = (const float*)s;
  switch (mod) {
    case ORC_MOD_BPSK: /* demod_soft.c:46-50 */
      for (int i = 0; i < n; i++) llr[i] = (int8_t)(-20 * (s[i].re + s[i].im) / sqrt(2));
      return 0;
    case ORC_MOD_QPSK: { /* demod_soft.c:64-66 -> vector_simd.c:431-497: 16 floats per step, cvtt + packs32 + packs16 */
      float g   = (float)(-20 * sqrt(2));
      int   len = 2 * n, body = len - (len % 16);
      for (int i = 0; i < body; i++) llr[i] = pack8(pack16(cvt_trunc(x[i] * g)));
      for (int i = body; i < len; i++) llr[i] = (int8_t)(x[i] * g);
      return 0;
    }
    case ORC_MOD_16QAM: { /* demod_soft.c:135-186 */
      int8_t off = (int8_t)(2 * 30 / sqrt(10));
      int    i   = 0;
      for (; i < 8 * (n / 8); i++) {
        int8_t re = pack8(pack16(cvt_rne(s[i].re * -30.0f))), im = pack8(pack16(cvt_rne(s[i].im * -30.0f)));
        llr[4 * i + 0] = re;
        llr[4 * i + 1] = im;
        llr[4 * i + 2] = (int8_t)(abs8(re) - off);
        llr[4 * i + 3] = (int8_t)(abs8(im) - off);
      }
      for (; i < n; i++) {
        short yre = (int8_t)(30 * s[i].re), yim = (int8_t)(30 * s[i].im);
        llr[4 * i + 0] = (int8_t)-yre;
        llr[4 * i + 1] = (int8_t)-yim;
        llr[4 * i + 2] = (int8_t)(abs(yre) - 2 * 30 / sqrt(10));
        llr[4 * i + 3] = (int8_t)(abs(yim) - 2 * 30 / sqrt(10));
      }
      return 0;
    }
    case ORC_MOD_64QAM: { /* demod_soft.c:303-392 */
      int8_t o1 = (int8_t)(4 * 40 / sqrt(42)), o2 = (int8_t)(2 * 40 / sqrt(42));
      int    i  = 0;
      for (; i < 8 * (n / 8); i++) {
        int8_t re = pack8(pack16(cvt_rne(s[i].re * -40.0f))), im = pack8(pack16(cvt_rne(s[i].im * -40.0f)));
        int8_t a1 = (int8_t)(abs8(re) - o1), b1 = (int8_t)(abs8(im) - o1);
        llr[6 * i + 0] = re; llr[6 * i + 1] = im; llr[6 * i + 2] = a1; llr[6 * i + 3] = b1;
        llr[6 * i + 4] = (int8_t)(abs8(a1) - o2);
        llr[6 * i + 5] = (int8_t)(abs8(b1) - o2);
      }
      for (; i < n; i++) {
        float yre = (int8_t)(40 * s[i].re), yim = (int8_t)(40 * s[i].im);
        llr[6 * i + 0] = (int8_t)-yre;
        llr[6 * i + 1] = (int8_t)-yim;
        llr[6 * i + 2] = (int8_t)(abs((int)yre) - 4 * 40 / sqrt(42));
        llr[6 * i + 3] = (int8_t)(abs((int)yim) - 4 * 40 / sqrt(42));
        llr[6 * i + 4] = (int8_t)(abs(llr[6 * i + 2]) - 2 * 40 / sqrt(42));
        llr[6 * i + 5] = (int8_t)(abs(llr[6 * i + 3]) - 2 * 40 / sqrt(42));
      }
      return 0;
    }
    case ORC_MOD_256QAM: /* demod_soft.c:435-455 */
      for (int i = 0; i < n; i++) {
        float       re = -s[i].re, im = -s[i].im;
        const float o[3] = {8.0f / sqrtf(170.0f), 4.0f / sqrtf(170.0f), 2.0f / sqrtf(170.0f)};
        llr[8 * i] = (int8_t)(50 * re); llr[8 * i + 1] = (int8_t)(50 * im);
        for (int j = 0; j < 3; j++) {
          re = fabsf(re) - o[j]; im = fabsf(im) - o[j];
          llr[8 * i + 2 + 2 * j] = (int8_t)(50 * re); llr[8 * i + 3 + 2 * j] = (int8_t)(50 * im);
        }
      }
      return 0;
  }
  return -1;
}
