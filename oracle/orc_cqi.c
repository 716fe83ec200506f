/*
 * oracle/orc_cqi.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 * CQI / PMI report multiplexed on the PUSCH (36.212 5.2.2.6, 5.2.2.6.4): what srslte_uci_encode_cqi_pusch /
 * srslte_uci_decode_cqi_pusch do (lib/src/phy/phch/uci.c:264-494), and the pieces they call:
 *   up to 11 bits : (32, O) block code, repeated to Q' Qm bits (encode_cqi_short :283-302); decoded by adding up the repetitions and
 *                   correlating with all 2^O code words (decode_cqi_short :305-341)
 *   above 11 bits : CRC-8, tail-biting rate-1/3 convolutional code (fec/convcoder.c:43-72), rate matching (fec/rm_conv.c:44-89);
 *                   decoded by srslte_rm_conv_rx_s (:160-219), the 16-bit AVX2 Viterbi decoder an x86 build selects
 *                   (fec/viterbi37_avx2_16bit.c, three repetitions of the frame for the tail biting, fec/viterbi.c:129-152), then the CRC.
 *                   NOTE: the reference's int16 wrapper in front of that decoder overflows (see orc_uci_cqi_decode): the long report is
 *                   decoded with the quantisation of the reference's float entry point instead, and pinned through that entry point.
 * The coded bits sit in front of the UL-SCH bits in the stream the channel interleaver reads (sch.c:1133-1160,:1031-1060).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const float BETA_CQI[16] = {-1.0f, -1.0f, 1.125f, 1.25f, 1.375f, 1.625f, 1.750f, 2.0f, 2.25f, 2.5f, 2.875f, 3.125f, 3.5f, 4.0f, 5.0f, 6.25f}; /* sch.c:51-52 */

/* 36.212 Table 5.2.2.6.4-1, rows as 11-bit masks: bit n = M[i][n] */
static const uint16_t M32[32] = {0x403, 0x607, 0x749, 0x50D, 0x48F, 0x5D3, 0x755, 0x599, 0x69B, 0x65D, 0x6E5, 0x567, 0x7A9, 0x6AB, 0x4B1, 0x6F3,
                                 0x277, 0x139, 0x0FB, 0x061, 0x445, 0x60B, 0x591, 0x717, 0x3DF, 0x4E3, 0x32D, 0x3AF, 0x175, 0x1FD, 0x7FF, 0x001};

int orc_uci_cqi_qprime(uint32_t O_cqi, uint32_t I_offset_cqi, uint32_t L_prb, uint32_t nof_symb, uint32_t K_segm, uint32_t Qprime_ri)
{ /* Q_prime_cqi (uci.c:264-281): min(ceil((O + L) M_sc N_symb beta / sum K_r), M_sc N_symb - Q'_ri), L = 8 CRC bits above 11 (the
     reference tests O < 11 here, so an 11-bit report is sized with the CRC it does not carry) */
  if (I_offset_cqi > 15 || BETA_CQI[I_offset_cqi] < 0) return -1;
  uint32_t L = O_cqi < 11 ? 0 : 8, x = 999999;
  if (K_segm > 0) x = (uint32_t)ceilf((float)(O_cqi + L) * L_prb * 12 * nof_symb * BETA_CQI[I_offset_cqi] / K_segm);
  uint32_t m = L_prb * 12 * nof_symb - Qprime_ri;
  return (int)(x < m ? x : m);
}

static int parity32(uint32_t v)
{
  v ^= v >> 16; v ^= v >> 8; v ^= v >> 4; v ^= v >> 2; v ^= v >> 1;
  return (int)(v & 1);
}

static uint32_t crc8(const uint8_t* bits, uint32_t n)
{ /* srslte_crc_checksum with SRSLTE_LTE_CRC8 = 0x19B (phy_common.h:61) on one-bit-per-byte data */
  uint32_t r = 0;
  for (uint32_t i = 0; i < n + 8; i++) {
    r = (r << 1) | (i < n ? (bits[i] & 1) : 0);
    if (r & 0x100) r ^= 0x19B;
  }
  return r & 0xff;
}

static const uint8_t PERM_CC[32] = {1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31, 0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30};
static const uint8_t PERM_CC_INV[32] = {16, 0, 24, 8, 20, 4, 28, 12, 18, 2, 26, 10, 22, 6, 30, 14, 17, 1, 25, 9, 21, 5, 29, 13, 19, 3, 27, 11, 23, 7, 31, 15};
static const int     POLY[3] = {0x6D, 0x4F, 0x57};

int orc_uci_cqi_encode(const uint8_t* cqi /* one bit per byte */, uint32_t O, uint8_t* q_bits, uint32_t Q)
{
  if (O == 0 || O > 64) return -1;
  if (O <= 11) { /* encode_cqi_short: word w = bits MSB first (srslte_bit_pack), code bit i = parity of the set bits n of M[i][n] */
    uint32_t cw = 0;
    for (uint32_t i = 0; i < 32; i++) {
      int b = 0;
      for (uint32_t n = 0; n < O; n++) b ^= cqi[n] & (M32[i] >> n) & 1;
      cw |= (uint32_t)b << i;
    }
    for (uint32_t i = 0; i < Q; i++) q_bits[i] = (cw >> (i % 32)) & 1;
    return 0;
  }
  /* encode_cqi_long */
  const uint32_t F = O + 8;
  uint8_t        in[72], enc[3 * 72];
  memcpy(in, cqi, O);
  uint32_t c = crc8(cqi, O);
  for (int i = 0; i < 8; i++) in[O + i] = (c >> (7 - i)) & 1;
  uint32_t sr = 0;
  for (uint32_t i = F - 6; i < F; i++) sr = (sr << 1) | in[i]; /* tail biting: the register starts with the last K-1 bits */
  for (uint32_t i = 0; i < F; i++) {
    sr = (sr << 1) | in[i];
    for (int j = 0; j < 3; j++) enc[3 * i + j] = (uint8_t)parity32(sr & (uint32_t)POLY[j]);
  }
  /* srslte_rm_conv_tx */
  const int nrows = (int)((F - 1) / 32 + 1), K_p = nrows * 32, ndummy = K_p - (int)F;
  uint8_t   tmp[3 * 32 * 4];
  int       k = 0;
  for (int s = 0; s < 3; s++) {
    for (int j = 0; j < 32; j++) {
      for (int i = 0; i < nrows; i++) {
        const int pos = i * 32 + PERM_CC[j];
        tmp[k++]      = pos < ndummy ? 100 : enc[(pos - ndummy) * 3 + s];
      }
    }
  }
  uint32_t o = 0;
  int      j = 0;
  while (o < Q) {
    if (tmp[j] != 100) q_bits[o++] = tmp[j];
    if (++j == 3 * K_p) j = 0;
  }
  return 0;
}

/* The 16-bit AVX2 Viterbi decoder an x86 build selects (VITERBI_16, viterbi.c:41-46; fec/viterbi37_avx2_16bit.c:196-330) on quantised
 * soft bits us[3 F] (0 = certain 0, 65535 = certain 1), tail biting by three repetitions of the frame (viterbi.c:129-152):
 * branch metric = avg(avg(b0 ^ s0, b1 ^ s1), b2 ^ s2) >> 3 with _mm256_avg_epu16 (so the third symbol weighs 1/2, the others 1/4),
 * 16-bit wrapping path metrics compared modulo 2^16; the normalisation never subtracts anything (its reduction shifts a 128-bit lane
 * by 16 bytes, :287, which clears it: the minimum it finds is 0); the best end state is the LAST one with the smallest metric;
 * the chain-back reads decisions 6 steps past the ones written, which are zero (:143). */
static void viterbi37_16_tb(const uint16_t* us, uint32_t F, uint8_t* out)
{
  const uint32_t steps = 3 * F;
  uint16_t       old[64], nw[64], bt[3][32];
  uint64_t*      dec = calloc(steps + 6, sizeof(uint64_t));
  for (int s = 0; s < 32; s++) {
    for (int p = 0; p < 3; p++) bt[p][s] = parity32((uint32_t)(2 * s) & (uint32_t)POLY[p]) ? 65535 : 0;
  }
  for (int i = 0; i < 64; i++) old[i] = 63;
  for (uint32_t t = 0; t < steps; t++) {
    const uint16_t* s3 = &us[3 * (t % F)];
    uint64_t        d  = 0;
    for (int i = 0; i < 32; i++) {
      const unsigned a = bt[0][i] ^ s3[0], b = bt[1][i] ^ s3[1], c = bt[2][i] ^ s3[2];
      const unsigned m01 = (a + b + 1) >> 1, met = ((c + m01 + 1) >> 1) >> 3, mm = 8191 - met;
      const uint16_t m0 = (uint16_t)(old[i] + met), m1 = (uint16_t)(old[32 + i] + mm), m2 = (uint16_t)(old[i] + mm), m3 = (uint16_t)(old[32 + i] + met);
      const int      d0 = (int16_t)(uint16_t)(m0 - m1) > 0, d1 = (int16_t)(uint16_t)(m2 - m3) > 0;
      nw[2 * i]     = d0 ? m1 : m0;
      nw[2 * i + 1] = d1 ? m3 : m2;
      d |= (uint64_t)d0 << (2 * i) | (uint64_t)d1 << (2 * i + 1);
    }
    dec[t] = d;
    memcpy(old, nw, sizeof(old));
  }
  uint32_t best = 0;
  uint16_t mn   = 65535;
  for (uint32_t i = 0; i < 64; i++) {
    if (old[i] <= mn) { best = i; mn = old[i]; }
  }
  uint32_t endstate = (best % 64) << 2;
  uint8_t* all      = malloc(steps);
  for (int n = (int)steps - 1; n >= 0; n--) { /* chainback_viterbi37_avx2_16bit (:120-152) */
    const uint32_t k = (uint32_t)(dec[6 + n] >> (endstate >> 2)) & 1;
    endstate         = (endstate >> 1) | (k << 7);
    all[n]           = (uint8_t)k;
  }
  memcpy(out, &all[F], F); /* the middle repetition (viterbi.c:150) */
  free(all);
  free(dec);
}

/* srslte_viterbi_decode_f (viterbi.c:518-548) for the tail-biting K = 7 rate-1/3 code: soft bits scaled by gain / max |.| (gain =
 * DEFAULT_GAIN_16 = 1000), offset 32767.5, clipped to 0..65535 (srslte_vec_quant_fus, vector.c:401-413), then the decoder above. */
void orc_viterbi37_tb_f(const float* sym, uint32_t F, uint8_t* out)
{
  const uint32_t len = 3 * F;
  float          mx  = -9e9f;
  for (uint32_t i = 0; i < len; i++) mx = fabsf(sym[i]) > mx ? fabsf(sym[i]) : mx;
  const float gain = 1000.0f / mx;
  uint16_t*   us   = malloc(len * sizeof(uint16_t));
  for (uint32_t i = 0; i < len; i++) {
    long t = (long)fmaf(gain, sym[i], 32767.5f); /* the -Ofast -mfma build contracts offset + gain * in */
    us[i]  = (uint16_t)(t < 0 ? 0 : (t > 65535 ? 65535 : t));
  }
  viterbi37_16_tb(us, F, out);
  free(us);
}

int orc_uci_cqi_decode(int16_t* q_llr /* Q LLRs; the short decoder accumulates into the first 32 */, uint32_t Q, uint32_t O, uint8_t* cqi, uint8_t* crc_ok)
{
  if (O == 0 || O > 64) return -1;
  *crc_ok = 1;
  if (O <= 11) { /* decode_cqi_short (uci.c:305-341): srslte_vec_sum_sss wraps; the first word of the highest correlation wins */
    if (Q > 32) {
      uint32_t i = 1;
      for (; i < Q / 32; i++) {
        for (int j = 0; j < 32; j++) q_llr[j] = (int16_t)(q_llr[j] + q_llr[i * 32 + j]);
      }
      for (uint32_t j = 0; j < Q % 32; j++) q_llr[j] = (int16_t)(q_llr[j] + q_llr[i * 32 + j]);
    }
    const uint32_t n = Q < 32 ? Q : 32;
    uint32_t       best = 0;
    int32_t        mc = INT32_MIN;
    for (uint32_t w = 0; w < (1u << O); w++) {
      int32_t corr = 0;
      for (uint32_t i = 0; i < n; i++) { /* word w: bit n of the report = bit (O-1-n) of w (srslte_bit_unpack) */
        int b = 0;
        for (uint32_t k = 0; k < O; k++) b ^= ((w >> (O - 1 - k)) & 1) & ((M32[i] >> k) & 1);
        corr += b ? q_llr[i] : -q_llr[i];
      }
      if (corr > mc) { mc = corr; best = w; }
    }
    for (uint32_t k = 0; k < O; k++) cqi[k] = (best >> (O - 1 - k)) & 1;
    return 0;
  }
  const uint32_t F = O + 8;
  const int      nrows = (int)((F - 1) / 32 + 1), K_p = nrows * 32, ndummy = K_p - (int)F;
  int16_t        tmp[3 * 32 * 4], dem[3 * 72];
  for (int i = 0; i < 3 * K_p; i++) tmp[i] = 10000; /* SRSLTE_RX_NULL: also what an accumulated value of 10000 is taken for */
  uint32_t k = 0;
  int      j = 0;
  while (k < Q) {
    const int d_i = (j % K_p) / nrows, d_j = (j % K_p) % nrows;
    if (d_j * 32 + PERM_CC[d_i] >= ndummy) {
      if (tmp[j] == 10000) {
        tmp[j] = q_llr[k];
      } else if (q_llr[k] != 10000) {
        tmp[j] = (int16_t)(tmp[j] + q_llr[k]);
      }
      k++;
    }
    if (++j == 3 * K_p) j = 0;
  }
  for (uint32_t i = 0; i < F; i++) {
    const int d_i = (int)(i + ndummy) / 32, d_j = (int)(i + ndummy) % 32;
    for (int s = 0; s < 3; s++) {
      const int16_t o = tmp[K_p * s + PERM_CC_INV[d_j] * nrows + d_i];
      dem[i * 3 + s]  = o != 10000 ? o : 0;
    }
  }
  /* The reference hands these int16 values to srslte_viterbi_decode_s, whose 16-bit branch quantises with (int16_t)(32767 + in)
     (srslte_vec_quant_sus, vector.c:443-453, viterbi.c:571): that overflows for every positive soft bit, which then reads as a certain 0,
     and the decoder returns all zeros on clean input (tests/test_oracle_vs_ref.py::test_reference_viterbi_decode_s_loses_positive_soft_bits).
     Nothing upstream exercises this path (pusch_test only sends the 4-bit wide-band report). The report is decoded here the way the
     reference's float entry point does it - same decoder, soft bits scaled to the 16-bit range - which is pinned. */
  uint8_t bits[72];
  float   demf[3 * 72];
  for (uint32_t i = 0; i < 3 * F; i++) demf[i] = (float)dem[i];
  orc_viterbi37_tb_f(demf, F, bits);
  *crc_ok = crc8(bits, O) == (uint32_t)((bits[O] << 7) | (bits[O + 1] << 6) | (bits[O + 2] << 5) | (bits[O + 3] << 4) | (bits[O + 4] << 3) | (bits[O + 5] << 2) |
                                        (bits[O + 6] << 1) | bits[O + 7]);
  if (*crc_ok) memcpy(cqi, bits, O); /* decode_cqi_long copies the report out only when the CRC matches (uci.c:408-413) */
  return 0;
}
