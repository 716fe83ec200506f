/*
 * oracle/orc_uci.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 * HARQ-ACK (1 or 2 bits) multiplexed on the PUSCH, the one piece of UCI restated so far: 36.212 5.2.2.6/5.2.2.8 as
 * srslte_uci_encode_ack_ri / srslte_uci_decode_ack_ri (uci.c:497-520,:547-602,:627-656,:695-788), their use in srslte_ulsch_encode /
 * uci_decode_ri_ack (sch.c:929-966,:1170-1215) and the placeholder / repetition handling after scrambling in srslte_pusch_encode
 * (pusch.c:384-400). No RI, no CQI: the ACK symbols overwrite UL-SCH symbols, the rate matching is unchanged.
 */
#include "orc.h"
#include <math.h>
#include <string.h>

static const float BETA_HARQ[16] = {2.0f, 2.5f, 3.125f, 4.0f, 5.0f, 6.250f, 8.0f, 10.0f, 12.625f, 15.875f, 20.0f, 31.0f, 50.0f, 80.0f, 126.0f, -1.0f}; /* 36.213 Table 8.6.3-1 (sch.c:43-44) */

int orc_uci_ack_qprime(uint32_t O_ack, uint32_t I_offset_ack, uint32_t L_prb, uint32_t nof_symb, uint32_t K_segm)
{ /* Q_prime_ri_ack (uci.c:547-571) with UL-SCH present: min(ceil(O M_sc N_symb beta / sum K_r), 4 M_sc), float arithmetic */
  if (I_offset_ack > 15 || BETA_HARQ[I_offset_ack] < 0 || K_segm == 0) return -1;
  uint32_t x = (uint32_t)ceilf((float)O_ack * L_prb * 12 * nof_symb * BETA_HARQ[I_offset_ack] / K_segm);
  uint32_t m = 4 * L_prb * 12;
  return (int)(x < m ? x : m);
}

/* ACK symbol i (0 .. Q'-1) sits in row R-1-i/4 (sub-carrier) of column {2,3,8,9}[(3i)%4] (data symbol) of the interleaver matrix
   (uci.c:497-520; the extended-CP / shortened set {1,2,6,7} when there are at most 10 columns): first q-bit index of that symbol */
static uint32_t ack_symbol_qpos(uint32_t i, uint32_t Qm, uint32_t rows, uint32_t nof_symb)
{
  static const uint32_t norm[4] = {2, 3, 8, 9}, ext[4] = {1, 2, 6, 7};
  uint32_t row = rows - 1 - i / 4, col = nof_symb > 10 ? norm[(3 * i) % 4] : ext[(3 * i) % 4];
  return row * Qm + rows * col * Qm;
}

/* type of encoded bit e of the repeated ACK pattern (uci.c:573-602): 0 / 1 = that value, 2 = repetition of the previous bit, 3 = placeholder */
static int ack_bit_type(const uint8_t ack[2], uint32_t O_ack, uint32_t Qm, uint32_t e)
{
  if (O_ack == 1) {
    uint32_t r = e % Qm;
    return r == 0 ? ack[0] : (r == 1 ? 2 : 3);
  }
  uint32_t r = e % (3 * Qm), s = r / Qm, b = r % Qm;
  if (b >= 2) return 3;
  const uint8_t v[3] = {ack[0], ack[1], (uint8_t)(ack[0] ^ ack[1])};
  return v[(2 * s + b) % 3]; /* o0 o1 | o2 o0 | o1 o2 */
}

int orc_uci_ack_insert(uint8_t* q_bits /* one bit per byte, interleaved, scrambled */, const uint8_t* c_seq, const uint8_t ack[2], uint32_t O_ack,
                       uint32_t Qm, uint32_t nof_re, uint32_t nof_symb, uint32_t Qprime)
{ /* the net effect of sch.c:1203-1215 (ACK bits overwrite the interleaved stream), srslte_scrambling_bytes and pusch.c:386-400 on an
     already scrambled stream: value bits are scrambled, placeholders become 1, a repetition bit copies the transmitted bit before it */
  const uint32_t rows = nof_re / nof_symb;
  if (O_ack < 1 || O_ack > 2 || rows < 1 + (Qprime ? (Qprime - 1) / 4 : 0)) return -1;
  for (uint32_t i = 0; i < Qprime; i++) {
    const uint32_t p = ack_symbol_qpos(i, Qm, rows, nof_symb);
    for (uint32_t k = 0; k < Qm; k++) {
      const int t = ack_bit_type(ack, O_ack, Qm, i * Qm + k);
      q_bits[p + k] = t == 3 ? 1 : (t == 2 ? q_bits[p + k - 1] : (uint8_t)(t ^ c_seq[p + k]));
    }
  }
  return 0;
}

int orc_uci_ack_extract(int16_t* q_llr /* descrambled, interleaved order; ACK positions are zeroed */, const uint8_t* c_seq, uint8_t ack[2],
                        uint32_t O_ack, uint32_t Qm, uint32_t nof_re, uint32_t nof_symb, uint32_t Qprime)
{ /* srslte_uci_decode_ack_ri (uci.c:748-788) + "set zeros to HARQ bits" (sch.c:958-961). 1 bit: the value bit plus its repetition, whose
     descrambling is undone and redone with the value bit's scrambling bit (:627-640). 2 bits: triplets of symbols are combined when the
     loop index reaches the NEXT multiple of three, so the last complete triplet is only used if another symbol follows (:776-777) */
  const uint32_t rows = nof_re / nof_symb;
  int32_t        sum[3] = {0, 0, 0};
  if (O_ack < 1 || O_ack > 2) return -1;
  for (uint32_t i = 0; i < Qprime; i++) {
    if (O_ack == 2 && (i % 3 == 0) && i > 0) {
      uint32_t p[3];
      for (int s = 0; s < 3; s++) p[s] = ack_symbol_qpos(i - 3 + s, Qm, rows, nof_symb);
      sum[0] += q_llr[p[0]] + q_llr[p[1] + 1];
      sum[1] += q_llr[p[0] + 1] + q_llr[p[2]];
      sum[2] += q_llr[p[1]] + q_llr[p[2] + 1];
    } else if (O_ack == 1) {
      const uint32_t p0 = ack_symbol_qpos(i, Qm, rows, nof_symb), p1 = p0 + 1;
      q_llr[p1] = c_seq[p1] ? (int16_t)-q_llr[p1] : q_llr[p1];
      const int16_t q1 = c_seq[p0] ? (int16_t)-q_llr[p1] : q_llr[p1];
      sum[0] += q_llr[p0] + q1;
    }
  }
  ack[0] = sum[0] > 0;
  ack[1] = O_ack == 2 ? sum[1] > 0 : 0;
  for (uint32_t i = 0; i < Qprime; i++) {
    const uint32_t p = ack_symbol_qpos(i, Qm, rows, nof_symb);
    for (uint32_t k = 0; k < Qm; k++) q_llr[p + k] = 0;
  }
  return 0;
}
