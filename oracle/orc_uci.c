/*
 * oracle/orc_uci.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 * HARQ-ACK (1 or 2 bits) multiplexed on the PUSCH, the one piece of UCI restated so far: 36.212 5.2.2.6/5.2.2.8 as
 * srslte_uci_encode_ack_ri / srslte_uci_decode_ack_ri (uci.c:497-520,:547-602,:627-656,:695-788), their use in srslte_ulsch_encode /
 * uci_decode_ri_ack (sch.c:929-966,:1170-1215) and the placeholder / repetition handling after scrambling in srslte_pusch_encode
 * (pusch.c:384-400). The ACK symbols overwrite UL-SCH symbols, the rate matching is unchanged. The rank indication (1 or 2 bits,
 * sch.c:968-979,:1110-1129) uses the same encoder on its own columns, but its symbols are left out by the channel interleaver
 * (ulsch_interleave_gen, sch.c:580-598) and the UL-SCH is rate-matched to what remains. No CQI.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const float BETA_HARQ[16] = {2.0f, 2.5f, 3.125f, 4.0f, 5.0f, 6.250f, 8.0f, 10.0f, 12.625f, 15.875f, 20.0f, 31.0f, 50.0f, 80.0f, 126.0f, -1.0f}; /* 36.213 Table 8.6.3-1 (sch.c:43-44) */

int orc_uci_ack_qprime(uint32_t O_ack, uint32_t I_offset_ack, uint32_t L_prb, uint32_t nof_symb, uint32_t K_segm)
{ /* Q_prime_ri_ack (uci.c:547-571) with UL-SCH present: min(ceil(O M_sc N_symb beta / sum K_r), 4 M_sc), float arithmetic */
  if (I_offset_ack > 15 || BETA_HARQ[I_offset_ack] < 0 || K_segm == 0) return -1;
  uint32_t x = (uint32_t)ceilf((float)O_ack * L_prb * 12 * nof_symb * BETA_HARQ[I_offset_ack] / K_segm);
  uint32_t m = 4 * L_prb * 12;
  return (int)(x < m ? x : m);
}

/* The same WITHOUT UL-SCH data (a CQI-only PUSCH, grant.tb.tbs == 0; 36.212 5.2.4.1): the callers divide the offset by the CQI's
   (sch.c:943-946,:970-973 on the receive side, :1111-1114,:1171-1174 on the transmit side) and Q_prime_ri_ack takes the CQI report's
   size - with its 8 CRC bits above 11 - in the place of sum K_r (uci.c:557-564). is_ri: the rank indication's offset table. */
static const float BETA_CQI_[16] = {-1.0f, -1.0f, 1.125f, 1.25f, 1.375f, 1.625f, 1.750f, 2.0f, 2.25f, 2.5f, 2.875f, 3.125f, 3.5f, 4.0f, 5.0f, 6.25f}; /* sch.c:51-52 */
static const float BETA_RI_[16] = {1.25f, 1.625f, 2.0f, 2.5f, 3.125f, 4.0f, 5.0f, 6.25f, 8.0f, 10.0f, 12.625f, 15.875f, 20.0f, -1.0f, -1.0f, -1.0f};
int orc_uci_ack_ri_qprime_nodata(uint32_t O, uint32_t I_offset, int is_ri, uint32_t O_cqi, uint32_t I_offset_cqi, uint32_t L_prb, uint32_t nof_symb)
{
  const float* tab = is_ri ? BETA_RI_ : BETA_HARQ;
  if (I_offset > 15 || I_offset_cqi > 15 || tab[I_offset] < 0 || BETA_CQI_[I_offset_cqi] < 0 || O_cqi == 0) return -1;
  float beta = tab[I_offset];
  beta /= BETA_CQI_[I_offset_cqi];
  uint32_t K = O_cqi <= 11 ? O_cqi : O_cqi + 8;
  uint32_t x = (uint32_t)ceilf((float)O * L_prb * 12 * nof_symb * beta / K);
  uint32_t m = 4 * L_prb * 12;
  return (int)(x < m ? x : m);
}

/* ACK symbol i (0 .. Q'-1) sits in row R-1-i/4 (sub-carrier) of column {2,3,8,9}[(3i)%4] (data symbol) of the interleaver matrix
   (uci.c:497-520; the extended-CP / shortened set {1,2,6,7} when there are at most 10 columns): first q-bit index of that symbol */
static int g_is_ri = 0; /* which column set the helpers below use (set by the entry points; the oracle is single-threaded) */
static uint32_t ack_symbol_qpos(uint32_t i, uint32_t Qm, uint32_t rows, uint32_t nof_symb)
{
  static const uint32_t norm[4] = {2, 3, 8, 9}, ext[4] = {1, 2, 6, 7};
  static const uint32_t ri_norm[4] = {1, 4, 7, 10}, ri_ext[4] = {0, 3, 5, 8}; /* uci.c:525-526 */
  uint32_t row = rows - 1 - i / 4, c = (3 * i) % 4;
  uint32_t col = g_is_ri ? (nof_symb > 10 ? ri_norm[c] : ri_ext[c]) : (nof_symb > 10 ? norm[c] : ext[c]);
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

static int uci_insert(uint8_t* q_bits, const uint8_t* c_seq, const uint8_t ack[2], uint32_t O_ack, uint32_t Qm, uint32_t rows, uint32_t nof_symb,
                      uint32_t Qprime);

int orc_uci_ack_insert(uint8_t* q_bits /* one bit per byte, interleaved, scrambled */, const uint8_t* c_seq, const uint8_t ack[2], uint32_t O_ack,
                       uint32_t Qm, uint32_t nof_re, uint32_t nof_symb, uint32_t Qprime)
{ /* the net effect of sch.c:1203-1215 (ACK bits overwrite the interleaved stream), srslte_scrambling_bytes and pusch.c:386-400 on an
     already scrambled stream: value bits are scrambled, placeholders become 1, a repetition bit copies the transmitted bit before it */
  const uint32_t rows = nof_re / nof_symb;
  g_is_ri = 0;
  return uci_insert(q_bits, c_seq, ack, O_ack, Qm, rows, nof_symb, Qprime);
}

static int uci_insert(uint8_t* q_bits, const uint8_t* c_seq, const uint8_t ack[2], uint32_t O_ack, uint32_t Qm, uint32_t rows, uint32_t nof_symb,
                      uint32_t Qprime)
{
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

static int uci_extract(int16_t* q_llr, const uint8_t* c_seq, uint8_t ack[2], uint32_t O_ack, uint32_t Qm, uint32_t rows, uint32_t nof_symb,
                       uint32_t Qprime, int zero);

int orc_uci_ack_extract(int16_t* q_llr /* descrambled, interleaved order; ACK positions are zeroed */, const uint8_t* c_seq, uint8_t ack[2],
                        uint32_t O_ack, uint32_t Qm, uint32_t nof_re, uint32_t nof_symb, uint32_t Qprime)
{ /* srslte_uci_decode_ack_ri (uci.c:748-788) + "set zeros to HARQ bits" (sch.c:958-961). 1 bit: the value bit plus its repetition, whose
     descrambling is undone and redone with the value bit's scrambling bit (:627-640). 2 bits: triplets of symbols are combined when the
     loop index reaches the NEXT multiple of three, so the last complete triplet is only used if another symbol follows (:776-777) */
  g_is_ri = 0;
  return uci_extract(q_llr, c_seq, ack, O_ack, Qm, nof_re / nof_symb, nof_symb, Qprime, 1);
}

static int uci_extract(int16_t* q_llr, const uint8_t* c_seq, uint8_t ack[2], uint32_t O_ack, uint32_t Qm, uint32_t rows, uint32_t nof_symb,
                       uint32_t Qprime, int zero)
{
  int32_t sum[3] = {0, 0, 0};
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
  for (uint32_t i = 0; zero && i < Qprime; i++) {
    const uint32_t p = ack_symbol_qpos(i, Qm, rows, nof_symb);
    for (uint32_t k = 0; k < Qm; k++) q_llr[p + k] = 0;
  }
  return 0;
}

/* ---- rank indication on the PUSCH */
static const float BETA_RI[16] = {1.25f, 1.625f, 2.0f, 2.5f, 3.125f, 4.0f, 5.0f, 6.25f, 8.0f, 10.0f, 12.625f, 15.875f, 20.0f, -1.0f, -1.0f, -1.0f}; /* 36.213 Table 8.6.3-2 (sch.c:47-48) */

int orc_uci_ri_qprime(uint32_t O_ri, uint32_t I_offset_ri, uint32_t L_prb, uint32_t nof_symb, uint32_t K_segm)
{ /* Q_prime_ri_ack (uci.c:547-571) with beta_ri_offset (sch.c:1111) */
  if (I_offset_ri > 15 || BETA_RI[I_offset_ri] < 0 || K_segm == 0) return -1;
  uint32_t x = (uint32_t)ceilf((float)O_ri * L_prb * 12 * nof_symb * BETA_RI[I_offset_ri] / K_segm);
  uint32_t m = 4 * L_prb * 12;
  return (int)(x < m ? x : m);
}

int orc_uci_ri_insert(uint8_t* q_bits, const uint8_t* c_seq, const uint8_t ri[2], uint32_t O_ri, uint32_t Qm, uint32_t nof_re, uint32_t nof_symb,
                      uint32_t Qprime)
{ /* as orc_uci_ack_insert on the RI columns (uci.c:521-545): the interleaver left these symbols free */
  g_is_ri = 1;
  int r = uci_insert(q_bits, c_seq, ri, O_ri, Qm, nof_re / nof_symb, nof_symb, Qprime);
  g_is_ri = 0;
  return r;
}

int orc_uci_ri_extract(int16_t* q_llr, const uint8_t* c_seq, uint8_t ri[2], uint32_t O_ri, uint32_t Qm, uint32_t nof_re, uint32_t nof_symb,
                       uint32_t Qprime)
{ /* srslte_uci_decode_ack_ri with is_ri (sch.c:968-979): the LLRs stay (no zeroing), the 1-bit repetition LLRs keep their re-scrambled sign */
  g_is_ri = 1;
  int r = uci_extract(q_llr, c_seq, ri, O_ri, Qm, nof_re / nof_symb, nof_symb, Qprime, 0);
  g_is_ri = 0;
  return r;
}

int orc_ulsch_interleaver_lut(uint32_t Qm, uint32_t nof_re, uint32_t nof_symb, uint32_t Qprime_ri, uint32_t* lut /* [nof_re * Qm] */)
{ /* ulsch_interleave_gen (sch.c:580-598): lut[q index] = g index, the matrix (rows = sub-carriers, columns = data symbols, stored column
     by column in q) filled row by row, RI positions skipped and marked 0. ulsch_deinterleave (:891-918) scatters g[lut[i]] = q[i] in
     ascending i, so g[0] ends up holding the LLR of the LAST RI position; orc_ulsch_deinterleave below does the same. */
  const uint32_t rows = nof_re / nof_symb;
  uint8_t*       present = calloc(nof_re * Qm, 1);
  g_is_ri = 1;
  for (uint32_t i = 0; i < Qprime_ri; i++) {
    const uint32_t p = ack_symbol_qpos(i, Qm, rows, nof_symb);
    for (uint32_t k = 0; k < Qm; k++) present[p + k] = 1;
  }
  g_is_ri = 0;
  uint32_t idx = 0;
  for (uint32_t j = 0; j < rows; j++) {
    for (uint32_t i = 0; i < nof_symb; i++) {
      for (uint32_t k = 0; k < Qm; k++) {
        const uint32_t pos = j * Qm + i * rows * Qm + k;
        lut[pos] = present[pos] ? 0 : idx++;
      }
    }
  }
  free(present);
  return (int)idx;
}

void orc_ulsch_deinterleave(const int16_t* q_llr, const uint32_t* lut, int16_t* g_llr, uint32_t n)
{ /* srslte_vec_lut_sis (vector.c:112-116) as ulsch_deinterleave uses it */
  for (uint32_t i = 0; i < n; i++) g_llr[lut[i]] = q_llr[i];
}
