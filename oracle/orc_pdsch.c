/*
 * oracle/orc_pdsch.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 * CPU restatement of the PDSCH glue between the named hot-path stages:
 * RE (de)mapping (pdsch.c:81-206 + prb_dl.c:45-91), single-port one-tap equaliser
 * (precoding.c:238-322), LLR descrambling (scrambling.c:45-48, sequences.c:58-60).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_pdsch_indices(const orc_cell_t* cell, uint32_t sf_idx, uint32_t lstart, const uint8_t* prb_mask, uint32_t* idx)
{ /* pdsch.c:81-206 expressed per RE: symbol-major, PRB ascending, skipping CRS and (FDD) the central
     72 sub-carriers of PSS/SSS symbols (slot 0, last two symbols, sf 0/5) and PBCH symbols (slot 1, l<4, sf 0).
     prb_mask: NULL = every PRB in both slots, else [2][nof_prb] bytes = srslte_pdsch_grant_t.prb_idx[s][n] (pdsch.c:119).
     For odd nof_prb the sync region cuts PRBs nof_prb/2-3 and nof_prb/2+3 in half (pdsch.c:167-188): the same per-RE rule, with one
     upstream quirk kept: in those half PRBs the CRS position comes from the variable `offset`, which is only assigned in the branch
     of whole PRBs (pdsch.c:147-157) - so it holds what the last CRS-bearing whole PRB before it in the loop left there (0 if none). */
  uint32_t P = cell->nof_prb, nre = 12 * P, nsymb = cell->cp_norm ? 7 : 6, n = 0;
  uint32_t nof_refs = cell->nof_ports == 1 ? 2 : 4;
  uint32_t offset_var = 0; /* the reference's `offset`, pdsch.c:91 */
  /* grant->nof_symb_slot[s] (ra_dl.c:446-460): the cyclic prefix's count, or the DwPTS share of a TDD special subframe */
  const uint32_t nss[2] = {orc_nof_symb_slot(cell, sf_idx, 0), orc_nof_symb_slot(cell, sf_idx, 1)};
  for (uint32_t s = 0; s < 2; s++) {
    for (uint32_t l = (s == 0 ? lstart : 0); l < nss[s]; l++) {
      bool has_ref = (l == 1 && cell->nof_ports == 4) || l == 0 || l == nsymb - 3; /* phy_common.h:139-141 */
      uint32_t offset = nof_refs == 2 ? (l == 0 ? cell->id % 6 : (cell->id + 3) % 6) : cell->id % 3;
      /* pdsch.c:124-140: FDD PSS / SSS on the last two symbols of slot 0; TDD SSS on the last symbol of slot 1 (subframes 0 / 5) and PSS
         on symbol 2 of slot 0 (subframes 1 / 6); PBCH the same in both */
      bool sync = cell->frame_type ? ((s == 1 && (sf_idx == 0 || sf_idx == 5) && l + 1 >= nss[1]) || (s == 0 && (sf_idx == 1 || sf_idx == 6) && l == 2))
                                   : (s == 0 && (sf_idx == 0 || sf_idx == 5) && l + 2 >= nss[0]);
      sync      = sync || (s == 1 && sf_idx == 0 && l < 4);
      for (uint32_t p = 0; p < P; p++) {
        if (prb_mask && !prb_mask[s * P + p]) continue;
        bool centre = p >= P / 2 - 3 && p < P / 2 + 3 + (P % 2);
        bool skip   = centre && sync;
        uint32_t off_used = offset;
        if (!skip) {
          if (has_ref) offset_var = offset;
        } else {
          off_used = offset_var;
        }
        for (uint32_t k = 12 * p; k < 12 * p + 12; k++) {
          if (sync && k + 36 >= nre / 2 && k < nre / 2 + 36) continue;
          if (has_ref && (k % (12 / nof_refs)) == off_used % (12 / nof_refs)) continue;
          idx[n++] = (s * nss[0] + l) * nre + k; /* lp = l + s * nof_symb_slot[0], pdsch.c:142 */
        }
      }
    }
  }
  return (int)n;
}

int orc_pdsch_cp(const orc_cell_t* cell, uint32_t sf_idx, uint32_t lstart, const uint8_t* prb_mask, orc_cf_t* grid, orc_cf_t* syms, bool put)
{
  uint32_t* idx = malloc(sizeof(uint32_t) * 14 * 12 * cell->nof_prb);
  int       n   = orc_pdsch_indices(cell, sf_idx, lstart, prb_mask, idx);
  for (int i = 0; i < n; i++) {
    if (put) {
      grid[idx[i]] = syms[i];
    } else {
      syms[i] = grid[idx[i]];
    }
  }
  free(idx);
  return n;
}

void orc_predecoding_single(const orc_cf_t* y, const orc_cf_t* h, orc_cf_t* x, int nsym, float scaling, float noise_estimate)
{ /* precoding.c:277-288 (scalar form of the csi variant srslte_pdsch_decode reaches: x = y h* (1/scaling) / (|h|^2 + N0)).
     NOTE: the reference's AVX body replaces the division by _mm256_rcp_ps (12-bit approximation whose
     result differs between CPU vendors), so parity with it is checked at 1e-3, not 1e-4. */
  float norm = 1.0f / scaling;
  for (int i = 0; i < nsym; i++) {
    float re = y[i].re * h[i].re + y[i].im * h[i].im, im = y[i].im * h[i].re - y[i].re * h[i].im;
    float csi = h[i].re * h[i].re + h[i].im * h[i].im + noise_estimate;
    x[i].re = re * norm / csi;
    x[i].im = im * norm / csi;
  }
}

void orc_predecoding_single_multi(const orc_cf_t* const* y, const orc_cf_t* const* h, orc_cf_t* x, int nof_rx, int nsym, float scaling,
                                  float noise_estimate)
{ /* srslte_predecoding_single_multi (precoding.c:138-262): maximum-ratio combining, exact division in every code path */
  for (int i = 0; i < nsym; i++) {
    float re = 0, im = 0, hh = 0;
    for (int p = 0; p < nof_rx; p++) {
      const orc_cf_t Y = y[p][i], H = h[p][i];
      float pr = Y.re * H.re + Y.im * H.im, pi = Y.im * H.re - Y.re * H.im, ph = H.re * H.re + H.im * H.im;
      re = p ? re + pr : pr;
      im = p ? im + pi : pi;
      hh = p ? hh + ph : ph;
    }
    if (noise_estimate > 0) hh += noise_estimate;
    x[i].re = re / hh * (1 / scaling);
    x[i].im = im / hh * (1 / scaling);
  }
}

void orc_scramble_s(int16_t* llr, const uint8_t* c, int len)
{ /* scrambling.c:45-48 -> srslte_vec_neg_sss: negate where c_short = 1-2c is negative */
  for (int i = 0; i < len; i++) {
    if (c[i]) llr[i] = (int16_t)-llr[i];
  }
}

void orc_scramble_b(int8_t* llr, const uint8_t* c, int len)
{ /* scrambling.c:48-51 -> srslte_vec_neg_bbb (_mm256_sign_epi8 with c_char = 1-2c, never 0): -(-128) stays -128 */
  for (int i = 0; i < len; i++) {
    if (c[i]) llr[i] = (int8_t)-llr[i];
  }
}

uint32_t orc_pdsch_cinit(uint16_t rnti, uint32_t cw, uint32_t sf_idx, uint32_t cell_id)
{ /* sequences.c:58-60 with nslot = 2*sf_idx (pdsch.c:469) */
  return ((uint32_t)rnti << 14) + (cw << 13) + (((2 * sf_idx) / 2) << 9) + cell_id;
}

/* ------------------------------------------------------------------ PMCH in an MBSFN subframe */

int orc_pmch_indices(uint32_t nof_prb, uint32_t lstart, uint32_t* idx)
{ /* pmch_cp (pmch.c:44-99): every PRB of the 12-symbol (extended-CP) subframe from symbol lstart on; in the symbols that carry the MBSFN
     reference signal (SRSLTE_SYMBOL_HAS_REF_MBSFN, phy_common.h:145: l = 2 of slot 0, l = 0 and 4 of slot 1) prb_cp_ref with 6 references per
     PRB (prb_dl.c:45-76) leaves every second RE: the odd sub-carriers, the even ones in symbol 0 of slot 1 (offset 1) */
  int n = 0;
  for (uint32_t s = 0; s < 2; s++) {
    for (uint32_t l = 0; l < 6; l++) {
      if (s == 0 && l < lstart) continue;
      const int has_ref = (l == 2 && s == 0) || (l == 0 && s == 1) || (l == 4 && s == 1);
      const uint32_t keep = (l == 0 && s == 1) ? 0 : 1; /* parity of the sub-carriers that carry data */
      for (uint32_t k = 0; k < 12 * nof_prb; k++) {
        if (has_ref && (k & 1) != keep) continue;
        idx[n++] = (l + 6 * s) * 12 * nof_prb + k;
      }
    }
  }
  return n;
}

uint32_t orc_pmch_cinit(uint32_t sf_idx, uint32_t area_id)
{ /* srslte_sequence_pmch (sequences.c:76-80) called with nslot = 2 * sf_idx (pmch.c:263-275) */
  return (((2 * sf_idx) / 2) << 9) + area_id;
}

/* ------------------------------------------------------------------ 2-port transmit diversity (SFBC), SURVEY §8f N4 */

void orc_precoding_diversity2(const orc_cf_t* d, orc_cf_t* y0, orc_cf_t* y1, int nof_symbols, float scaling)
{ /* srslte_layermap_diversity (layermap.c:36-44) + srslte_precoding_diversity for 2 ports (precoding.c:1848-1861): d[nof_symbols]
     -> layers x0 = d[2i], x1 = d[2i+1] -> y0[2i] = x0, y1[2i] = -x1*, y0[2i+1] = x1, y1[2i+1] = x0*, all times scaling / sqrt(2) */
  const float g = scaling / sqrtf(2);
  for (int i = 0; i < nof_symbols / 2; i++) {
    const orc_cf_t x0 = d[2 * i], x1 = d[2 * i + 1];
    y0[2 * i]     = (orc_cf_t){x0.re * g, x0.im * g};
    y1[2 * i]     = (orc_cf_t){-x1.re * g, x1.im * g};
    y0[2 * i + 1] = (orc_cf_t){x1.re * g, x1.im * g};
    y1[2 * i + 1] = (orc_cf_t){x0.re * g, -x0.im * g};
  }
}

void orc_predecoding_diversity2(const orc_cf_t* const* y, const orc_cf_t* const* h /* [port * nof_rx + antenna] */, orc_cf_t* d, float* csi,
                                int nof_rx, int nof_symbols, float scaling)
{ /* srslte_predecoding_diversity_csi for 2 ports (precoding.c:564-598; the variant srslte_pdsch_decode reaches because the UE object
     always carries a csi buffer) followed by srslte_layerdemap_diversity (layermap.c:140-148): d[2i] = x0[i], d[2i+1] = x1[i].
     The upstream accumulates hh across antennas BEFORE testing it for zero, inside the antenna loop; reproduced. */
  for (int i = 0; i < nof_symbols / 2; i++) {
    float hh = 0, x0r = 0, x0i = 0, x1r = 0, x1i = 0;
    for (int p = 0; p < nof_rx; p++) {
      const orc_cf_t h00 = h[p][2 * i], h01 = h[p][2 * i + 1], h10 = h[nof_rx + p][2 * i], h11 = h[nof_rx + p][2 * i + 1];
      const orc_cf_t r0 = y[p][2 * i], r1 = y[p][2 * i + 1];
      hh += h00.re * h00.re + h00.im * h00.im + h11.re * h11.re + h11.im * h11.im;
      if (hh == 0) hh = 1e-4f;
      /* x0 += conj(h00) r0 + h11 conj(r1) */
      x0r += h00.re * r0.re + h00.im * r0.im + h11.re * r1.re + h11.im * r1.im;
      x0i += h00.re * r0.im - h00.im * r0.re + h11.im * r1.re - h11.re * r1.im;
      /* x1 += -h10 conj(r0) + conj(h01) r1 */
      x1r += -(h10.re * r0.re + h10.im * r0.im) + h01.re * r1.re + h01.im * r1.im;
      x1i += -(h10.im * r0.re - h10.re * r0.im) + h01.re * r1.im - h01.im * r1.re;
    }
    if (csi) {
      csi[2 * i]     = hh;
      csi[2 * i + 1] = hh;
    }
    hh *= scaling;
    d[2 * i]     = (orc_cf_t){(float)((double)(x0r / hh) * sqrt(2)), (float)((double)(x0i / hh) * sqrt(2))};
    d[2 * i + 1] = (orc_cf_t){(float)((double)(x1r / hh) * sqrt(2)), (float)((double)(x1i / hh) * sqrt(2))};
  }
}

/* ------------------------------------------------------------------ CSI weighting of the LLRs (pdsch.c:574-690, cfg->csi_enable) */

void orc_predecoding_csi(const orc_cf_t* const* h, float* csi, int nof_rx, int nsym, float noise_estimate)
{ /* the csi side output of srslte_predecoding_single_csi (precoding.c:251-291): sum over antennas of |h|^2, plus the noise estimate */
  for (int i = 0; i < nsym; i++) {
    float hh = 0;
    for (int p = 0; p < nof_rx; p++) hh += h[p][i].re * h[p][i].re + h[p][i].im * h[p][i].im;
    csi[i] = hh + noise_estimate;
  }
}

static int16_t csi_weight(float csi, float scale)
{ /* _mm_cvtps_pi16: round to nearest even, then saturate to int16 */
  float v = nearbyintf(csi * scale);
  return (int16_t)(v > 32767.0f ? 32767 : (v < -32768.0f ? -32768 : (int)v));
}

void orc_csi_correction_s(int16_t* e, const float* csi, int nsym, int mod)
{ /* pdsch.c:599-603,:615-689, 16-bit LLRs on an SSE host: weights w = round(csi * 32767 / csi_max), e <- (e * w) >> 16 for whole groups of
     4 (QPSK: two symbols; 16QAM: one), 12 (64QAM: two symbols) or 8 (256QAM: one) LLRs; the symbols a group does not cover get
     e <- (int16)(e * csi / csi_max) instead - without the factor 1/2 the high-half multiplication implies; reproduced.
     In the two-symbol groups _mm_blend_ps(a, b, 3) takes its LOW two lanes from b: QPSK weighs symbol 2j with csi[2j+1] and vice versa,
     64QAM weighs LLRs 4,5 of a group (symbol 2j) with csi[2j+1] and LLRs 6,7 (symbol 2j+1) with csi[2j]; reproduced. */
  const int qm = orc_mod_bits(mod), nbits = nsym * qm;
  float     csi_max = 1.0f;
  if (nsym > 0) {
    int imax = 0;
    for (int i = 1; i < nsym; i++) {
      if (csi[i] > csi[imax]) imax = i;
    }
    csi_max = csi[imax];
  }
  const float scale = 32767.0f / csi_max;
  int         i = 0, s = 0; /* LLR index, symbol index */
#define MULHI(idx, w) e[idx] = (int16_t)(((int)e[idx] * (int)(w)) >> 16)
  switch (mod) {
    case ORC_MOD_QPSK:
      for (; i < nbits - 3; i += 4, s += 2) {
        const int16_t w1 = csi_weight(csi[s], scale), w2 = csi_weight(csi[s + 1], scale);
        MULHI(i, w2); MULHI(i + 1, w2); MULHI(i + 2, w1); MULHI(i + 3, w1);
      }
      break;
    case ORC_MOD_16QAM:
      for (; i < nbits - 3; i += 4, s++) {
        const int16_t w = csi_weight(csi[s], scale);
        for (int k = 0; k < 4; k++) MULHI(i + k, w);
      }
      break;
    case ORC_MOD_64QAM:
      for (; i < nbits - 11; i += 12, s += 2) {
        const int16_t w1 = csi_weight(csi[s], scale), w3 = csi_weight(csi[s + 1], scale);
        for (int k = 0; k < 4; k++) MULHI(i + k, w1);
        MULHI(i + 4, w3); MULHI(i + 5, w3); MULHI(i + 6, w1); MULHI(i + 7, w1);
        for (int k = 8; k < 12; k++) MULHI(i + k, w3);
      }
      break;
    case ORC_MOD_256QAM:
      for (; i < nbits - 7; i += 8, s++) {
        const int16_t w = csi_weight(csi[s], scale);
        for (int k = 0; k < 8; k++) MULHI(i + k, w);
      }
      break;
    default: break;
  }
#undef MULHI
  const float inv = 1.0f / csi_max; /* the -Ofast build multiplies by the hoisted reciprocal */
  for (i /= qm; i < nsym; i++) {
    const float c = csi[i] * inv;
    for (int k = 0; k < qm; k++) e[qm * i + k] = (int16_t)((float)e[qm * i + k] * c);
  }
}

void orc_csi_correction_b(int8_t* e, const float* csi, int nsym, int mod)
{ /* pdsch.c:607-614, 8-bit LLRs: e <- (int8)(e * (csi / csi_max)), truncating */
  const int qm = orc_mod_bits(mod);
  float     csi_max = 1.0f;
  if (nsym > 0) {
    int imax = 0;
    for (int i = 1; i < nsym; i++) {
      if (csi[i] > csi[imax]) imax = i;
    }
    csi_max = csi[imax];
  }
  const float inv = 1.0f / csi_max; /* the -Ofast build multiplies by the hoisted reciprocal */
  for (int i = 0; i < nsym; i++) {
    const float c = csi[i] * inv;
    for (int k = 0; k < qm; k++) e[qm * i + k] = (int8_t)((float)e[qm * i + k] * c);
  }
}

/* ------------------------------------------------------------------ 4-port transmit diversity (SFBC + FSTD), SURVEY §8f N4 */

void orc_precoding_diversity4(const orc_cf_t* d, orc_cf_t* const* y /* [4] */, int nof_symbols, float scaling)
{ /* srslte_layermap_diversity with 4 layers (layermap.c:36-44) + srslte_precoding_diversity for 4 ports (precoding.c:1862-1890):
     ports 0/2 carry the Alamouti pair of sub-carriers 4i, 4i+1, ports 1/3 that of 4i+2, 4i+3, the other two ports stay silent */
  const float    g = scaling / sqrtf(2);
  const orc_cf_t z = {0, 0};
  for (int i = 0; i < nof_symbols / 4; i++) {
    const orc_cf_t x0 = d[4 * i], x1 = d[4 * i + 1], x2 = d[4 * i + 2], x3 = d[4 * i + 3];
    y[0][4 * i] = (orc_cf_t){x0.re * g, x0.im * g};      y[1][4 * i] = z; y[2][4 * i] = (orc_cf_t){-x1.re * g, x1.im * g};     y[3][4 * i] = z;
    y[0][4 * i + 1] = (orc_cf_t){x1.re * g, x1.im * g};  y[1][4 * i + 1] = z; y[2][4 * i + 1] = (orc_cf_t){x0.re * g, -x0.im * g}; y[3][4 * i + 1] = z;
    y[0][4 * i + 2] = z; y[1][4 * i + 2] = (orc_cf_t){x2.re * g, x2.im * g};  y[2][4 * i + 2] = z; y[3][4 * i + 2] = (orc_cf_t){-x3.re * g, x3.im * g};
    y[0][4 * i + 3] = z; y[1][4 * i + 3] = (orc_cf_t){x3.re * g, x3.im * g};  y[2][4 * i + 3] = z; y[3][4 * i + 3] = (orc_cf_t){x2.re * g, -x2.im * g};
  }
}

void orc_predecoding_diversity4(const orc_cf_t* const* y, const orc_cf_t* const* h /* [port * nof_rx + antenna] */, orc_cf_t* d, float* csi,
                                int nof_rx, int nof_symbols, float scaling)
{ /* srslte_predecoding_diversity_csi for 4 ports (precoding.c:599-650) + srslte_layerdemap_diversity with 4 layers: every symbol of a
     group of four has its own divisor (|h|^2 of the two ports at its own and its pair partner's sub-carrier); csi = divisor / nof_rx */
  const int m_ap = (nof_symbols % 4) ? (nof_symbols - 2) / 4 : nof_symbols / 4;
  for (int i = 0; i < m_ap; i++) {
    float a[4] = {0, 0, 0, 0}, xr[4] = {0, 0, 0, 0}, xi[4] = {0, 0, 0, 0};
    for (int p = 0; p < nof_rx; p++) {
      for (int half = 0; half < 2; half++) { /* sub-carriers 4i, 4i+1 with ports 0 and 2; 4i+2, 4i+3 with ports 1 and 3 */
        const int      k = 4 * i + 2 * half;
        const orc_cf_t h00 = h[(0 + half) * nof_rx + p][k], h01 = h[(2 + half) * nof_rx + p][k];
        const orc_cf_t h10 = h[(0 + half) * nof_rx + p][k + 1], h11 = h[(2 + half) * nof_rx + p][k + 1];
        const orc_cf_t r0 = y[p][k], r1 = y[p][k + 1];
        a[2 * half] += h00.re * h00.re + h00.im * h00.im + h11.re * h11.re + h11.im * h11.im;
        a[2 * half + 1] += h10.re * h10.re + h10.im * h10.im + h01.re * h01.re + h01.im * h01.im;
        /* x0 += conj(h00) r0 + h11 conj(r1) */
        xr[2 * half] += h00.re * r0.re + h00.im * r0.im + h11.re * r1.re + h11.im * r1.im;
        xi[2 * half] += h00.re * r0.im - h00.im * r0.re + h11.im * r1.re - h11.re * r1.im;
        /* x1 += -h01 conj(r0) + conj(h10) r1 */
        xr[2 * half + 1] += -(h01.re * r0.re + h01.im * r0.im) + h10.re * r1.re + h10.im * r1.im;
        xi[2 * half + 1] += -(h01.im * r0.re - h01.re * r0.im) + h10.re * r1.im - h10.im * r1.re;
      }
    }
    for (int j = 0; j < 4; j++) {
      a[j] *= scaling;
      if (csi) csi[4 * i + j] = a[j] / nof_rx;
      d[4 * i + j] = (orc_cf_t){xr[j] / a[j] * sqrtf(2.0f), xi[j] / a[j] * sqrtf(2.0f)};
    }
  }
}
