/*
 * oracle/orc_mimo.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 * Two-layer transmission modes of the PDSCH on a 2-port cell received with 2 antennas:
 *   large-delay CDD (TM3)            srslte_precoding_cdd_2x2_gen           precoding.c:1946-1957
 *                                    srslte_predecoding_ccd_2x2_mmse_csi    precoding.c:918-1014
 *   closed-loop multiplexing (TM4)   srslte_precoding_multiplex             precoding.c:1984-2104
 *                                    srslte_predecoding_multiplex_2x2_mmse_csi  :1326-1438, _2x1_mrc_csi :1624-1707
 * with the 2x2 MMSE solver srslte_mat_2x2_mmse_csi_gen (utils/mat.c). srslte_pdsch_decode always reaches the *_csi variants (the UE
 * object carries csi buffers, pdsch.c:914-926) and the MMSE ones (the file-static mimo_decoder defaults to MMSE, precoding.c:46;
 * cfg->decoder_type ZF only zeroes the noise term, pdsch.c:866). nof_layers == nof_tb in every case the reference accepts
 * (ra_dl.c:556-600), so there is no layer de-mapping: layer l is codeword l.
 * The reference's x86 build runs the SIMD bodies of these functions, whose reciprocals are the 12-bit _mm256_rcp_ps approximations
 * (simd.h:284-300,:959-978): its equalised symbols differ from the exact arithmetic here by up to ~4e-4 relative; the pins in
 * tests/test_oracle_vs_ref.py state that tolerance.
 * Channel layout: h[port * 2 + antenna], y[antenna].
 */
#include "orc.h"
#include <complex.h>
#include <math.h>

typedef float _Complex cfl;
static inline cfl C_(orc_cf_t v) { return v.re + v.im * I; }
static inline orc_cf_t O_(cfl v) { return (orc_cf_t){crealf(v), cimagf(v)}; }

void orc_precoding_cdd2(const orc_cf_t* x0, const orc_cf_t* x1, orc_cf_t* y0, orc_cf_t* y1, int nof_symbols, float scaling)
{ /* srslte_precoding_cdd_2x2_gen: W U D(i): the second port's sign of layer 0 / layer 1 alternates with the symbol index */
  const float s = scaling / 2.0f;
  for (int i = 0; i < nof_symbols; i++) {
    const cfl a = C_(x0[i]), b = C_(x1[i]);
    y0[i] = O_((a + b) * s);
    y1[i] = (i & 1) ? O_((-a + b) * s) : O_((a - b) * s);
  }
}

int orc_precoding_mux2(const orc_cf_t* x0, const orc_cf_t* x1, orc_cf_t* y0, orc_cf_t* y1, int nof_layers, int codebook_idx, int nof_symbols,
                       float scaling)
{ /* srslte_precoding_multiplex for 2 ports: 36.211 Table 6.3.4.2.3-1 */
  if (nof_layers == 1) {
    if (codebook_idx < 0 || codebook_idx > 3) return -1;
    const float s = scaling / sqrtf(2.0f);
    for (int i = 0; i < nof_symbols; i++) {
      const cfl a = C_(x0[i]);
      y0[i]       = O_(a * s);
      y1[i]       = codebook_idx == 0 ? O_(a * s) : (codebook_idx == 1 ? O_(a * -s) : (codebook_idx == 2 ? O_(a * (I * s)) : O_(a * (-I * s))));
    }
    return 0;
  }
  if (nof_layers != 2 || codebook_idx < 0 || codebook_idx > 2) return -1;
  const float s = codebook_idx == 0 ? scaling / sqrtf(2.0f) : scaling / 2.0f;
  for (int i = 0; i < nof_symbols; i++) {
    const cfl a = C_(x0[i]), b = C_(x1[i]);
    if (codebook_idx == 0) {
      y0[i] = O_(a * s);
      y1[i] = O_(b * s);
    } else if (codebook_idx == 1) {
      y0[i] = O_((a + b) * s);
      y1[i] = O_((a - b) * s);
    } else {
      y0[i] = O_((a + b) * s);
      y1[i] = O_((I * a - I * b) * s);
    }
  }
  return 0;
}

static void mmse_2x2(cfl y0, cfl y1, cfl h00, cfl h01, cfl h10, cfl h11, orc_cf_t* x0, orc_cf_t* x1, float* csi0, float* csi1, float noise,
                     float norm)
{ /* srslte_mat_2x2_mmse_csi_gen: A = H'H + N0 I, B = norm A^-1, W = B H', x = W y, csi_l = 1 / Re(B_ll) */
  const cfl _h00 = conjf(h00), _h01 = conjf(h01), _h10 = conjf(h10), _h11 = conjf(h11);
  const cfl a00 = _h00 * h00 + _h10 * h10 + noise, a01 = _h00 * h01 + _h10 * h11, a10 = _h01 * h00 + _h11 * h10, a11 = _h01 * h01 + _h11 * h11 + noise;
  const cfl det = a00 * a11 - a01 * a10;
  const cfl rcp = conjf(det) / (crealf(det) * crealf(det) + cimagf(det) * cimagf(det)); /* srslte_mat_cf_recip_gen */
  const cfl n_  = norm * rcp;
  const cfl b00 = a11 * n_, b01 = -a01 * n_, b10 = -a10 * n_, b11 = a00 * n_;
  const cfl w00 = b00 * _h00 + b01 * _h01, w01 = b00 * _h10 + b01 * _h11, w10 = b10 * _h00 + b11 * _h01, w11 = b10 * _h10 + b11 * _h11;
  *x0   = O_(y0 * w00 + y1 * w01);
  *x1   = O_(y0 * w10 + y1 * w11);
  *csi0 = 1.0f / crealf(b00);
  *csi1 = 1.0f / crealf(b11);
}

void orc_predecoding_cdd_2x2(const orc_cf_t* const* y, const orc_cf_t* const* h, orc_cf_t* x0, orc_cf_t* x1, float* csi0, float* csi1, int nof_symbols,
                             float scaling, float noise_estimate)
{ /* srslte_predecoding_ccd_2x2_mmse_csi: the effective channel of symbol i is H W U D(i) */
  const float norm = 2.0f / scaling;
  for (int i = 0; i < nof_symbols; i++) {
    const cfl p0a0 = C_(h[0][i]), p0a1 = C_(h[1][i]), p1a0 = C_(h[2][i]), p1a1 = C_(h[3][i]);
    cfl       h00, h01, h10, h11;
    if (!(i & 1)) {
      h00 = p0a0 + p1a0; h10 = p0a1 + p1a1; h01 = p0a0 - p1a0; h11 = p0a1 - p1a1;
    } else {
      h00 = p0a0 - p1a0; h10 = p0a1 - p1a1; h01 = p0a0 + p1a0; h11 = p0a1 + p1a1;
    }
    mmse_2x2(C_(y[0][i]), C_(y[1][i]), h00, h01, h10, h11, &x0[i], &x1[i], &csi0[i], &csi1[i], noise_estimate, norm);
  }
}

int orc_predecoding_mux_2x2(const orc_cf_t* const* y, const orc_cf_t* const* h, orc_cf_t* x0, orc_cf_t* x1, float* csi0, float* csi1, int codebook_idx,
                            int nof_symbols, float scaling, float noise_estimate)
{ /* srslte_predecoding_multiplex_2x2_mmse_csi */
  if (codebook_idx < 0 || codebook_idx > 2) return -1;
  const float norm = codebook_idx == 0 ? (float)M_SQRT2 / scaling : 2.0f / scaling;
  for (int i = 0; i < nof_symbols; i++) {
    const cfl p0a0 = C_(h[0][i]), p0a1 = C_(h[1][i]), p1a0 = C_(h[2][i]), p1a1 = C_(h[3][i]);
    cfl       h00, h01, h10, h11;
    if (codebook_idx == 0) {
      h00 = p0a0; h01 = p1a0; h10 = p0a1; h11 = p1a1;
    } else if (codebook_idx == 1) {
      h00 = p0a0 + p1a0; h01 = p0a0 - p1a0; h10 = p0a1 + p1a1; h11 = p0a1 - p1a1;
    } else {
      h00 = p0a0 + I * p1a0; h01 = p0a0 - I * p1a0; h10 = p0a1 + I * p1a1; h11 = p0a1 - I * p1a1;
    }
    mmse_2x2(C_(y[0][i]), C_(y[1][i]), h00, h01, h10, h11, &x0[i], &x1[i], &csi0[i], &csi1[i], noise_estimate, norm);
  }
  return 0;
}

int orc_predecoding_mux_2x1(const orc_cf_t* const* y, const orc_cf_t* const* h, orc_cf_t* x, float* csi, int codebook_idx, int nof_symbols, float scaling)
{ /* srslte_predecoding_multiplex_2x1_mrc_csi: one layer over both ports, maximum-ratio combining of the two antennas */
  if (codebook_idx < 0 || codebook_idx > 3) return -1;
  const float norm = (float)M_SQRT2 / scaling;
  for (int i = 0; i < nof_symbols; i++) {
    const cfl p0a0 = C_(h[0][i]), p0a1 = C_(h[1][i]), p1a0 = C_(h[2][i]), p1a1 = C_(h[3][i]);
    cfl       h0, h1;
    switch (codebook_idx) {
      case 0: h0 = p0a0 + p1a0; h1 = p0a1 + p1a1; break;
      case 1: h0 = p0a0 - p1a0; h1 = p0a1 - p1a1; break;
      case 2: h0 = p0a0 + I * p1a0; h1 = p0a1 + I * p1a1; break;
      default: h0 = p0a0 - I * p1a0; h1 = p0a1 - I * p1a1; break;
    }
    const float c_ = crealf(h0) * crealf(h0) + cimagf(h0) * cimagf(h0) + crealf(h1) * crealf(h1) + cimagf(h1) * cimagf(h1);
    const float hh = norm / c_;
    x[i]           = O_((conjf(h0) * C_(y[0][i]) + conjf(h1) * C_(y[1][i])) * hh);
    csi[i]         = c_ / norm * (float)M_SQRT1_2;
  }
  return 0;
}
