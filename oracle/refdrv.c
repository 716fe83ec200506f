/*
 * oracle/refdrv.c — TEST INFRASTRUCTURE ONLY (never linked or called by the product).
 *
 * A small C driver, written for this repo, that calls the REFERENCE's own compiled functions (oracle/_ref/libsrslte_ref.so, built by
 * ref.mk from the sources where they lie) the way the reference's own callers do, but starting from a frequency-domain subframe:
 * the reference's OFDM (lib/src/phy/dft/ofdm.c) needs FFTW, which this image lacks, so whoever calls this driver supplies the grid
 * (the oracle's orc_ofdm_rx_sf, or the device's). Everything after the FFT is the reference:
 *
 *   refdrv_dl_estimate           = estimate_pdcch_pcfich          lib/src/phy/ue/ue_dl.c:334-367
 *   refdrv_dl_find_dci           = srslte_ue_dl_find_dl_dci       ue_dl.c:620-646 (SI/P/RA-RNTI: :534-564, C-RNTI: :566-618) + dci_blind_search :422-478
 *   refdrv_dl_decode_pdsch       = the rest of srslte_ue_dl_find_and_decode  ue_dl.c:1295-1358
 *   refdrv_dl_pmch_decode        = lib/src/phy/phch/test/pmch_file_test.c:189-216
 *   refdrv_dl_rx_loop            = the receive loop of lib/test/phy/phy_dl_test.c:197-259 for a known grant (CPU baseline of bench.py)
 *
 * With it the reference's recorded-IQ CTests (lib/src/phy/phch/test/CMakeLists.txt:233-238: pdsch_pdcch_file_test, pcfich_file_test,
 * pmch_file_test, pbch_file_test) are re-run on a grid that did not come from FFTW: they reach the result those tests assert only
 * if the OFDM demodulator's CP offsets, bin order, DC skip and MBSFN slot layout are the reference's.
 *
 * Built by oracle/ref.mk into oracle/_ref/librefdrv.so (with the reference's headers; not built where /root/reference is absent).
 */
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/time.h>

#include "srslte/phy/ch_estimation/chest_dl.h"
#include "srslte/phy/phch/dci.h"
#include "srslte/phy/phch/pbch.h"
#include "srslte/phy/phch/pcfich.h"
#include "srslte/phy/phch/pdcch.h"
#include "srslte/phy/phch/pdsch.h"
#include "srslte/phy/phch/pmch.h"
#include "srslte/phy/phch/ra_dl.h"
#include "srslte/phy/phch/regs.h"
#include "srslte/phy/utils/vector.h"

#define REFDRV_MAX_CAND 64

typedef struct {
  srslte_cell_t          cell;
  uint32_t               nof_rx;
  cf_t*                  sf_symbols[SRSLTE_MAX_PORTS];
  srslte_chest_dl_t      chest;
  srslte_chest_dl_res_t  chest_res;
  srslte_regs_t          regs;
  srslte_pcfich_t        pcfich;
  srslte_pdcch_t         pdcch;
  srslte_pdsch_t         pdsch;
  srslte_pmch_t          pmch;
  srslte_pbch_t          pbch;
  bool                   have_pbch;
  srslte_softbuffer_rx_t softbuffer;
  srslte_dl_sf_cfg_t     sf;
  srslte_chest_dl_cfg_t  chest_cfg;
  srslte_dci_cfg_t       dci_cfg;
  srslte_dci_dl_t        dci_dl;
  srslte_pdsch_cfg_t     pdsch_cfg;
  uint8_t*               payload;
} refdrv_dl_t;

void refdrv_dl_free(refdrv_dl_t* q)
{
  if (!q) return;
  srslte_chest_dl_free(&q->chest);
  srslte_chest_dl_res_free(&q->chest_res);
  srslte_regs_free(&q->regs);
  srslte_pcfich_free(&q->pcfich);
  srslte_pdcch_free(&q->pdcch);
  srslte_pdsch_free(&q->pdsch);
  srslte_pmch_free(&q->pmch);
  if (q->have_pbch) srslte_pbch_free(&q->pbch);
  srslte_softbuffer_rx_free(&q->softbuffer);
  for (int i = 0; i < SRSLTE_MAX_PORTS; i++) free(q->sf_symbols[i]);
  free(q->payload);
  free(q);
}

/* srslte_ue_dl_init + srslte_ue_dl_set_cell (ue_dl.c:64-246) without the OFDM objects and PHICH; FDD, mi = 1 (regs[0]). */
refdrv_dl_t* refdrv_dl_new(uint32_t nof_prb, uint32_t nof_ports, uint32_t cell_id, int cp_ext, uint32_t nof_rx, int phich_resources, int phich_ext)
{
  refdrv_dl_t* q = calloc(1, sizeof(*q));
  if (!q) return NULL;
  q->cell.nof_prb         = nof_prb;
  q->cell.nof_ports       = nof_ports;
  q->cell.id              = cell_id;
  q->cell.cp              = cp_ext ? SRSLTE_CP_EXT : SRSLTE_CP_NORM;
  q->cell.phich_length    = phich_ext ? SRSLTE_PHICH_EXT : SRSLTE_PHICH_NORM;
  q->cell.phich_resources = (srslte_phich_r_t)phich_resources;
  q->cell.frame_type      = SRSLTE_FDD;
  q->nof_rx               = nof_rx;
  for (int i = 0; i < SRSLTE_MAX_PORTS; i++) {
    q->sf_symbols[i] = srslte_vec_malloc(sizeof(cf_t) * SRSLTE_SF_LEN_RE(SRSLTE_MAX_PRB, SRSLTE_CP_NORM));
    bzero(q->sf_symbols[i], sizeof(cf_t) * SRSLTE_SF_LEN_RE(SRSLTE_MAX_PRB, SRSLTE_CP_NORM));
  }
  q->payload = calloc(1, 100000);
  if (srslte_chest_dl_init(&q->chest, nof_prb, nof_rx) || srslte_chest_dl_res_init(&q->chest_res, nof_prb) ||
      srslte_pcfich_init(&q->pcfich, nof_rx) || srslte_pdcch_init_ue(&q->pdcch, nof_prb, nof_rx) || srslte_pdsch_init_ue(&q->pdsch, nof_prb, nof_rx) ||
      srslte_pmch_init(&q->pmch, nof_prb, nof_rx) || srslte_softbuffer_rx_init(&q->softbuffer, nof_prb)) {
    refdrv_dl_free(q);
    return NULL;
  }
  if (srslte_regs_init(&q->regs, q->cell) || srslte_chest_dl_set_cell(&q->chest, q->cell) || srslte_pcfich_set_cell(&q->pcfich, &q->regs, q->cell) ||
      srslte_pdcch_set_cell(&q->pdcch, &q->regs, q->cell) || srslte_pdsch_set_cell(&q->pdsch, q->cell) || srslte_pmch_set_cell(&q->pmch, q->cell)) {
    refdrv_dl_free(q);
    return NULL;
  }
  q->pdsch_cfg.softbuffers.rx[0] = &q->softbuffer;
  return q;
}

cf_t* refdrv_dl_grid(refdrv_dl_t* q, uint32_t ant) { return q->sf_symbols[ant]; }
cf_t* refdrv_dl_ce(refdrv_dl_t* q, uint32_t port, uint32_t ant) { return q->chest_res.ce[port][ant]; }
uint8_t* refdrv_dl_payload(refdrv_dl_t* q) { return q->payload; }
srslte_chest_dl_res_t* refdrv_dl_chest_res(refdrv_dl_t* q) { return &q->chest_res; }

void refdrv_dl_set_rnti(refdrv_dl_t* q, uint16_t rnti)
{
  srslte_pdsch_set_rnti(&q->pdsch, rnti);
  q->pdsch_cfg.rnti = rnti;
}

int refdrv_dl_set_mbsfn_area_id(refdrv_dl_t* q, uint16_t area_id)
{ /* srslte_ue_dl_set_mbsfn_area_id, ue_dl.c:296-313 */
  if (srslte_chest_dl_set_mbsfn_area_id(&q->chest, area_id)) return -1;
  return srslte_pmch_set_area_id(&q->pmch, area_id);
}

void refdrv_dl_set_chest_cfg(refdrv_dl_t* q, int noise_alg, int filter_type, float coef0, float coef1, int interpolate_subframe, uint16_t mbsfn_area_id)
{
  bzero(&q->chest_cfg, sizeof(q->chest_cfg));
  q->chest_cfg.noise_alg            = (srslte_chest_dl_noise_alg_t)noise_alg;
  q->chest_cfg.filter_type          = (srslte_chest_filter_t)filter_type;
  q->chest_cfg.filter_coef[0]       = coef0;
  q->chest_cfg.filter_coef[1]       = coef1;
  q->chest_cfg.interpolate_subframe = interpolate_subframe;
  q->chest_cfg.mbsfn_area_id        = mbsfn_area_id;
}

void refdrv_dl_set_pdsch_cfg(refdrv_dl_t* q, uint32_t max_iterations, int mmse, int csi_enable, int llr_8bit)
{
  q->pdsch_cfg.max_nof_iterations = max_iterations;
  q->pdsch_cfg.decoder_type       = mmse ? SRSLTE_MIMO_DECODER_MMSE : SRSLTE_MIMO_DECODER_ZF;
  q->pdsch_cfg.csi_enable         = csi_enable;
  q->pdsch.llr_is_8bit            = llr_8bit;
  q->pdsch.dl_sch.llr_is_8bit     = llr_8bit;
}

/* estimate_pdcch_pcfich (ue_dl.c:334-367) on the grid in refdrv_dl_grid(): channel estimate, PCFICH, PDCCH LLRs.
 * sf_type 1 = MBSFN (the caller sets dl_sf.cfi itself afterwards, as pmch_file_test.c:187 does). */
int refdrv_dl_estimate(refdrv_dl_t* q, uint32_t tti, int sf_type, uint32_t cfi_in, uint32_t* cfi, float* cfi_corr)
{
  bzero(&q->sf, sizeof(q->sf));
  q->sf.tti     = tti;
  q->sf.cfi     = cfi_in;
  q->sf.sf_type = sf_type ? SRSLTE_SF_MBSFN : SRSLTE_SF_NORM;
  srslte_pdcch_set_regs(&q->pdcch, &q->regs);
  if (srslte_chest_dl_estimate_cfg(&q->chest, &q->sf, &q->chest_cfg, q->sf_symbols, &q->chest_res) < 0) return -1;
  float corr = 0;
  if (srslte_pcfich_decode(&q->pcfich, &q->sf, &q->chest_res, q->sf_symbols, &corr) < 0) return -1;
  if (srslte_pdcch_extract_llr(&q->pdcch, &q->sf, &q->chest_res, q->sf_symbols)) return -1;
  if (cfi) *cfi = q->sf.cfi;
  if (cfi_corr) *cfi_corr = corr;
  return 0;
}

/* the default-configuration entry pcfich_file_test.c:228-233 uses: srslte_chest_dl_estimate + srslte_pcfich_decode */
int refdrv_dl_pcfich(refdrv_dl_t* q, uint32_t tti, uint32_t* cfi, float* cfi_corr)
{
  bzero(&q->sf, sizeof(q->sf));
  q->sf.tti = tti;
  if (srslte_chest_dl_estimate(&q->chest, &q->sf, q->sf_symbols, &q->chest_res) < 0) return -1;
  float corr = 0;
  int   n    = srslte_pcfich_decode(&q->pcfich, &q->sf, &q->chest_res, q->sf_symbols, &corr);
  *cfi       = q->sf.cfi;
  *cfi_corr  = corr;
  return n;
}

static int blind_search(refdrv_dl_t* q, uint16_t rnti, srslte_dci_location_t* loc, uint32_t nloc, srslte_dci_format_t format, srslte_dci_msg_t* out)
{ /* dci_blind_search, ue_dl.c:422-478, first hit only (cif disabled) */
  for (uint32_t i = 0; i < nloc; i++) {
    srslte_dci_msg_t m;
    bzero(&m, sizeof(m));
    m.location = loc[i];
    m.format   = format;
    m.rnti     = 0;
    if (srslte_pdcch_decode_msg(&q->pdcch, &q->sf, &q->dci_cfg, &m)) return -1;
    if (m.rnti == rnti && m.nof_bits > 0 && m.format == format) {
      *out = m;
      return 1;
    }
  }
  return 0;
}

/* srslte_ue_dl_find_dl_dci (ue_dl.c:620-646): common search space, formats 1A then 1C for SI/P/RA-RNTI; for a C-RNTI the UE search
 * space with the two formats of the transmission mode (TM1/TM2: 1A, 1) and then 1A in the common space. Returns 1 if found. */
int refdrv_dl_find_dci(refdrv_dl_t* q, uint16_t rnti, int tm, uint32_t* mcs, int* tbs, uint32_t* nof_prb, int* rv)
{
  srslte_dci_location_t loc[REFDRV_MAX_CAND];
  srslte_dci_msg_t      msg;
  int                   found = 0;
  bzero(&q->dci_cfg, sizeof(q->dci_cfg));
  if (rnti == SRSLTE_SIRNTI || rnti == SRSLTE_PRNTI || SRSLTE_RNTI_ISRAR(rnti)) {
    uint32_t            n      = srslte_pdcch_common_locations(&q->pdcch, loc, REFDRV_MAX_CAND, q->sf.cfi);
    srslte_dci_format_t fmts[] = {SRSLTE_DCI_FORMAT1A, SRSLTE_DCI_FORMAT1C};
    for (int f = 0; f < 2 && n > 0 && !found; f++) {
      if ((found = blind_search(q, rnti, loc, n, fmts[f], &msg)) < 0) return -1;
    }
  } else {
    uint32_t            n      = srslte_pdcch_ue_locations(&q->pdcch, &q->sf, loc, REFDRV_MAX_CAND, rnti);
    srslte_dci_format_t fmts[] = {SRSLTE_DCI_FORMAT1A, SRSLTE_DCI_FORMAT1};
    for (int f = 0; f < 2 && !found; f++) {
      if ((found = blind_search(q, rnti, loc, n, fmts[f], &msg)) < 0) return -1;
    }
    if (!found) {
      n = srslte_pdcch_common_locations(&q->pdcch, loc, REFDRV_MAX_CAND, q->sf.cfi);
      if (n > 0 && (found = blind_search(q, rnti, loc, n, SRSLTE_DCI_FORMAT1A, &msg)) < 0) return -1;
    }
  }
  if (!found) return 0;
  bzero(&q->dci_dl, sizeof(q->dci_dl));
  if (srslte_dci_msg_unpack_pdsch(&q->cell, &q->sf, &q->dci_cfg, &msg, &q->dci_dl)) return -1;
  if (srslte_ra_dl_dci_to_grant(&q->cell, &q->sf, (srslte_tm_t)tm, false, &q->dci_dl, &q->pdsch_cfg.grant)) return -1;
  q->pdsch_cfg.rnti = rnti;
  if (q->pdsch_cfg.grant.tb[0].rv < 0) { /* ue_dl.c:1310-1316 */
    uint32_t sfn               = q->sf.tti / 10;
    uint32_t k                 = (sfn / 2) % 4;
    q->pdsch_cfg.grant.tb[0].rv = ((uint32_t)ceilf((float)1.5 * k)) % 4;
  }
  if (mcs) *mcs = q->dci_dl.tb[0].mcs_idx;
  if (tbs) *tbs = q->pdsch_cfg.grant.tb[0].tbs;
  if (nof_prb) *nof_prb = q->pdsch_cfg.grant.nof_prb;
  if (rv) *rv = q->pdsch_cfg.grant.tb[0].rv;
  return 1;
}

/* A known grant instead of a decoded DCI (what phy_dl_test.c:463-485 transmits: format 1, type-0 allocation, one transport block). */
int refdrv_dl_set_grant(refdrv_dl_t* q, uint32_t tti, uint32_t cfi, uint16_t rnti, int tm, uint32_t rbg_bitmask, uint32_t mcs, int rv, int use_tbs_index_alt,
                        int* tbs, uint32_t* nof_re)
{
  bzero(&q->sf, sizeof(q->sf));
  q->sf.tti = tti;
  q->sf.cfi = cfi;
  bzero(&q->dci_dl, sizeof(q->dci_dl));
  q->dci_dl.rnti                    = rnti;
  q->dci_dl.format                  = SRSLTE_DCI_FORMAT1;
  q->dci_dl.alloc_type              = SRSLTE_RA_ALLOC_TYPE0;
  q->dci_dl.type0_alloc.rbg_bitmask = rbg_bitmask;
  q->dci_dl.tb[0].mcs_idx           = mcs;
  q->dci_dl.tb[0].rv                = rv;
  SRSLTE_DCI_TB_DISABLE(q->dci_dl.tb[1]);
  q->pdsch_cfg.use_tbs_index_alt = use_tbs_index_alt;
  if (srslte_ra_dl_dci_to_grant(&q->cell, &q->sf, (srslte_tm_t)tm, use_tbs_index_alt, &q->dci_dl, &q->pdsch_cfg.grant)) return -1;
  q->pdsch_cfg.rnti = rnti;
  if (tbs) *tbs = q->pdsch_cfg.grant.tb[0].tbs;
  if (nof_re) *nof_re = q->pdsch_cfg.grant.nof_re;
  return 0;
}

/* Replace the PRB allocation of the current grant by arbitrary per-slot masks (srslte_pdsch_grant_t.prb_idx[s][n], what
 * srslte_ra_dl_grant_to_grant_prb_allocation fills for type 0 / 1 / 2-localized / 2-distributed allocations, ra_dl.c) and recompute
 * nof_re / nof_bits as srslte_ra_dl_dci_to_grant does (ra_dl.c:645). Transport block size and modulation stay those of the MCS and
 * PRB count the grant was made with. Returns nof_re. */
int refdrv_dl_set_prb_masks(refdrv_dl_t* q, const uint8_t* slot0, const uint8_t* slot1)
{
  for (uint32_t n = 0; n < q->cell.nof_prb; n++) {
    q->pdsch_cfg.grant.prb_idx[0][n] = slot0[n] != 0;
    q->pdsch_cfg.grant.prb_idx[1][n] = slot1[n] != 0;
  }
  srslte_ra_dl_compute_nof_re(&q->cell, &q->sf, &q->pdsch_cfg.grant);
  return (int)q->pdsch_cfg.grant.nof_re;
}

/* what the current grant says: PRB masks of both slots, modulation (srslte_mod_t), TBS, nof_re, nof_bits, rv */
void refdrv_dl_grant_info(refdrv_dl_t* q, uint8_t* slot0, uint8_t* slot1, int* mod, int* tbs, uint32_t* nof_re, uint32_t* nof_bits, int* rv)
{
  for (uint32_t n = 0; n < q->cell.nof_prb; n++) {
    slot0[n] = q->pdsch_cfg.grant.prb_idx[0][n];
    slot1[n] = q->pdsch_cfg.grant.prb_idx[1][n];
  }
  *mod      = (int)q->pdsch_cfg.grant.tb[0].mod;
  *tbs      = q->pdsch_cfg.grant.tb[0].tbs;
  *nof_re   = q->pdsch_cfg.grant.nof_re;
  *nof_bits = q->pdsch_cfg.grant.tb[0].nof_bits;
  *rv       = q->pdsch_cfg.grant.tb[0].rv;
}

/* type-2 allocation (format 1A): localized or distributed virtual resource blocks, L_crb blocks from RB_start (ra_dl.c:206-290) */
int refdrv_dl_set_grant_type2(refdrv_dl_t* q, uint32_t tti, uint32_t cfi, uint16_t rnti, uint32_t L_crb, uint32_t RB_start, int distributed, int n_gap2,
                              uint32_t mcs, int rv, int* tbs, uint32_t* nof_re)
{
  bzero(&q->sf, sizeof(q->sf));
  q->sf.tti = tti;
  q->sf.cfi = cfi;
  bzero(&q->dci_dl, sizeof(q->dci_dl));
  q->dci_dl.rnti              = rnti;
  q->dci_dl.format            = SRSLTE_DCI_FORMAT1A;
  q->dci_dl.alloc_type        = SRSLTE_RA_ALLOC_TYPE2;
  q->dci_dl.type2_alloc.mode  = distributed ? SRSLTE_RA_TYPE2_DIST : SRSLTE_RA_TYPE2_LOC;
  q->dci_dl.type2_alloc.n_gap = n_gap2 ? SRSLTE_RA_TYPE2_NG2 : SRSLTE_RA_TYPE2_NG1;
  q->dci_dl.type2_alloc.riv   = srslte_ra_type2_to_riv(L_crb, RB_start, q->cell.nof_prb);
  q->dci_dl.tb[0].mcs_idx     = mcs;
  q->dci_dl.tb[0].rv          = rv;
  SRSLTE_DCI_TB_DISABLE(q->dci_dl.tb[1]);
  /* transmit diversity on a cell with more than one port, as a format-1A grant gets it (ra_dl.c:530-552) */
  if (srslte_ra_dl_dci_to_grant(&q->cell, &q->sf, q->cell.nof_ports > 1 ? SRSLTE_TM2 : SRSLTE_TM1, false, &q->dci_dl, &q->pdsch_cfg.grant)) return -1;
  q->pdsch_cfg.rnti = rnti;
  if (tbs) *tbs = q->pdsch_cfg.grant.tb[0].tbs;
  if (nof_re) *nof_re = q->pdsch_cfg.grant.nof_re;
  return 0;
}

/* srslte_pdsch_encode on the current grant (eNB side of the same object family: needs its own srslte_pdsch_t in eNB mode): used to
 * pin the oracle's transmitter for arbitrary allocations. grid_out: the port-0 resource grid, zero outside the PDSCH. */
int refdrv_dl_encode_pdsch(refdrv_dl_t* q, const uint8_t* data, cf_t* grid_out)
{
  static srslte_pdsch_t          tx;
  static srslte_softbuffer_tx_t  sb;
  static bool                    init = false;
  static uint32_t                tx_prb = 0, tx_id = 0xffffffff;
  if (!init || tx_prb != q->cell.nof_prb || tx_id != q->cell.id) {
    if (init) {
      srslte_pdsch_free(&tx);
      srslte_softbuffer_tx_free(&sb);
    }
    if (srslte_pdsch_init_enb(&tx, q->cell.nof_prb) || srslte_pdsch_set_cell(&tx, q->cell) || srslte_softbuffer_tx_init(&sb, q->cell.nof_prb)) return -1;
    init = true; tx_prb = q->cell.nof_prb; tx_id = q->cell.id;
  }
  srslte_pdsch_cfg_t cfg = q->pdsch_cfg;
  cfg.softbuffers.tx[0]  = &sb;
  srslte_softbuffer_tx_reset_tbs(&sb, (uint32_t)cfg.grant.tb[0].tbs);
  cf_t*    grids[SRSLTE_MAX_PORTS] = {grid_out, NULL, NULL, NULL};
  uint8_t* d[SRSLTE_MAX_CODEWORDS] = {(uint8_t*)data, NULL};
  bzero(grid_out, sizeof(cf_t) * SRSLTE_SF_LEN_RE(q->cell.nof_prb, q->cell.cp));
  return srslte_pdsch_encode(&tx, &q->sf, &cfg, d, grids);
}

/* channel estimate only (srslte_chest_dl_estimate_cfg) for the subframe set by refdrv_dl_set_grant */
int refdrv_dl_chest(refdrv_dl_t* q) { return srslte_chest_dl_estimate_cfg(&q->chest, &q->sf, &q->chest_cfg, q->sf_symbols, &q->chest_res); }

/* ue_dl.c:1318-1350: reset the soft buffer (new data) and srslte_pdsch_decode. Returns the CRC flag, <0 on error. */
int refdrv_dl_decode_pdsch(refdrv_dl_t* q, int new_data, float* avg_iterations)
{
  if (new_data) srslte_softbuffer_rx_reset_tbs(q->pdsch_cfg.softbuffers.rx[0], (uint32_t)q->pdsch_cfg.grant.tb[0].tbs);
  srslte_pdsch_res_t res[SRSLTE_MAX_CODEWORDS];
  bzero(res, sizeof(res));
  res[0].payload = q->payload;
  if (srslte_pdsch_decode(&q->pdsch, &q->sf, &q->pdsch_cfg, &q->chest_res, q->sf_symbols, res)) return -1;
  if (avg_iterations) *avg_iterations = res[0].avg_iterations_block;
  return res[0].crc ? 1 : 0;
}

/* pmch_file_test.c:187-210: after refdrv_dl_estimate(..., sf_type = 1): the forced MBSFN grant and srslte_pmch_decode */
int refdrv_dl_pmch_decode(refdrv_dl_t* q, uint32_t cfi, uint16_t area_id, uint32_t mcs, int* tbs)
{
  q->sf.cfi = cfi;
  srslte_pmch_cfg_t pmch_cfg;
  bzero(&pmch_cfg, sizeof(pmch_cfg));
  pmch_cfg.area_id                     = area_id;
  pmch_cfg.pdsch_cfg.softbuffers.rx[0] = &q->softbuffer;
  srslte_softbuffer_rx_reset(&q->softbuffer);
  srslte_dci_dl_t dci;
  bzero(&dci, sizeof(dci));
  dci.rnti                    = SRSLTE_MRNTI;
  dci.format                  = SRSLTE_DCI_FORMAT1;
  dci.alloc_type              = SRSLTE_RA_ALLOC_TYPE0;
  dci.type0_alloc.rbg_bitmask = 0xffffffff;
  dci.tb[0].mcs_idx           = mcs;
  SRSLTE_DCI_TB_DISABLE(dci.tb[1]);
  if (srslte_ra_dl_dci_to_grant(&q->cell, &q->sf, SRSLTE_TM1, false, &dci, &pmch_cfg.pdsch_cfg.grant)) return -1;
  if (tbs) *tbs = pmch_cfg.pdsch_cfg.grant.tb[0].tbs;
  srslte_pdsch_res_t res;
  bzero(&res, sizeof(res));
  res.payload = q->payload;
  if (srslte_pmch_decode(&q->pmch, &q->sf, &pmch_cfg, &q->chest_res, q->sf_symbols, &res) < 0) return -1;
  return res.crc ? 1 : 0;
}

/* pbch_file_test.c:182-214: srslte_chest_dl_estimate on a slot-1 grid placed as a subframe, then srslte_pbch_decode. Returns the
 * decode result (1 = MIB found), nof_ports and sfn offset. The caller hands over the whole subframe's grid. */
int refdrv_dl_pbch_decode(refdrv_dl_t* q, uint32_t* nof_tx_ports, int* sfn_offset, uint8_t* bch_payload)
{
  if (!q->have_pbch) {
    if (srslte_pbch_init(&q->pbch) || srslte_pbch_set_cell(&q->pbch, q->cell)) return -1;
    q->have_pbch = true;
  }
  bzero(&q->sf, sizeof(q->sf));
  if (srslte_chest_dl_estimate(&q->chest, &q->sf, q->sf_symbols, &q->chest_res) < 0) return -1;
  srslte_pbch_decode_reset(&q->pbch);
  return srslte_pbch_decode(&q->pbch, &q->chest_res, q->sf_symbols, bch_payload, nof_tx_ports, sfn_offset);
}

/* CPU baseline of bench.py: `n` subframes of one known grant (all PRBs, format 1) through srslte_chest_dl_estimate_cfg +
 * srslte_pdsch_decode, the grids produced by `ofdm_rx` (the oracle's FFT: the reference's needs FFTW). One core, no Python in the
 * loop. iq: n subframes of sf_len samples; tb_out: n x tb_stride bytes; ok_out: n flags. Returns elapsed seconds, and in t_ofdm the
 * share of the OFDM callback. */
typedef void (*refdrv_ofdm_fn)(const void* ofdm_obj, const cf_t* in_time, cf_t* out_grid);
double refdrv_dl_rx_loop(refdrv_dl_t* q, refdrv_ofdm_fn ofdm_rx, const void* ofdm_obj, const cf_t* iq, uint32_t sf_len, uint32_t n, uint32_t tti0,
                         uint32_t cfi, uint16_t rnti, uint32_t mcs, int use_tbs_index_alt, uint8_t* tb_out, uint32_t tb_stride, uint8_t* ok_out, double* t_ofdm)
{
  struct timeval t0, t1, t2;
  double         ofdm_s = 0;
  gettimeofday(&t0, NULL);
  for (uint32_t b = 0; b < n; b++) {
    int tbs = 0;
    gettimeofday(&t1, NULL);
    ofdm_rx(ofdm_obj, &iq[(size_t)b * sf_len], q->sf_symbols[0]);
    gettimeofday(&t2, NULL);
    ofdm_s += (t2.tv_sec - t1.tv_sec) + 1e-6 * (t2.tv_usec - t1.tv_usec);
    if (refdrv_dl_set_grant(q, tti0 + b, cfi, rnti, 1, 0xffffffff, mcs, 0, use_tbs_index_alt, &tbs, NULL)) return -1;
    if (refdrv_dl_chest(q) < 0) return -1;
    int crc = refdrv_dl_decode_pdsch(q, 1, NULL);
    if (crc < 0) return -1;
    ok_out[b] = (uint8_t)crc;
    memcpy(&tb_out[(size_t)b * tb_stride], q->payload, (size_t)tbs / 8);
  }
  gettimeofday(&t1, NULL);
  if (t_ofdm) *t_ofdm = ofdm_s;
  return (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec);
}
