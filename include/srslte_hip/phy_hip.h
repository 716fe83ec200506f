/*
 * include/srslte_hip/phy_hip.h — batched C ABI of libsrslte_phy_hip.so (MI355X / gfx950).
 *
 * This is the throughput path of the drop-in (last row): device-resident buffers, one launch per
 * stage per batch of subframes. Every entry point is extern "C", takes plain pointers/sizes (device pointers are
 * marked d_), returns SRSLTE_SUCCESS 0 / SRSLTE_ERROR -1 / SRSLTE_ERROR_INVALID_INPUTS -2 (config.h:58-66) and
 * names the reference interface it replaces (paths relative to the reference tree).
 * The single-subframe srslte_* look-alikes that a caller such as lib/src/phy/ue/ue_dl.c binds are declared in
 * include/srslte_hip/srslte_compat.h and are thin host wrappers over these.
 *
 * Layouts (all cf_t = interleaved float re,im):
 *   time samples  [nof_sf][15*N]            N = srslte_symbol_sz(nof_prb)
 *   resource grid [nof_sf][nsym][12*prb]    nsym = 14 (normal CP) / 12, sub-carrier ascending, DC removed
 *   LLRs          [nof_sf][nof_re*Qm]       bit order b0(I) b1(Q) b2 ... as demod_soft.c
 */
#ifndef SRSLTE_HIP_PHY_HIP_H
#define SRSLTE_HIP_PHY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ device plumbing (no HIP headers needed by callers) */
int   srslte_hip_device_count(void);
int   srslte_hip_set_device(int device);
void* srslte_hip_malloc(size_t nbytes);
void  srslte_hip_free(void* d_ptr);
int   srslte_hip_memcpy_h2d(void* d_dst, const void* h_src, size_t nbytes);
int   srslte_hip_memcpy_d2h(void* h_dst, const void* d_src, size_t nbytes);
int   srslte_hip_memset(void* d_dst, int value, size_t nbytes);
int   srslte_hip_sync(void);
void* srslte_hip_stream_create(void);
void  srslte_hip_stream_destroy(void* stream);
int   srslte_hip_stream_sync(void* stream);
void* srslte_hip_event_create(void);                      /* HIP events on the caller's stream, for kernel timing */
int   srslte_hip_event_record(void* event, void* stream);
float srslte_hip_event_elapsed_ms(void* start, void* stop);
void  srslte_hip_event_destroy(void* event);
/* The single-call (compat) layer gives every host thread its own stream (SURVEY 8b "Threading": one object per worker thread, calls from
 * several threads at once). Returns the CALLING thread's stream, created on first use, and its hipStreamGetFlags value in *flags. */
void* srslte_hip_compat_thread_stream(unsigned* flags);
/* Counters of the compat layer since process start: [0] stream waits (one per synchronous call), [1] srslte_dlsch_decode2 calls served by the
 * device pipeline, [2] single-code-block decoder calls (srslte_tdec_run_all / _iteration), [3] reserved. With SRSLTE_HIP_STATS set in the
 * environment the library prints them to stderr at exit ("[srslte_hip] stats: ..."): how a caller's processes used the boundary. */
void  srslte_hip_compat_stats(unsigned long long out[4]);

/* ------------------------------------------------------------------ OFDM (replaces srslte_ofdm_rx_sf / srslte_ofdm_tx_sf,
 * lib/include/srslte/phy/dft/ofdm.h:82-153, lib/src/phy/dft/ofdm.c:384-594, and FFTW behind dft_fftw.c) */
typedef struct srslte_hip_ofdm srslte_hip_ofdm_t;
srslte_hip_ofdm_t* srslte_hip_ofdm_create(int nof_prb, int cp_is_norm, int is_rx);     /* ofdm.c:235-273 */
/* the same with the symbol size given, as srslte_ofdm_init_ takes it (ofdm.c:38-57): srslte_symbol_sz(nof_prb) of either rate family - 128 / 256 /
 * 384 / 768 / 1024 / 1536, or 128 / 256 / 512 / 1024 / 1536 / 2048 with srslte_use_standard_symbol_size(true) (phy_common.c:304-345) */
srslte_hip_ofdm_t* srslte_hip_ofdm_create_sz(int nof_prb, int symbol_sz, int cp_is_norm, int is_rx);
void               srslte_hip_ofdm_destroy(srslte_hip_ofdm_t* q);                      /* ofdm.c:214-233 */
int                srslte_hip_ofdm_set_normalize(srslte_hip_ofdm_t* q, int enable);    /* ofdm.c:576-578 */
int                srslte_hip_ofdm_set_freq_shift(srslte_hip_ofdm_t* q, float shift);  /* ofdm.c:360-378 */
int                srslte_hip_ofdm_symbol_sz(const srslte_hip_ofdm_t* q);
int                srslte_hip_ofdm_sf_len(const srslte_hip_ofdm_t* q);
int srslte_hip_ofdm_rx_sf_batch(srslte_hip_ofdm_t* q, const void* d_in_time, void* d_out_grid, int nof_sf, void* stream); /* ofdm.c:453-467 */
int srslte_hip_ofdm_tx_sf_batch(srslte_hip_ofdm_t* q, const void* d_in_grid, void* d_out_time, int nof_sf, void* stream); /* ofdm.c:580-594 */
/* MBSFN subframe layout on an extended-CP object (srslte_ofdm_{rx,tx}_init_mbsfn + srslte_ofdm_set_non_mbsfn_region,
 * ofdm.c:120-137,:248-258,:286-305; slot layout of srslte_ofdm_rx_slot_mbsfn :424-437 / srslte_ofdm_tx_slot_mbsfn :558-574).
 * The samples of the guard between the two regions are neither read (rx) nor written (tx). */
int srslte_hip_ofdm_set_mbsfn(srslte_hip_ofdm_t* q, int enable, int non_mbsfn_region);
/* one slot of each subframe (srslte_ofdm_rx_slot/_tx_slot ofdm.c:398-422,:488-530; mbsfn_layout=1: the MBSFN slot-0 layout);
 * rx or tx according to the object; d_in/d_out are subframe bases as in the _sf_batch calls */
int srslte_hip_ofdm_slot_batch(srslte_hip_ofdm_t* q, const void* d_in, void* d_out, int nof_sf, int slot_in_sf, int mbsfn_layout, void* stream);

/* generic batched c2c DFT (replaces srslte_dft_run_guru_c, dft.h:137-152, dft_fftw.c:137-165,307-313) */
int srslte_hip_dft_batch(const void* d_in, void* d_out, int N, int howmany, int idist, int odist, int forward, float scale, void* stream);
/* SC-FDMA transform precoding (replaces srslte_dft_precoding, dft_precoding.h:39-64, dft_precoding.c:88-113) */
int srslte_hip_dft_precoding_valid_prb(uint32_t nof_prb);
int srslte_hip_dft_precoding_batch(const void* d_in, void* d_out, uint32_t nof_prb, uint32_t nof_symbols, int forward, void* stream);

/* ------------------------------------------------------------------ DL channel estimator (replaces srslte_chest_dl_estimate_cfg,
 * ch_estimation/chest_dl.h:49-156, chest_dl.c:598-908) */
typedef struct srslte_hip_chest_dl srslte_hip_chest_dl_t;
typedef struct {             /* same members, order and meaning as srslte_chest_dl_cfg_t (chest_dl.h:116-130) */
  int      noise_alg;        /* 0 REFS, 1 PSS, 2 EMPTY (chest_dl.h:85-89); PSS / EMPTY renew the estimate in subframes 0 and 5 and report the
                              * kept one otherwise, carried through the batch in subframe order and between calls on the object */
  int      filter_type;      /* 0 GAUSS, 1 TRIANGLE, 2 NONE (chest_common.h:30-34) */
  float    filter_coef[2];
  uint16_t mbsfn_area_id;
  uint8_t  interpolate_subframe;
  uint8_t  rsrp_neighbour;
  uint8_t  cfo_estimate_enable;
  uint32_t cfo_estimate_sf_mask;
  uint8_t  sync_error_enable;
} srslte_hip_chest_dl_cfg_t;
typedef struct {             /* scalar members of srslte_chest_dl_res_t (chest_dl.h:49-67), one per subframe */
  float noise_estimate, noise_estimate_dbm, snr_db, rsrp, rsrp_dbm, rsrq, rsrq_db, rssi_dbm, cfo, sync_error;
} srslte_hip_chest_dl_res_t;
/* cp_is_norm = 0: extended-CP cell, 12 symbols per subframe: every "[14]" below reads "[12]" then (CRS on symbols 0, 3, 6, 9; chest_dl.c:497-502) */
srslte_hip_chest_dl_t* srslte_hip_chest_dl_create(uint32_t cell_id, uint32_t nof_prb, uint32_t nof_ports, int cp_is_norm); /* chest_dl.c:69-160,193-300 */
/* TDD cell (srslte_cell_t.frame_type = SRSLTE_TDD; sf_config / ss_config = srslte_tdd_config_t of srslte_dl_sf_cfg_t, phy_common.h:381-388): special
 * subframes are estimated from the CRS symbols their DwPTS holds (refsignal_dl.c:162-225). sf_config < 0: FDD again. */
int srslte_hip_chest_dl_set_tdd(srslte_hip_chest_dl_t* q, int sf_config, int ss_config);
void                   srslte_hip_chest_dl_destroy(srslte_hip_chest_dl_t* q);
/* the symbol size the CFO and timing-error estimates scale with (chest_dl.c:575,:695: srslte_symbol_sz(cell.nof_prb), read at every call);
 * default: the default rate family's; e.g. 2048 for 100 PRB after srslte_use_standard_symbol_size(true) */
int                    srslte_hip_chest_dl_set_symbol_sz(srslte_hip_chest_dl_t* q, int symbol_sz);
const void*            srslte_hip_chest_dl_pilots(const srslte_hip_chest_dl_t* q); /* device CRS table [10][4][2*prb] (refsignal_dl.c:66-116) */
int srslte_hip_chest_dl_estimate_batch(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0, const void* d_grid,
                                       void* d_ce, void* d_res, int nof_sf, void* stream);
/* nof_rx receive antennas x the object's 1 or 2 tx ports (chest_dl.c:884-908 loops over antennas and ports; fill_res :747-871
 * combines them): d_grid is [nof_sf][nof_rx][14][12*nof_prb], d_ce [nof_sf][nof_ports][nof_rx][14][12*nof_prb], d_res stays one
 * entry per subframe. 4-port cells: not with cfg->interpolate_subframe (upstream's result is undefined there, chest_dl.c:467-471). */
int srslte_hip_chest_dl_estimate_batch_multi(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0, const void* d_grid,
                                             void* d_ce, void* d_res, int nof_sf, int nof_rx, void* stream);
/* device pointer to [nof_sf][nof_ports][nof_rx] x {noise_estimate, rsrp, rssi, cfo, sync_err, rsrp_corr} (6 floats) of the last call
 * with d_res: the per-antenna / per-port terms of fill_res (chest_dl.c:860-870) and of get_rsrp_neighbour (:821-843) */
const float* srslte_hip_chest_dl_last_raw(const srslte_hip_chest_dl_t* q);
/* MBSFN subframes (srslte_chest_dl_set_mbsfn_area_id chest_dl.c:244-262 with the reference signal of refsignal_dl.c:361-400,
 * estimate_port_mbsfn :718-745 and the MBSFN branches of :304-556). 1- and 2-port cells, cfg->interpolate_subframe set (the reference's
 * result without it is undefined), area id from cfg->mbsfn_area_id. d_grid [nof_sf][nof_rx][14][12*nof_prb] holds the 12 symbols of
 * the extended-CP subframe; d_ce [nof_sf][nof_ports][nof_rx][14][12*nof_prb] gets symbols 0-11. d_noise (or NULL):
 * [nof_sf][nof_ports][nof_rx] REFS noise estimates, written only with cfg->noise_alg == REFS. As in the reference an MBSFN subframe
 * measures nothing else: rsrp, rssi, cfo and the sync error keep the values of the last normal subframe (the compat layer keeps them). */
int         srslte_hip_chest_dl_set_mbsfn_area_id(srslte_hip_chest_dl_t* q, uint16_t mbsfn_area_id);
const void* srslte_hip_chest_dl_mbsfn_pilots(const srslte_hip_chest_dl_t* q, uint16_t mbsfn_area_id); /* device [10][3][6*nof_prb] or NULL */
int srslte_hip_chest_dl_estimate_mbsfn_batch(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0, const void* d_grid,
                                             void* d_ce, float* d_noise, int nof_sf, int nof_rx, void* stream);

/* ------------------------------------------------------------------ UL channel estimator (replaces
 * srslte_chest_ul_init/_set_cell/_pregen/_estimate_pusch, ch_estimation/chest_ul.h:47-104, chest_ul.c:51-327, and the PUSCH DMRS of
 * refsignal_ul.c:118-487). Normal CP, grants of >= 1 PRB (the 1- and 2-PRB base sequences from the tables of 36.211 5.5.1.2), same
 * allocation in both slots (the reference's estimator does not support intra-subframe hopping either, chest_ul.c:297-299). */
typedef struct srslte_hip_chest_ul srslte_hip_chest_ul_t;
typedef struct { /* srslte_refsignal_dmrs_pusch_cfg_t, refsignal_ul.h:46-51 */
  uint32_t cyclic_shift, delta_ss;
  int      group_hopping_en, sequence_hopping_en;
} srslte_hip_dmrs_pusch_cfg_t;
typedef struct { /* scalar part of srslte_chest_ul_res_t */
  float noise_estimate, noise_estimate_dbm, snr, snr_db, cfo;
} srslte_hip_chest_ul_res_t;
srslte_hip_chest_ul_t* srslte_hip_chest_ul_create(uint32_t cell_id, uint32_t nof_prb, int cp_is_norm, const srslte_hip_dmrs_pusch_cfg_t* cfg);
void                   srslte_hip_chest_ul_destroy(srslte_hip_chest_ul_t* q);
/* srslte_refsignal_dmrs_pusch_gen (refsignal_ul.c:459-487): r_host [2][12*L_prb] cf32 in HOST memory */
int srslte_hip_refsignal_dmrs_pusch_gen(const srslte_hip_chest_ul_t* q, uint32_t L_prb, uint32_t sf_idx, uint32_t n_dmrs, void* r_host);
/* d_grid [nof_sf][14][12*nof_prb]; d_ce same shape or NULL - only the granted PRBs are written, as upstream; d_res [nof_sf] or NULL;
 * subframe b is TTI tti0 + b; one grant (L_prb, n_prb, n_dmrs) for the batch */
int srslte_hip_chest_ul_estimate_pusch_batch(srslte_hip_chest_ul_t* q, uint32_t tti0, uint32_t L_prb, uint32_t n_prb, uint32_t n_dmrs,
                                             const void* d_grid, void* d_ce, void* d_res, int nof_sf, void* stream);
/* the same with a PRB offset per slot (srslte_pusch_grant_t.n_prb[0 / 1], intra-subframe hopping: chest_ul.c:244-266,:293-295) */
int srslte_hip_chest_ul_estimate_pusch_batch_hop(srslte_hip_chest_ul_t* q, uint32_t tti0, uint32_t L_prb, uint32_t n_prb, uint32_t n_prb_slot1,
                                                 uint32_t n_dmrs, const void* d_grid, void* d_ce, void* d_res, int nof_sf, void* stream);

/* ------------------------------------------------------------------ soft demapper (replaces srslte_demod_soft_demodulate{,_s,_b},
 * modem/demod_soft.h:39-53, demod_soft.c:479-549). mod: 0 BPSK, 1 QPSK, 2 16QAM, 3 64QAM, 4 256QAM (srslte_mod_t).
 * ncalls independent calls of nsymbols each (the scalar-tail rounding of the reference depends on nsymbols). */
int srslte_hip_demod_soft_demodulate_batch(int mod, const void* d_symbols, float* d_llr, int nsymbols, int ncalls, void* stream);
int srslte_hip_demod_soft_demodulate_s_batch(int mod, const void* d_symbols, short* d_llr, int nsymbols, int ncalls, void* stream);
int srslte_hip_demod_soft_demodulate_b_batch(int mod, const void* d_symbols, int8_t* d_llr, int nsymbols, int ncalls, void* stream);

/* ------------------------------------------------------------------ turbo decoder (replaces srslte_tdec_run_all / srslte_tdec_iteration,
 * fec/turbodecoder.h:63-135, turbodecoder.c:146-593, turbodecoder_iter.h:71-139, turbodecoder_win.h, turbodecoder_gen.c) */
typedef struct srslte_hip_tdec srslte_hip_tdec_t;
srslte_hip_tdec_t* srslte_hip_tdec_create(uint32_t max_long_cb, uint32_t max_nof_cb);
void               srslte_hip_tdec_destroy(srslte_hip_tdec_t* q);
uint32_t           srslte_hip_tdec_autoimp_get_subblocks(uint32_t long_cb);           /* turbodecoder.c:394-406 */
uint32_t           srslte_hip_tdec_input_len(uint32_t long_cb, int sb_layout);        /* int16 per code block */
/* Decodes nof_cb code blocks of equal length long_cb.
 *   d_input: [nof_cb][in_stride] int16; layout = [s p0 p1]*K + 12 tail (sb_layout = 0, srslte_tdec_force_not_sb)
 *            or the rm_turbo "SB" layout 3*(K+32)+12 (sb_layout = 1; turbodecoder_iter.h:84-91)
 *   nof_iterations: SISO passes (one srslte_tdec_iteration each)
 *   crc_poly: 0 = run all passes (srslte_tdec_run_all); else stop a block at the first pass whose hard decision has
 *             a zero CRC-24 remainder over crc_nbits bits (sch.c:353-383)
 *   d_output: [nof_cb][out_stride] bytes, K/8 per block, MSB first; d_iters/d_crc_ok: [nof_cb] or NULL */
int srslte_hip_tdec_run_batch(srslte_hip_tdec_t* q, const int16_t* d_input, uint32_t in_stride, int sb_layout, uint32_t long_cb,
                              uint32_t nof_cb, uint32_t nof_iterations, uint32_t crc_poly, uint32_t crc_nbits, uint8_t* d_output,
                              uint32_t out_stride, uint32_t* d_iters, uint8_t* d_crc_ok, void* stream);

/* srslte_tdec_init_manual equivalent (turbodecoder.c:168-215): nof_subblocks 0 = generic, 8 = sse16, 16 = avx16 numerics on any K */
int srslte_hip_tdec_run_batch_manual(srslte_hip_tdec_t* q, const int16_t* d_input, uint32_t in_stride, int sb_layout, uint32_t long_cb,
                                     uint32_t nof_subblocks, uint32_t nof_cb, uint32_t nof_iterations, uint32_t crc_poly, uint32_t crc_nbits,
                                     uint8_t* d_output, uint32_t out_stride, uint32_t* d_iters, uint8_t* d_crc_ok, void* stream);

/* 8-bit LLRs (replaces srslte_tdec_run_all_8bit / srslte_tdec_iteration_8bit, turbodecoder.h:117-135, turbodecoder.c:438-469,
 * :565-593). Back-end per K as AUTO selects on an AVX2 host: K > 2048 avx8 (32 windows), K > 800 sse8 (16 windows)
 * - saturating int8, max-normalisation every step, output >> 1 (turbodecoder_win.h:92-173) - and below that the LLRs are widened
 * and the 16-bit back-ends run. sb_layout as above with 8-bit elements (srslte_rm_turbo_rx_lut_8bit output). */
uint32_t srslte_hip_tdec_autoimp_get_subblocks_8bit(uint32_t long_cb); /* turbodecoder.c:421-436 */
int srslte_hip_tdec_run_batch_8bit(srslte_hip_tdec_t* q, const int8_t* d_input, uint32_t in_stride, int sb_layout, uint32_t long_cb,
                                   uint32_t nof_cb, uint32_t nof_iterations, uint32_t crc_poly, uint32_t crc_nbits, uint8_t* d_output,
                                   uint32_t out_stride, uint32_t* d_iters, uint8_t* d_crc_ok, void* stream);

/* ------------------------------------------------------------------ turbo encoder (replaces srslte_tcod_encode, fec/turbocoder.h:44-76,
 * turbocoder.c:76-186): bits in (one per byte) -> 3K+12 bits out ([s p0 p1] triplets + 12 tail), nof_cb blocks */
int srslte_hip_tcod_encode_batch(const uint8_t* d_input, uint8_t* d_output, uint32_t long_cb, uint32_t nof_cb, void* stream);
/* byte-packed form (replaces the encoder proper of srslte_tcod_encode_lut, turbocoder.c:189-367; the CRC attachment of that
 * call stays with the caller): d_input [nof_cb][in_stride] K/8 bytes MSB first; d_parity [nof_cb][par_stride] K/4+1 bytes =
 * p1[K] t1[4] p2[K] t2[4] as one bit stream; d_sys_tail [nof_cb] = the byte the reference stores in input[K/8] */
int srslte_hip_tcod_encode_bytes_batch(const uint8_t* d_input, uint32_t in_stride, uint8_t* d_parity, uint32_t par_stride, uint8_t* d_sys_tail,
                                       uint32_t long_cb, uint32_t nof_cb, void* stream);

/* ------------------------------------------------------------------ segmentation / interleaver (host, replaces cbsegm.c, tc_interl_lte.c) */
typedef struct { /* same members and order as srslte_cbsegm_t (cbsegm.h:33-44) */
  uint32_t F, C, K1, K2, K1_idx, K2_idx, C1, C2, tbs;
} srslte_hip_cbsegm_t;
int srslte_hip_cbsegm(srslte_hip_cbsegm_t* s, uint32_t tbs);
int srslte_hip_cbsegm_cbindex(uint32_t long_cb);
int srslte_hip_cbsegm_cbsize(uint32_t index);
int srslte_hip_tc_interl_LTE_gen_interl(uint16_t* forward, uint16_t* reverse, uint32_t long_cb, uint32_t interl_win);

/* ------------------------------------------------------------------ PDSCH receive pipeline (SURVEY §8f N1 glue fused on device):
 * OFDM RX -> chest_dl -> RE extraction + one-tap MMSE -> soft demap + descramble -> turbo rate de-matching ->
 * turbo decode with CRC early stop -> TB CRC. One codeword, TM1 (single port) or TM2 (2-port transmit diversity), 1..4 rx antennas,
 * full-band grant, rv 0, FDD, normal CP. */
typedef struct srslte_hip_dl_rx srslte_hip_dl_rx_t;
typedef struct {
  uint32_t cell_id, nof_prb, cfi;
  uint16_t rnti;
  int      mod;            /* srslte_mod_t */
  uint32_t tbs;            /* transport block size, bits */
  uint32_t max_iterations; /* SISO passes, sch.c:114 */
  uint32_t max_batch;      /* subframes per call */
  int      mmse;           /* 1: noise_estimate from chest (pdsch.c:862), 0: ZF */
  srslte_hip_chest_dl_cfg_t chest_cfg;
  int      llr_8bit;       /* 1: the 8-bit LLR path the applications select (q->llr_is_8bit: pdsch.c:760-779 demod_b + int8
                              descrambling, sch.c:336-356 srslte_rm_turbo_rx_lut_8bit + srslte_tdec_iteration_8bit) */
  uint32_t nof_rx_antennas; /* 0 or 1: one antenna; 2..4: per-antenna estimation + srslte_predecoding_single_multi (precoding.c:325-348,
                               pdsch.c:890-935); d_iq / d_grid are then [nof_sf][nof_rx][...] */
  uint32_t nof_ports;       /* 0 or 1: single antenna port (TM1); 2 or 4: cell with that many ports and transmit diversity (TM2): multi-port
                               chest_dl and RE mapping, srslte_predecoding_diversity_multi + srslte_layerdemap_diversity (precoding.c:564-650,
                               layermap.c:140-148), for nof_rx_antennas 1..4; 4 ports: not with
                               chest_cfg.interpolate_subframe */
  int      csi_enable;      /* srslte_pdsch_cfg_t.csi_enable (pdsch_cfg.h:63; the srsUE default): LLRs weighted by each symbol's channel
                               gain relative to the subframe's largest (csi_correction, pdsch.c:574-690, applied inside the rate
                               de-matching kernels as they read the LLRs) */
  int      power_scale;     /* srslte_pdsch_cfg_t.power_scale / p_a (pdsch_cfg.h:58-62, pdsch.c:518-554,:852-858): the equaliser divides by
                               rho_a = 10^(p_a/20) (x sqrt(2) for a 2-port cell). Only p_b values with rho_b = 1 (no rescaling of the
                               CRS-bearing symbols) are covered */
  float    p_a;             /* dB */
  int      tx_scheme;       /* 0: by nof_ports as above. srslte_tx_scheme_t of the grant for the two-layer modes of a 2-port cell received with 2
                               antennas: SRSLTE_TXSCHEME_CDD (3): large-delay CDD, TM3, two transport blocks
                               (srslte_predecoding_ccd_2x2_mmse_csi, precoding.c:918-1014); SRSLTE_TXSCHEME_SPATIALMUX (2): closed-loop
                               multiplexing, TM4, two transport blocks with pmi 0-1 (srslte_predecoding_multiplex_2x2_mmse_csi :1326-1438) or
                               one with pmi 0-3 (srslte_predecoding_multiplex_2x1_mrc_csi :1624-1707). mmse = 0 zeroes the noise term
                               (pdsch.c:866). 16-bit LLRs */
  uint32_t pmi;             /* srslte_pdsch_grant_t.pmi */
  int      mod2;            /* grant.tb[1].mod */
  uint32_t tbs2;            /* grant.tb[1].tbs; 0: one transport block. With two, d_tb / d_tb_ok of the batch calls have 2 * nof_sf rows:
                               row b = transport block 0 of subframe b, row nof_sf + b = transport block 1; tb_stride covers the larger */
  int      cp_ext;          /* 1: extended-CP cell (srslte_cell_t.cp = SRSLTE_CP_EXT): 12 symbols per subframe - grids, estimates and RE lists
                               are [12][12 * nof_prb] -, CRS on symbols 0 and 3 of each slot, PSS / SSS on symbols 5 and 4 of slot 0
                               (ofdm.c:424-437, chest_dl.c:497-502, pdsch.c:81-206 with nof_symb_slot = 6) */
  int      tdd;             /* 1: TDD cell (srslte_cell_t.frame_type = SRSLTE_TDD) with tdd_sf_config (uplink-downlink configuration 0-6) and
                               tdd_ss_config (special-subframe configuration 0-9) = the srslte_tdd_config_t of its subframes. Served by the
                               per-subframe-grant entry points (srslte_hip_dl_rx_batch_grants*): SSS on the last symbol of slot 1 in subframes
                               0 / 5, PSS on symbol 2 of subframes 1 / 6, and in the special subframes PDSCH and CRS on the DwPTS symbols only
                               (pdsch.c:124-140, ra_dl.c:446-460, refsignal_dl.c:162-225); uplink subframes of the batch carry tbs = 0. The
                               fixed-grant calls refuse a TDD cell (their three subframe classes are FDD's) */
  uint32_t tdd_sf_config, tdd_ss_config;
  int      mbsfn;           /* 1: the batch is MBSFN subframes carrying the PMCH of MBSFN area mbsfn_area_id (0-255) - srslte_ofdm_rx_sf on an MBSFN
                               object with non_mbsfn_region (1 or 2) symbols in front (ofdm.c:424-437), srslte_chest_dl_estimate_cfg with sf_type
                               MBSFN (chest_dl.c:718-745; chest_cfg.interpolate_subframe is implied, the estimate is undefined without it) and
                               srslte_pmch_decode (pmch.c:291-394): every PRB from symbol SRSLTE_NOF_CTRL_SYMBOLS(cfi) on, every second RE in the
                               symbols of the MBSFN reference signal, the area's scrambling sequence (sequences.c:76-80), rv 0. Grids, estimates
                               and RE lists are [12][12 * nof_prb] whatever the cell's CP (cp_ext: the CRS sequence of symbol 0); single-port
                               cell, 16-bit LLRs, 1-4 receive antennas; mod / tbs = the PMCH's (srslte_configure_pmch). Fixed-grant calls only */
  uint32_t mbsfn_area_id, non_mbsfn_region;
} srslte_hip_dl_rx_cfg_t;
srslte_hip_dl_rx_t* srslte_hip_dl_rx_create(const srslte_hip_dl_rx_cfg_t* cfg);
void                srslte_hip_dl_rx_destroy(srslte_hip_dl_rx_t* q);
uint32_t            srslte_hip_dl_rx_nof_re(const srslte_hip_dl_rx_t* q, uint32_t sf_idx);
/* d_iq: [nof_sf][15*N]; outputs: d_tb [nof_sf][tb_stride] bytes (tbs/8 + 3 CRC bytes used), d_tb_ok [nof_sf]. The two output pointers may be
 * device-visible HOST memory (hipHostMalloc / a pinned allocation): the pipeline's last phase then stores the results where the MAC reads them and no
 * copy follows the batch (bench.py's N = 1 line; tests/test_gpu_fullsize.py::test_results_straight_into_pinned_host_memory) */
int srslte_hip_dl_rx_batch(srslte_hip_dl_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, uint8_t* d_tb, uint32_t tb_stride,
                           uint8_t* d_tb_ok, void* stream);
/* HARQ (decode_tb_cb sch.c:299-414 on a srslte_softbuffer_rx_t per transport block, softbuffer.c:46-150): slot b of the object keeps
 * its blocks' soft buffers, CRC flags and decoded bytes between calls. new_data != 0: new transport blocks (the MAC's
 * srslte_softbuffer_rx_reset_tbs on a toggled NDI). new_data == 0: retransmission with redundancy version rv, de-matched LLRs are
 * added to the kept soft buffers (rm_turbo.c:407-409), blocks whose CRC already passed are left alone (sch.c:317-318). A retransmission
 * into a slot whose transport block has already passed (the MAC does not schedule one) is answered as upstream answers it: d_tb_ok 0
 * (decode_tb_cb saves passed blocks' bytes only while the block as a whole has failed, sch.c:399-410, and the reassembled block fails its
 * CRC-24A); the same holds for every HARQ entry point of this header (uplink, per-subframe grants) */
int srslte_hip_dl_rx_batch_harq(srslte_hip_dl_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, uint32_t rv, int new_data,
                                uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* stream);
/* the same with a redundancy version and a new-data flag per transport block (two-layer modes: grant.tb[0 / 1].rv and the state of
 * softbuffers.rx[0 / 1]); srslte_hip_dl_rx_batch_harq applies one pair to both */
int srslte_hip_dl_rx_batch_harq2(srslte_hip_dl_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, const uint32_t rv[2], const int new_data[2],
                                 uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* stream);
/* same from frequency-domain grids d_grid [nof_sf][14][12*nof_prb] (the part of srslte_ue_dl_decode after srslte_ofdm_rx_sf,
 * ue_dl.c:375-397; SURVEY §8d cfg5 feeds grids) */
int srslte_hip_dl_rx_grid_batch(srslte_hip_dl_rx_t* q, const void* d_grid, uint32_t tti0, uint32_t nof_sf, uint8_t* d_tb, uint32_t tb_stride,
                                uint8_t* d_tb_ok, void* stream);
/* A new grant every subframe, as srslte_pdsch_decode takes it (pdsch.c:833-997 with a srslte_pdsch_grant_t per call, pdsch_cfg.h:38-50):
 * subframe b of the batch is received with grants[b] (host array). prb_mask[s]: bit n = srslte_pdsch_grant_t.prb_idx[s][n], the PRBs of
 * slot s (any subset; the two slots may differ, as with distributed virtual resource blocks), walked as srslte_pdsch_cp does
 * (pdsch.c:81-206). tbs = 0: no transport block in that subframe (tb_ok = 0). rv / new_data as srslte_hip_dl_rx_batch_harq, per subframe.
 * cfg.tbs of the object bounds every grant's tbs; cfg.mod / cfg.rnti / cfg.cfi are not used. Single antenna port or transmit diversity (cfg.nof_ports); cfg.llr_8bit, cfg.csi_enable
 * (csi_correction with every subframe's own allocation and modulation) and cfg.nof_rx_antennas apply.
 * With 16-bit LLRs the transport blocks are assembled and judged by the decoder launch itself (no assembly kernel behind it; blocks kept from an
 * earlier transmission contribute their stored bytes). Environment SRSLTE_HIP_GRANTS_TB_DIRECT=0, read when an object serves its first grants
 * call, keeps the separate assembly kernel (A/B, tests). Results are the same either way. */
typedef struct {
  uint32_t prb_mask[2][4];
  int      mod;      /* srslte_mod_t: 1 QPSK, 2 16QAM, 3 64QAM, 4 256QAM */
  uint32_t tbs;      /* bits */
  uint32_t rv;
  uint32_t cfi;
  uint16_t rnti;
  int      new_data; /* != 0: new transport block in HARQ slot b (soft buffer overwritten); 0: retransmission (combined) */
} srslte_hip_dl_grant_t;
int srslte_hip_dl_rx_batch_grants(srslte_hip_dl_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, const srslte_hip_dl_grant_t* grants,
                                  uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* stream);
/* The same with the transmission scheme of each subframe's grant and a second transport block (srslte_pdsch_grant_t.tx_scheme / pmi / tb[1],
 * as srslte_ra_dl_dci_to_grant fills them from DCI formats 1 / 1A / 2 / 2A, ra_dl.c:530-600): on a 2-port cell received with 2 antennas
 * (cfg.nof_ports = cfg.nof_rx_antennas = 2, cfg.tx_scheme = 0) a batch may mix transmit diversity (tx_scheme 0 or 1), large-delay CDD
 * (SRSLTE_TXSCHEME_CDD 3: two transport blocks) and closed-loop multiplexing (SRSLTE_TXSCHEME_SPATIALMUX 2: two blocks with pmi 0-1, or one
 * with pmi 0-3). Then d_tb / d_tb_ok have 2 * nof_sf rows: row b = transport block 0 of subframe b, row nof_sf + b = transport block 1
 * (tb_ok 0 where there is none); HARQ slot of block 1: its own. On other cells tx_scheme must be 0 / 1 and the rows are nof_sf. */
typedef struct {
  srslte_hip_dl_grant_t tb0; /* allocation, CFI, RNTI and transport block 0 */
  int      tx_scheme;         /* srslte_tx_scheme_t */
  uint32_t pmi;               /* srslte_pdsch_grant_t.pmi */
  int      mod2;              /* transport block 1: modulation, size (0: none), redundancy version, new-data flag */
  uint32_t tbs2, rv2;
  int      new_data2;
} srslte_hip_dl_grant2_t;
int srslte_hip_dl_rx_batch_grants2(srslte_hip_dl_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, const srslte_hip_dl_grant2_t* grants,
                                   uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* stream);
/* ... from frequency-domain grids d_grid, as srslte_hip_dl_rx_grid_batch takes them (the caller ran the OFDM demodulation) */
int srslte_hip_dl_rx_grid_batch_grants2(srslte_hip_dl_rx_t* q, const void* d_grid, uint32_t tti0, uint32_t nof_sf, const srslte_hip_dl_grant2_t* grants,
                                        uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* stream);
/* A pool of `depth` receive pipelines behind ONE submission call. A batch of 128 subframes is 832 decoder wavefronts for 1024 SIMDs and a chain of
 * dependent launches: one object on one stream reaches 0.4 of what the chip does with four batches in flight ("One call, one stream").
 * The pool owns the objects, a non-blocking stream and a completion event each, and takes the batches round-robin: the caller - one host thread, one
 * call per batch, as a worker of srsue / srsenb would make it - gets the overlapped rate without managing streams. submit() queues batch n on object
 * n % depth (waiting first, on the host, for the batch that used that object `depth` submissions ago) and returns a ticket; wait(ticket) blocks
 * until that batch's d_tb / d_tb_ok are final (their device buffers are the caller's; copy them on any stream after wait, or pass a pinned host
 * record to have the pool copy [nof_sf][tb_stride] bytes + [nof_sf] flags there on the batch's stream before the event). grants may be NULL (the
 * object's fixed grant). All calls from one host thread. */
typedef struct srslte_hip_dl_rx_pool srslte_hip_dl_rx_pool_t;
srslte_hip_dl_rx_pool_t* srslte_hip_dl_rx_pool_create(const srslte_hip_dl_rx_cfg_t* cfg, uint32_t depth);
void                     srslte_hip_dl_rx_pool_destroy(srslte_hip_dl_rx_pool_t* p);
int64_t srslte_hip_dl_rx_pool_submit(srslte_hip_dl_rx_pool_t* p, const void* d_iq, uint32_t tti0, uint32_t nof_sf, const srslte_hip_dl_grant_t* grants,
                                     uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* h_record /* pinned, or NULL */);
int     srslte_hip_dl_rx_pool_wait(srslte_hip_dl_rx_pool_t* p, int64_t ticket);

/* one stage of the chain (0 OFDM RX, 1 chest_dl, 2 extract+equalise+demap+descramble, 3 rate de-matching, 4 turbo decode, 5 TB CRC):
 * what srslte_hip_dl_rx_batch runs in order; exposed so that each kernel can be timed on its own */
int srslte_hip_dl_rx_stage(srslte_hip_dl_rx_t* q, int stage, const void* d_iq, uint32_t tti0, uint32_t nof_sf, uint8_t* d_tb,
                           uint32_t tb_stride, uint8_t* d_tb_ok, void* stream);
/* intermediate device buffers of the last call, for parity tests: 0 grid, 1 ce, 2 chest res, 3 d (NULL unless
 * srslte_hip_dl_rx_keep_symbols(q, 1): the equalised symbols are otherwise never written to memory), 4 e (LLRs, per-subframe stride
 * = max nof_re * Qm rounded up to 16; before the CSI weighting), 5 w, 6 cb iters, 7 cb ok, 8 cb bytes, 9 csi [nof_sf][max nof_re] (csi_enable),
 * 10 the subframes' largest csi [nof_sf] */
const void* srslte_hip_dl_rx_debug_buffer(const srslte_hip_dl_rx_t* q, int which);
int         srslte_hip_dl_rx_keep_symbols(srslte_hip_dl_rx_t* q, int enable);

/* ------------------------------------------------------------------ PUSCH receive pipeline (eNB side): OFDM RX with the
 * -1/2 carrier shift (enb_ul.c:58-63) -> chest_ul -> RE extraction + one-tap MMSE -> inverse transform precoding -> soft demap +
 * descramble + UL channel de-interleaver (pusch.c:423-520, sch.c:891-913,:991-1066) -> rate de-matching -> turbo decode -> TB CRC.
 * UL-SCH with optional HARQ-ACK / RI / CQI multiplexing (cfg fields below), one grant per object (optionally hopping between the slots), normal or extended CP (cfg.cp_ext),
 * 16-bit LLRs (pusch.llr_is_8bit has no counterpart here: with it the reference hands int8 LLRs to an int16 channel deinterleaver,
 * pusch.c:481-503 / sch.c:890-918, and fails its own noise-free round trip - tests/test_oracle_vs_ref.py); redundancy versions and soft
 * combining through srslte_hip_ul_rx_batch_harq. */
typedef struct srslte_hip_ul_rx srslte_hip_ul_rx_t;
typedef struct {
  uint32_t cell_id, nof_prb;
  uint16_t rnti;
  int      mod;            /* srslte_mod_t: QPSK, 16QAM, 64QAM */
  uint32_t tbs;            /* grant.tb.tbs. 0 with cqi_len > 0: a PUSCH WITHOUT UL-SCH data (36.212 5.2.4.1; srslte_ulsch_decode with cb_segm.tbs == 0,
                            * sch.c:943-975,:1031-1065): the CQI report fills what the rank indication leaves (uci.c:266-281), HARQ-ACK and RI are sized by
                            * the report (uci.c:557-564); the batch calls decode the UCI and set d_tb_ok to 0. In grants mode: per grant (tbs = 0 with cqi_len > 0),
                            * on an object made with the largest transport-block size */
  uint32_t L_prb, n_prb, n_dmrs; /* grant: srslte_pusch_grant_t.L_prb / n_prb_tilde / n_dmrs (pusch_cfg.h:47-60) */
  uint32_t max_iterations, max_batch;
  int      mmse;           /* 1: noise_estimate from chest_ul (pusch.c:475) */
  srslte_hip_dmrs_pusch_cfg_t dmrs_cfg;
  int      shortened;      /* 1: the subframe's last symbol is left to the SRS (srslte_ul_sf_cfg_t.shortened, N_srs = 1): 11 data symbols */
  uint32_t ack_len;        /* 0..2 HARQ-ACK bits multiplexed on the PUSCH (srslte_uci_cfg_t.ack[0].nof_acks, sch.c:1074-1090, uci.c:755-790) */
  uint32_t I_offset_ack;   /* beta_offset index of 36.213 Table 8.6.3-1 (srslte_uci_offset_cfg_t.I_offset_ack) */
  uint32_t ri_len;         /* 0..2 rank-indication bits on the PUSCH (srslte_cqi_cfg_t.ri_len, sch.c:968-979,:1110-1129): their symbols are
                            * left out by the channel interleaver and the UL-SCH is rate-matched to the rest */
  uint32_t I_offset_ri;    /* index into 36.213 Table 8.6.3-2 (srslte_uci_offset_cfg_t.I_offset_ri) */
  uint32_t cqi_len, I_offset_cqi; /* CQI / PMI report of cqi_len bits (srslte_cqi_size of uci_cfg.cqi, up to 64) multiplexed in front of the UL-SCH
                                     with srslte_uci_offset_cfg_t.I_offset_cqi (sch.c:1031-1060,:1133-1160, uci.c:264-494); 0 = none */
  uint32_t hopping, n_prb_slot1;  /* hopping != 0: intra-subframe hopping, slot 1 at PRB offset n_prb_slot1 (srslte_pusch_grant_t.n_prb[1] /
                                     n_prb_tilde[1]; pusch.c:52-91, chest_ul.c:244-266, refsignal_ul.c:316-346); 0: both slots at n_prb */
  uint32_t max_grants;            /* srslte_hip_ul_rx_batch_grants: PUSCHs (= HARQ slots) per call; 0 = max_batch */
  int      cp_ext;                /* 1: extended cyclic prefix (srslte_cell_t.cp): 6 symbols per slot, DMRS in symbol 2 of each slot (refsignal_ul.h:43,
                                     pusch.c:57-60), 10 data symbols (9 shortened; ra_ul.c:234), the UCI column sets of uci.c:502,:527, the DMRS
                                     cyclic-shift hopping read at a stride of 8 x 6 bits (refsignal_ul.c:127-133); d_iq is [nof_sf][15*N] as before */
} srslte_hip_ul_rx_cfg_t;
srslte_hip_ul_rx_t* srslte_hip_ul_rx_create(const srslte_hip_ul_rx_cfg_t* cfg);
void                srslte_hip_ul_rx_destroy(srslte_hip_ul_rx_t* q);
/* d_iq: [nof_sf][15*N]; d_tb [nof_sf][tb_stride] (tbs/8 + 3 CRC bytes used), d_tb_ok [nof_sf]; subframe b is TTI tti0 + b */
int srslte_hip_ul_rx_batch(srslte_hip_ul_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, uint8_t* d_tb, uint32_t tb_stride,
                           uint8_t* d_tb_ok, void* stream);
/* HARQ (srslte_ulsch_decode -> decode_tb with grant.tb.rv and cfg->softbuffers.rx, sch.c:1063, :299-414): slot b of the object keeps its code
 * blocks' soft buffers, CRC flags and bytes between calls. new_data != 0: new transport blocks (srslte_hip_ul_rx_batch is rv 0 / new data);
 * new_data == 0: a retransmission with redundancy version rv (0..3) is added to the kept buffers, blocks whose CRC already passed are neither
 * combined nor decoded again. UCI is decoded afresh on every call. */
int srslte_hip_ul_rx_batch_harq(srslte_hip_ul_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, uint32_t rv, int new_data, uint8_t* d_tb,
                                uint32_t tb_stride, uint8_t* d_tb_ok, void* stream);
/* Per-PUSCH grants (srslte_enb_ul_get_pusch once per scheduled UE after one srslte_enb_ul_fft per TTI, enb_ul.c:170-235): grants[p] names the
 * subframe of the batch it was received in (several PUSCHs may share one, on disjoint PRBs), its allocation (srslte_pusch_grant_t.L_prb,
 * n_prb_tilde[0 / 1]), n_dmrs, RNTI, modulation (1..3), transport block (<= cfg.tbs, no filler bits, one block size), redundancy version and
 * new-data flag. Slot p of the object is that PUSCH's srslte_softbuffer_rx_t between calls (HARQ as srslte_hip_ul_rx_batch_harq). Rows p of
 * d_tb / d_tb_ok. The object's cell, DMRS configuration, shortened flag, equaliser and pass limit apply (its own grant and UCI fields are those of
 * the fixed pipeline and play no part here); cfg.tbs = the largest transport block, cfg.max_grants >= nof_grants. HARQ-ACK and rank indication
 * per PUSCH (decisions: srslte_hip_ul_rx_grants_ack / _ri, [max_grants][2] device bytes each, row p; a call in which no grant carries either leaves
 * the rows as they were) and CQI reports (srslte_hip_ul_rx_grants_cqi:
 * [max_grants][64] bits, then [max_grants] CRC flags, as srslte_hip_ul_rx_cqi; rows whose grant carries no report keep their content). */
typedef struct {
  uint32_t sf;                       /* 0 .. nof_sf-1 */
  uint16_t rnti;
  uint32_t L_prb, n_prb, n_prb_slot1, n_dmrs;
  int      mod;
  uint32_t tbs, rv;
  int      new_data;
  uint32_t ack_len, I_offset_ack; /* 0..2 HARQ-ACK bits on this PUSCH and their offset index (as the cfg fields of the fixed pipeline) */
  uint32_t ri_len, I_offset_ri;   /* 0..2 rank-indication bits */
  uint32_t cqi_len, I_offset_cqi; /* 0..64 bits of CQI / PMI report (srslte_cqi_size) */
} srslte_hip_ul_grant_t;
int srslte_hip_ul_rx_batch_grants(srslte_hip_ul_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, const srslte_hip_ul_grant_t* grants,
                                  uint32_t nof_grants, uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* stream);
const uint8_t* srslte_hip_ul_rx_grants_ack(const srslte_hip_ul_rx_t* q);
const uint8_t* srslte_hip_ul_rx_grants_ri(const srslte_hip_ul_rx_t* q);
const uint8_t* srslte_hip_ul_rx_grants_cqi(const srslte_hip_ul_rx_t* q);
/* Device pointer to the HARQ-ACK decisions of the last batch on this object, [max_batch][2] bytes (srslte_uci_value_t.ack.ack_value of
 * srslte_pusch_decode); valid once the batch's stream work is done, all zero when cfg.ack_len == 0 */
const uint8_t* srslte_hip_ul_rx_ack(const srslte_hip_ul_rx_t* q);
const uint8_t* srslte_hip_ul_rx_ri(const srslte_hip_ul_rx_t* q); /* the same for the rank indication (srslte_uci_value_t.ri in [b][0]) */
/* the CQI reports of the last batch: [max_batch][64] bits (one per byte, srslte_cqi_value_pack order), then [max_batch] flags
 * (srslte_uci_value_t.cqi.data_crc: always 1 up to 11 bits, the CRC-8 result above; bits are only written when the flag is 1) */
const uint8_t* srslte_hip_ul_rx_cqi(const srslte_hip_ul_rx_t* q);
/* intermediate device buffers of the last call, for parity tests: 0 grid, 1 ce, 2 chest_ul res, 3 d (after de-precoding), 4 g (LLRs after
 * the de-interleaver), 5 w, 6 cb iters, 7 cb ok, 8 cb bytes, 9 z (equalised) */
const void* srslte_hip_ul_rx_debug_buffer(const srslte_hip_ul_rx_t* q, int which);

/* ------------------------------------------------------------------ PUSCH transmit pipeline (UE side): TB CRC24A +
 * segmentation + CB CRC24B -> turbo encoder -> rate matching + UL channel interleaver + scrambling + modulation -> transform precoding
 * -> RE mapping with the DMRS -> OFDM TX with 1/sqrt(N) and the +1/2 carrier shift (srslte_ue_ul_encode ue_ul.c:300-340 with
 * srslte_pusch_encode pusch.c:314-421, the UL-SCH part of srslte_ulsch_encode sch.c:1068-1160, srslte_refsignal_dmrs_pusch_put and
 * srslte_ofdm_tx_sf). Same restrictions as the receive pipeline. */
typedef struct srslte_hip_ul_tx srslte_hip_ul_tx_t;
typedef struct {
  uint32_t cell_id, nof_prb;
  uint16_t rnti;
  int      mod;            /* srslte_mod_t: QPSK, 16QAM, 64QAM */
  uint32_t tbs;            /* 0 with cqi_len > 0: a PUSCH without UL-SCH data (srslte_ulsch_encode with cb_segm.tbs == 0, sch.c:1111-1114,:1157-1174):
                            * d_tb of the batch calls is not read (may be NULL). In grants mode: per grant, on an object made with tbs > 0 */
  uint32_t L_prb, n_prb, n_dmrs;
  uint32_t max_batch;
  srslte_hip_dmrs_pusch_cfg_t dmrs_cfg;
  int      shortened;      /* as in srslte_hip_ul_rx_cfg_t */
  uint32_t ack_len, I_offset_ack; /* as in srslte_hip_ul_rx_cfg_t (srslte_ulsch_encode's uci_cfg, sch.c:1168-1215) */
  uint32_t ri_len, I_offset_ri;   /* as in srslte_hip_ul_rx_cfg_t (sch.c:1110-1129) */
  uint32_t cqi_len, I_offset_cqi; /* as in srslte_hip_ul_rx_cfg_t (srslte_uci_encode_cqi_pusch, sch.c:1133-1150) */
  uint32_t hopping, n_prb_slot1;  /* as in srslte_hip_ul_rx_cfg_t */
  uint32_t max_grants;            /* srslte_hip_ul_tx_batch_grants: PUSCHs per call; 0 = max_batch */
  int      cp_ext;                /* as in srslte_hip_ul_rx_cfg_t */
} srslte_hip_ul_tx_cfg_t;
srslte_hip_ul_tx_t* srslte_hip_ul_tx_create(const srslte_hip_ul_tx_cfg_t* cfg);
void                srslte_hip_ul_tx_destroy(srslte_hip_ul_tx_t* q);
/* d_tb: [nof_sf][tb_stride] payload bytes (tbs/8 used); d_iq: [nof_sf][15*N] cf32 time samples; subframe b is TTI tti0 + b */
int srslte_hip_ul_tx_batch(srslte_hip_ul_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, uint32_t tti0, uint32_t nof_sf, void* d_iq,
                           void* stream);
/* The same with the HARQ-ACK values d_ack [nof_sf][2] (0/1 bytes, device) multiplexed in; requires cfg.ack_len > 0 */
int srslte_hip_ul_tx_batch_ack(srslte_hip_ul_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, const uint8_t* d_ack, uint32_t tti0,
                               uint32_t nof_sf, void* d_iq, void* stream);
/* HARQ-ACK values d_ack and rank-indication bits d_ri, each [nof_sf][2] device bytes, each required exactly when configured */
int srslte_hip_ul_tx_batch_uci(srslte_hip_ul_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, const uint8_t* d_ack, const uint8_t* d_ri,
                               uint32_t tti0, uint32_t nof_sf, void* d_iq, void* stream);
/* ... and the CQI / PMI report d_cqi [nof_sf][64] device bytes (one bit each, the first cfg.cqi_len used: srslte_cqi_value_pack's output),
 * required exactly when cfg.cqi_len > 0 (srslte_uci_encode_cqi_pusch, sch.c:1133-1150) */
int srslte_hip_ul_tx_batch_uci_cqi(srslte_hip_ul_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, const uint8_t* d_ack, const uint8_t* d_ri,
                                   const uint8_t* d_cqi, uint32_t tti0, uint32_t nof_sf, void* d_iq, void* stream);
/* ... with a redundancy version rv (0..3; srslte_pusch_grant_t.tb.rv -> srslte_ulsch_encode -> srslte_rm_turbo_tx_lut, sch.c:1157-1160): what a
 * HARQ retransmission sends. d_ack / d_ri / d_cqi as above (NULL when not configured). */
int srslte_hip_ul_tx_batch_rv(srslte_hip_ul_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, const uint8_t* d_ack, const uint8_t* d_ri,
                              const uint8_t* d_cqi, uint32_t rv, uint32_t tti0, uint32_t nof_sf, void* d_iq, void* stream);
/* Per-PUSCH grants on the transmit side: grants[p] as srslte_hip_ul_rx_batch_grants takes them (new_data unused) - what srslte_ue_ul_encode sends
 * TTI after TTI as the grants come in (ue_ul.c:300-340); several PUSCHs on disjoint PRBs of one subframe give the composite signal of several
 * UEs. Row p of d_tb is its transport block; d_ack / d_ri [nof_grants][2] and d_cqi [nof_grants][64] device bytes, rows p (NULL when no grant of
 * the call carries that UCI). cfg.tbs bounds every grant's tbs, cfg.max_grants the PUSCHs per call. */
int srslte_hip_ul_tx_batch_grants(srslte_hip_ul_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, const uint8_t* d_ack, const uint8_t* d_ri,
                                  const uint8_t* d_cqi, uint32_t tti0, uint32_t nof_sf, const srslte_hip_ul_grant_t* grants, uint32_t nof_grants,
                                  void* d_iq, void* stream);
/* intermediate device buffers of the last call, for parity tests: 0 code blocks (stride (K/8+15)&~15), 1 parity streams (stride
 * (K/4+1+15)&~15), 2 d (modulated), 3 z (after transform precoding), 4 grid, 5 TB CRCs (one word per subframe) */
const void* srslte_hip_ul_tx_debug_buffer(const srslte_hip_ul_tx_t* q, int which);

/* ------------------------------------------------------------------ PDSCH transmit pipeline (eNB side): srslte_pdsch_encode
 * (pdsch.c:1059-1185: DL-SCH coding, scrambling, modulation, layer mapping + SFBC precoding, RE mapping) + CRS
 * (srslte_refsignal_cs_put_sf refsignal_dl.c:253-272) + srslte_ofdm_tx_sf with 1/sqrt(N) (enb_dl.c:56-62). One codeword, TM1 or 2-port
 * TM2, full-band grant; no control region, PSS/SSS or PBCH content (their REs stay zero). */
typedef struct srslte_hip_dl_tx srslte_hip_dl_tx_t;
typedef struct {
  uint32_t cell_id, nof_prb, cfi;
  uint16_t rnti;
  int      mod;            /* srslte_mod_t: QPSK .. 256QAM */
  uint32_t tbs;
  uint32_t max_batch;
  uint32_t nof_ports;      /* 0 or 1: TM1; 2 or 4: transmit diversity */
  float    p_a;            /* dB; rho_a = 10^(p_a/20) (x sqrt(2) for 2 ports), pdsch.c:518-554 with p_b giving rho_b = 1 */
  uint32_t max_grants;     /* srslte_hip_dl_tx_batch_grants: PDSCHs per call; 0 = max_batch */
  int      cp_ext;         /* 1: extended-CP cell (as srslte_hip_dl_rx_cfg_t.cp_ext): grids [12][12 * nof_prb], CRS on symbols 0 and 3 of each slot
                              with the extended-CP sequences (N_CP = 0 in c_init, refsignal_dl.c:79-99), srslte_ofdm_tx_sf with the long prefix */
  int      tdd;            /* as srslte_hip_dl_rx_cfg_t.tdd: TDD cell, per-PDSCH grants only (srslte_hip_dl_tx_batch_grants); special subframes get the
                              CRS symbols of their DwPTS (srslte_refsignal_cs_put_sf, refsignal_dl.c:253-272) and PDSCH there */
  uint32_t tdd_sf_config, tdd_ss_config;
  int      mbsfn;          /* as srslte_hip_dl_rx_cfg_t.mbsfn: srslte_pmch_encode (pmch.c:423-483) + srslte_refsignal_mbsfn_put_sf (refsignal_dl.c:297-326:
                              the CRS of symbol 0 and the MBSFN reference signal in symbols 2, 6, 10) + srslte_ofdm_tx_sf on an MBSFN object
                              (ofdm.c:558-574); single port, rv 0, fixed-grant calls only; rnti is not used */
  uint32_t mbsfn_area_id, non_mbsfn_region;
} srslte_hip_dl_tx_cfg_t;
srslte_hip_dl_tx_t* srslte_hip_dl_tx_create(const srslte_hip_dl_tx_cfg_t* cfg);
void                srslte_hip_dl_tx_destroy(srslte_hip_dl_tx_t* q);
/* d_tb: [nof_sf][tb_stride] payload bytes; d_iq: [nof_sf][nof_ports][15*N] cf32, one time-domain signal per antenna port; rv: the
 * redundancy version of the whole batch (the circular buffer is re-encoded, not kept) */
int srslte_hip_dl_tx_batch(srslte_hip_dl_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, uint32_t tti0, uint32_t nof_sf, uint32_t rv, void* d_iq,
                           void* stream);
/* Per-PDSCH grants (srslte_enb_dl_put_base once per TTI, srslte_enb_dl_put_pdsch once per scheduled UE, srslte_enb_dl_gen_signal; enb_dl.c:330-419):
 * grants[p] = the subframe of the batch and a grant as the receive side takes it (PRB masks of both slots, modulation, transport block <= cfg.tbs,
 * redundancy version, RNTI, CFI; new_data unused); row p of d_tb is its transport block. Several PDSCHs may share a subframe; its grids carry the
 * CRS of every port and nothing else besides the PDSCHs (no control region, PSS / SSS / PBCH). d_iq as srslte_hip_dl_tx_batch. */
typedef struct {
  uint32_t              sf; /* 0 .. nof_sf-1 */
  srslte_hip_dl_grant_t grant;
} srslte_hip_dl_tx_grant_t;
int srslte_hip_dl_tx_batch_grants(srslte_hip_dl_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, uint32_t tti0, uint32_t nof_sf,
                                  const srslte_hip_dl_tx_grant_t* grants, uint32_t nof_grants, void* d_iq, void* stream);
/* intermediate device buffers of the last call: 0 code blocks, 1 parity streams, 2 per-port symbol streams [nof_sf][nof_ports][max nof_re],
 * 3 grids [nof_sf][nof_ports][14][12*nof_prb] */
const void* srslte_hip_dl_tx_debug_buffer(const srslte_hip_dl_tx_t* q, int which);

/* ---- one transport block with HOST buffers in one device call: decode_tb_cb behind srslte_dlsch_decode2 / srslte_ulsch_decode
 * (lib/src/phy/phch/sch.c:299-414,:507-531): rate de-matching of every code block into its soft buffer, turbo decoding with CRC early stop,
 * decoded bytes. The single-call srslte_tdec_* API costs a host round trip per code block and per SISO pass; this entry costs one per
 * transport block. include/srslte_hip/srslte_compat.h exports srslte_dlsch_decode2 on top of it. ---- */
typedef struct srslte_hip_sch srslte_hip_sch_t;
srslte_hip_sch_t* srslte_hip_sch_create(uint32_t max_tbs, uint32_t max_e_bits, int llr_8bit);
void              srslte_hip_sch_destroy(srslte_hip_sch_t* q);
/* e_bits (host): nof_e_bits LLRs, int16 (int8 for an 8-bit object); mod 1 QPSK .. 4 256QAM; Nl 1 or 2 (sch.c:510-514); buffer_f[c] / cb_crc[c]:
 * the C soft buffers and CRC flags of the srslte_softbuffer_rx_t (in / out, host); cb_bytes [C][768]: K / 8 decoded bytes of every block this
 * call decoded; sum_passes: SISO passes over those blocks */
int srslte_hip_sch_decode(srslte_hip_sch_t* q, const void* e_bits, uint32_t nof_e_bits, uint32_t tbs, int mod, uint32_t Nl, uint32_t rv,
                          uint32_t max_iterations, int16_t** buffer_f, uint8_t* cb_crc, uint8_t* cb_bytes, uint32_t* sum_passes);

#ifdef __cplusplus
}
#endif
#endif
