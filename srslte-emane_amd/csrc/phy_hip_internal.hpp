// Internal (non-ABI) declarations shared between the translation units of libsrslte_phy_hip.so.
#pragma once
#include "common.hpp"
#include "srslte_hip/phy_hip.h"
#include <vector>

struct FftFactors {
  int N, nf;
  int radix[8];
  int inplace; // LDS->LDS passes fit one butterfly per thread: single LDS buffer
};

// Returns (creating on first use, per device) the factorisation and the device twiddle table exp(-j2*pi*k/N).
int fft_get_plan(int N, FftFactors* f, const cf32** d_tw);


// Gold sequence c(n) of 36.211 7.2 (sequence.c:48-79), host side, for init-time tables.
void lte_gold_sequence(uint32_t c_init, uint32_t len, std::vector<uint8_t>& c);

// demod.hip: type 0 float / 1 int16 / 2 int8; optional packed scrambling bits [10][scr_words] selected by (tti0+call)%10
int demod_launch(int type, int mod, const void* d_sym, void* d_llr, int nsym, int ncalls, const uint32_t* d_scr, int scr_words, int tti0,
                 hipStream_t st);

// fec_tables.cpp: 36.212 tables shared by encoder, decoder and rate matching (host)
struct QppRow { uint16_t K, f1, f2; };
extern const QppRow lte_qpp_table[188];
int  lte_cb_index(uint32_t K);
void lte_qpp_tables(uint32_t K, uint32_t W, std::vector<uint16_t>& fwd, std::vector<uint16_t>& rev);
void lte_rm_rx_table(uint32_t K, uint32_t rv, std::vector<uint32_t>& d_index); // circular-buffer order -> 3*i+s

// chest.hip: srslte_chest_ul_estimate_pusch for a list of PUSCHs of one (L_prb, n_dmrs): item i = {subframe of the batch, PRB offset of slot 0 / 1,
// row of d_res}; d_items on the device
struct ChestUlItem { int sf, n_prb, n_prb1, row; };
int chest_ul_estimate_items(srslte_hip_chest_ul_t* q, uint32_t tti0, uint32_t L_prb, uint32_t n_dmrs, const ChestUlItem* d_items, int n_items,
                            const void* d_grid, void* d_ce, void* d_res, hipStream_t st);
int chest_ul_dmrs_table_cached(srslte_hip_chest_ul_t* q, uint32_t L_prb, uint32_t n_dmrs, const void** d_r); // one table per (L_prb, n_dmrs), all kept
// chest.hip: device DMRS table of a PUSCH grant, [10][2][12 * L_prb] cf32 (owned by q)
int chest_ul_dmrs_table(srslte_hip_chest_ul_t* q, uint32_t L_prb, uint32_t n_dmrs, const void** d_r);
// chest.hip: chest_common.c's stand-alone array helpers on device buffers (filter_len <= nof_ref, nof_ref >= 2 as the extrapolation reads in[0..1])
int chest_average_pilots_launch(const void* d_in, void* d_out, const float* d_filt, int nof_ref, int nof_symbols, int filter_len, hipStream_t st);
int chest_noise_pilots_launch(const void* d_noisy, const void* d_noiseless, void* d_noise_vec, int n, float* d_power, hipStream_t st);
// chest.hip: the noise estimates [port][antenna] the PSS / EMPTY algorithms keep between calls (q->noise_estimate of the reference)
int chest_dl_set_noise_state(srslte_hip_chest_dl_t* q, const float* noise);
// chest.hip: srslte_hip_chest_dl_estimate_batch_multi with the estimates kept as ONE row per (subframe, port, antenna) (ce_compact; only
// without interpolate_subframe, where every symbol of the subframe gets the same row)
int chest_dl_estimate_batch_rows(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0, const void* d_grid, void* d_ce,
                                 void* d_res, int nof_sf, int nof_rx, int ce_compact, void* stream);
// chest.hip: the MBSFN estimate on grids of 2 nsl symbols per subframe with a result record (noise figure) per subframe, for the PMCH pipeline
int chest_dl_estimate_mbsfn_rows(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0, const void* d_grid, void* d_ce,
                                 int nof_sf, int nof_rx, int nsl, void* d_res, void* stream);
// tdec.hip: let the windowed decoders also emit each block's share of the transport-block CRC syndrome (nullptr: off).
// d_rem: [C][K] words, x^(tbs+24-1-position in the TB) mod g for the block's payload bits in the decoder's array order, 0 elsewhere
void tdec_set_tb_syndrome(srslte_hip_tdec_t* q, const uint32_t* d_rem, uint32_t C, uint32_t* d_syn);
// tdec.hip: the NEXT run (16-window 16-bit decoder with tdec_set_tb_syndrome, no skip flags, no block map) writes every block's payload bytes
// straight into its transport block d_tb[cb / C][...] and the last block of a transport block to finish writes d_tb_ok[cb / C] (all block CRCs,
// the XOR of the TB-CRC shares, a non-zero parity: sch.c:470-488): no assembly kernel behind the decoder. d_tb = nullptr: off
int tdec_set_tb_direct(srslte_hip_tdec_t* q, uint8_t* d_tb, uint32_t tb_stride, uint32_t payload_bytes_per_block, uint8_t* d_tb_ok);
// ... for the next tdec_run_groups (a ragged batch of 16-bit blocks, none skipped): transport-block slot v = block slot / width has d_Cof[v] blocks and
// row v (v < B) or rows0 + v - B of d_tb / d_tb_ok
int tdec_set_tb_ragged(srslte_hip_tdec_t* q, uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, const uint8_t* d_Cof, uint32_t width, uint32_t B,
                       uint32_t rows0);
// tdec.hip: blocks with d_skip[cb] != 0 are left alone by the following runs: bytes, CRC flag, TB-CRC share stay (nullptr: off)
void tdec_set_skip(srslte_hip_tdec_t* q, const uint8_t* d_skip);
// tdec.hip: the following runs work on the block slots d_map[0 .. nof_cb) instead of 0 .. nof_cb-1 (input, output, iteration count, CRC flag,
// skip flag; nullptr: off). For ragged batches, where the code blocks of one length are scattered over the batch's slots
void tdec_set_cb_map(srslte_hip_tdec_t* q, const uint32_t* d_map);
// tdec.hip: the NEXT run continues blocks whose passes 0..start_iter-1 the previous run on this object did (same inputs, same block
// slots): srslte_tdec_iteration's one-more-pass without redoing the earlier ones (turbodecoder.c:539-545)
void tdec_set_resume(srslte_hip_tdec_t* q, uint32_t start_iter);
// tdec.hip: a ragged batch in one call - groups of equal block length, in the order of the block map set with tdec_set_cb_map - with ONE launch
// per decoder kernel the lengths need instead of one per length (back-ends chosen per length as on an AVX2 host; CRC per group for the early stop)
struct srslte_hip_tdec_group_t {
  uint32_t K, nof_cb, crc_poly, crc_nbits;
};
int tdec_run_groups(srslte_hip_tdec_t* q, const void* d_input, int llr8, uint32_t in_stride, const srslte_hip_tdec_group_t* groups, uint32_t nof_groups,
                    uint32_t nof_iterations, uint8_t* d_output, uint32_t out_stride, uint32_t* d_iters, uint8_t* d_crc_ok, hipStream_t st);
// tdec.hip: srslte_hip_tdec_run_batch with an optional forced back-end (force_w = -1 auto, 0 generic, 8, 16, 32 with llr8);
// llr8: d_input is int8 and the 8-bit numerics / fall-backs of turbodecoder.c:438-487 apply
int tdec_run_batch_w(srslte_hip_tdec_t* q, const void* d_input, int llr8, uint32_t in_stride, int sb_layout, uint32_t K, int force_w,
                     uint32_t nof_cb, uint32_t nof_iterations, uint32_t crc_poly, uint32_t crc_nbits, uint8_t* d_output,
                     uint32_t out_stride, uint32_t* d_iters, uint8_t* d_crc_ok, hipStream_t st);
