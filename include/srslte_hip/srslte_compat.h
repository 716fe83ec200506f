/*
 * include/srslte_hip/srslte_compat.h — the reference's own single-call C API for the hot path, served by
 * libsrslte_phy_hip.so. Same symbol names, signatures, error codes and PUBLIC struct layouts as the headers under
 * /root/reference/lib/include/srslte/phy/ that each block cites, so that callers such as lib/src/phy/ue/ue_dl.c,
 * enb/enb_dl.c, phch/sch.c or the library tests link against this library instead of dft_fftw.c/ofdm.c/
 * dft_precoding.c/turbocoder.c/turbodecoder*.c/cbsegm.c/tc_interl_lte.c/chest_dl.c/demod_soft.c.
 *
 * Semantics: host pointers in, host pointers out, synchronous (copy in -> HIP kernels -> copy out). This path exists
 * for compatibility and for parity testing at the reference's own granularity; throughput comes from the batched
 * API in phy_hip.h. Device state hangs off the opaque slots the reference structs already have
 * (srslte_dft_plan_t.p, srslte_tdec_t.dec16_hdlr[0], srslte_chest_dl_t.tmp_noise ...).
 * tests/test_abi_layout.py checks sizeof/offsetof of every struct below against the reference headers.
 *
 * Not provided (documented in DESIGN.md): interpolate_subframe on a 4-port cell or switched off in an MBSFN subframe; those calls
 * return SRSLTE_ERROR with a message.
 */
#ifndef SRSLTE_HIP_SRSLTE_COMPAT_H
#define SRSLTE_HIP_SRSLTE_COMPAT_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
typedef struct { float re, im; } cf_t; /* storage-compatible with C99 float _Complex */
#else
#include <complex.h>
typedef _Complex float cf_t;           /* config.h:68 */
#endif

#define SRSLTE_SUCCESS 0
#define SRSLTE_ERROR -1
#define SRSLTE_ERROR_INVALID_INPUTS -2 /* config.h:58-66 */

#define SRSLTE_MAX_PORTS 4             /* phy_common.h:49 */
#define SRSLTE_MAX_PRB 110             /* phy_common.h:88 */
#define SRSLTE_NOF_SF_X_FRAME 10       /* phy_common.h:40 */
#define SRSLTE_NOF_TC_CB_SIZES 188     /* cbsegm.h:30 */
#define SRSLTE_PSS_LEN 62              /* sync/pss.h:53 */
#define SRSLTE_TCOD_MAX_LEN_CB 6144    /* turbodecoder.h:44 */
#define SRSLTE_TDEC_NOF_AUTO_MODES_8 2
#define SRSLTE_TDEC_NOF_AUTO_MODES_16 3

/* ------------------------------------------------------------------ common types (phy_common.h:160-212,241-247) */
typedef enum { SRSLTE_CP_NORM = 0, SRSLTE_CP_EXT } srslte_cp_t;
typedef enum { SRSLTE_SF_NORM = 0, SRSLTE_SF_MBSFN } srslte_sf_t;
typedef enum { SRSLTE_PHICH_NORM = 0, SRSLTE_PHICH_EXT } srslte_phich_length_t;
typedef enum { SRSLTE_PHICH_R_1_6 = 0, SRSLTE_PHICH_R_1_2, SRSLTE_PHICH_R_1, SRSLTE_PHICH_R_2 } srslte_phich_r_t;
typedef enum { SRSLTE_FDD = 0, SRSLTE_TDD = 1 } srslte_frame_type_t;
typedef enum { SRSLTE_MOD_BPSK = 0, SRSLTE_MOD_QPSK, SRSLTE_MOD_16QAM, SRSLTE_MOD_64QAM, SRSLTE_MOD_256QAM } srslte_mod_t;
typedef struct { uint32_t sf_config; uint32_t ss_config; bool configured; } srslte_tdd_config_t;
typedef struct {
  uint32_t nof_prb; uint32_t nof_ports; uint32_t id; srslte_cp_t cp; srslte_phich_length_t phich_length;
  srslte_phich_r_t phich_resources; srslte_frame_type_t frame_type;
} srslte_cell_t;
typedef struct { srslte_tdd_config_t tdd_config; uint32_t tti; uint32_t cfi; srslte_sf_t sf_type; uint32_t non_mbsfn_region; } srslte_dl_sf_cfg_t;

int  srslte_symbol_sz(uint32_t nof_prb); /* phy_common.c:322-345 */
int  srslte_symbol_sz_power2(uint32_t nof_prb); /* phy_common.c:304-320 */
void srslte_use_standard_symbol_size(bool enabled); /* phy_common.c:297-299: srslte_symbol_sz then returns the power-of-two family */

/* ------------------------------------------------------------------ DFT (dft.h:46-152, dft_fftw.c) */
typedef enum { SRSLTE_DFT_COMPLEX, SRSLTE_REAL } srslte_dft_mode_t;
typedef enum { SRSLTE_DFT_FORWARD, SRSLTE_DFT_BACKWARD } srslte_dft_dir_t;
typedef struct {
  int init_size; int size; void* in; void* out; void* p; bool is_guru; bool forward; bool mirror; bool db; bool norm; bool dc;
  srslte_dft_dir_t dir; srslte_dft_mode_t mode;
} srslte_dft_plan_t;
void srslte_dft_load(void);
void srslte_dft_exit(void);
int  srslte_dft_plan(srslte_dft_plan_t* plan, int dft_points, srslte_dft_dir_t dir, srslte_dft_mode_t type);
int  srslte_dft_plan_r(srslte_dft_plan_t* plan, int dft_points, srslte_dft_dir_t dir); /* FFTW R2HC / HC2R half-complex layout */
int  srslte_dft_replan(srslte_dft_plan_t* plan, int new_dft_points);
int  srslte_dft_replan_r(srslte_dft_plan_t* plan, int new_dft_points);
void srslte_dft_run_r(srslte_dft_plan_t* plan, const float* in, float* out);
int  srslte_dft_plan_c(srslte_dft_plan_t* plan, int dft_points, srslte_dft_dir_t dir);
int  srslte_dft_plan_guru_c(srslte_dft_plan_t* plan, int dft_points, srslte_dft_dir_t dir, cf_t* in_buffer, cf_t* out_buffer, int istride,
                            int ostride, int how_many, int idist, int odist);
int  srslte_dft_replan_c(srslte_dft_plan_t* plan, int new_dft_points);
int  srslte_dft_replan_guru_c(srslte_dft_plan_t* plan, int new_dft_points, cf_t* in_buffer, cf_t* out_buffer, int istride, int ostride,
                              int how_many, int idist, int odist);
void srslte_dft_plan_set_mirror(srslte_dft_plan_t* plan, bool val);
void srslte_dft_plan_set_db(srslte_dft_plan_t* plan, bool val);
void srslte_dft_plan_set_norm(srslte_dft_plan_t* plan, bool val);
void srslte_dft_plan_set_dc(srslte_dft_plan_t* plan, bool val);
void srslte_dft_plan_free(srslte_dft_plan_t* plan);
void srslte_dft_run(srslte_dft_plan_t* plan, const void* in, void* out);
void srslte_dft_run_c(srslte_dft_plan_t* plan, const cf_t* in, cf_t* out);
void srslte_dft_run_c_zerocopy(srslte_dft_plan_t* plan, const cf_t* in, cf_t* out);
void srslte_dft_run_guru_c(srslte_dft_plan_t* plan);

/* ------------------------------------------------------------------ OFDM (ofdm.h:42-153, ofdm.c) */
typedef struct {
  srslte_dft_plan_t fft_plan; srslte_dft_plan_t fft_plan_sf[2];
  uint32_t max_prb; uint32_t nof_symbols; uint32_t symbol_sz; uint32_t nof_guards; uint32_t nof_re; uint32_t slot_sz; uint32_t sf_sz;
  srslte_cp_t cp; cf_t* tmp; cf_t* in_buffer; cf_t* out_buffer;
  bool mbsfn_subframe; uint32_t mbsfn_guard_len; uint32_t nof_symbols_mbsfn; uint8_t non_mbsfn_region;
  bool freq_shift; float freq_shift_f; cf_t* shift_buffer;
} srslte_ofdm_t;
int  srslte_ofdm_rx_init(srslte_ofdm_t* q, srslte_cp_t cp_type, cf_t* in_buffer, cf_t* out_buffer, uint32_t max_prb);
int  srslte_ofdm_tx_init(srslte_ofdm_t* q, srslte_cp_t cp_type, cf_t* in_buffer, cf_t* out_buffer, uint32_t nof_prb);
int  srslte_ofdm_rx_set_prb(srslte_ofdm_t* q, srslte_cp_t cp, uint32_t nof_prb);
int  srslte_ofdm_tx_set_prb(srslte_ofdm_t* q, srslte_cp_t cp, uint32_t nof_prb);
void srslte_ofdm_rx_free(srslte_ofdm_t* q);
void srslte_ofdm_tx_free(srslte_ofdm_t* q);
void srslte_ofdm_rx_slot(srslte_ofdm_t* q, int slot_in_sf);
void srslte_ofdm_tx_slot(srslte_ofdm_t* q, int slot_in_sf);
void srslte_ofdm_rx_sf(srslte_ofdm_t* q);
void srslte_ofdm_tx_sf(srslte_ofdm_t* q);
void srslte_ofdm_rx_sf_ng(srslte_ofdm_t* q, cf_t* input, cf_t* output);
void srslte_ofdm_rx_slot_ng(srslte_ofdm_t* q, cf_t* input, cf_t* output);
int  srslte_ofdm_init_(srslte_ofdm_t* q, srslte_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, int symbol_sz, int nof_prb, srslte_dft_dir_t dir);
int  srslte_ofdm_init_mbsfn_(srslte_ofdm_t* q, srslte_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, int symbol_sz, int nof_prb, srslte_dft_dir_t dir,
                             srslte_sf_t sf_type);
int  srslte_ofdm_rx_init_mbsfn(srslte_ofdm_t* q, srslte_cp_t cp_type, cf_t* in_buffer, cf_t* out_buffer, uint32_t max_prb);
int  srslte_ofdm_tx_init_mbsfn(srslte_ofdm_t* q, srslte_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, uint32_t nof_prb);
void srslte_ofdm_set_non_mbsfn_region(srslte_ofdm_t* q, uint8_t non_mbsfn_region);
void srslte_ofdm_rx_slot_mbsfn(srslte_ofdm_t* q, cf_t* input, cf_t* output); /* extended-CP objects, region 1 or 2, no frequency shift */
void srslte_ofdm_tx_slot_mbsfn(srslte_ofdm_t* q, cf_t* input, cf_t* output);
int  srslte_ofdm_set_freq_shift(srslte_ofdm_t* q, float freq_shift);
void srslte_ofdm_set_normalize(srslte_ofdm_t* q, bool normalize_enable);

/* ------------------------------------------------------------------ SC-FDMA transform precoding (dft_precoding.h:39-64) */
typedef struct { uint32_t max_prb; srslte_dft_plan_t dft_plan[SRSLTE_MAX_PRB + 1]; } srslte_dft_precoding_t;
int  srslte_dft_precoding_init(srslte_dft_precoding_t* q, uint32_t max_prb, bool is_tx);
int  srslte_dft_precoding_init_tx(srslte_dft_precoding_t* q, uint32_t max_prb);
int  srslte_dft_precoding_init_rx(srslte_dft_precoding_t* q, uint32_t max_prb);
void srslte_dft_precoding_free(srslte_dft_precoding_t* q);
bool srslte_dft_precoding_valid_prb(uint32_t nof_prb);
int  srslte_dft_precoding(srslte_dft_precoding_t* q, cf_t* input, cf_t* output, uint32_t nof_prb, uint32_t nof_symbols);

/* ------------------------------------------------------------------ segmentation + interleaver (cbsegm.h:33-52, tc_interl.h:36-52) */
typedef struct { uint32_t F, C, K1, K2, K1_idx, K2_idx, C1, C2, tbs; } srslte_cbsegm_t;
int  srslte_cbsegm(srslte_cbsegm_t* s, uint32_t tbs);
int  srslte_cbsegm_cbsize(uint32_t index);
bool srslte_cbsegm_cbsize_isvalid(uint32_t size);
int  srslte_cbsegm_cbindex(uint32_t long_cb);
typedef struct { uint16_t* forward; uint16_t* reverse; uint32_t max_long_cb; } srslte_tc_interl_t;
int  srslte_tc_interl_init(srslte_tc_interl_t* h, uint32_t max_long_cb);
void srslte_tc_interl_free(srslte_tc_interl_t* h);
int  srslte_tc_interl_LTE_gen(srslte_tc_interl_t* h, uint32_t long_cb);
int  srslte_tc_interl_LTE_gen_interl(srslte_tc_interl_t* h, uint32_t long_cb, uint32_t interl_win);
int  srslte_tc_interl_UMTS_gen(srslte_tc_interl_t* h, uint32_t long_cb); /* tc_interl_umts.c:80-262 (25.212; no caller on the LTE path) */

/* ------------------------------------------------------------------ turbo encoder (turbocoder.h:44-76) */
typedef struct { uint32_t max_long_cb; uint8_t* temp; } srslte_tcod_t;
int  srslte_tcod_init(srslte_tcod_t* h, uint32_t max_long_cb);
void srslte_tcod_free(srslte_tcod_t* h);
int  srslte_tcod_encode(srslte_tcod_t* h, uint8_t* input, uint8_t* output, uint32_t long_cb);
typedef struct { /* srslte_crc_t, fec/crc.h:38-46 (crc.c itself stays in the reference library) */
  uint64_t table[256]; int polynom; int order; uint64_t crcinit; uint64_t crcmask; uint64_t crchighbit; uint32_t srslte_crc_out;
} srslte_crc_t;
int  srslte_tcod_encode_lut(srslte_tcod_t* h, srslte_crc_t* crc_tb, srslte_crc_t* crc_cb, uint8_t* input, uint8_t* parity, uint32_t cblen_idx,
                            bool last_cb);
void srslte_tcod_gentable(void);

/* ------------------------------------------------------------------ turbo decoder (turbodecoder.h:63-135) */
typedef enum { SRSLTE_TDEC_8, SRSLTE_TDEC_16 } srslte_tdec_llr_type_t;
typedef enum {
  SRSLTE_TDEC_AUTO = 0, SRSLTE_TDEC_GENERIC, SRSLTE_TDEC_SSE, SRSLTE_TDEC_SSE_WINDOW, SRSLTE_TDEC_NEON_WINDOW, SRSLTE_TDEC_AVX_WINDOW,
  SRSLTE_TDEC_SSE8_WINDOW, SRSLTE_TDEC_AVX8_WINDOW, SRSLTE_TDEC_NOF_IMP
} srslte_tdec_impl_type_t;
typedef struct {
  uint32_t max_long_cb;
  void* dec8_hdlr[SRSLTE_TDEC_NOF_AUTO_MODES_8]; void* dec16_hdlr[SRSLTE_TDEC_NOF_AUTO_MODES_16];
  void* dec8[SRSLTE_TDEC_NOF_AUTO_MODES_8]; void* dec16[SRSLTE_TDEC_NOF_AUTO_MODES_16];
  int nof_blocks8[SRSLTE_TDEC_NOF_AUTO_MODES_8]; int nof_blocks16[SRSLTE_TDEC_NOF_AUTO_MODES_16];
  void* app1; void* app2; void* ext1; void* ext2; void* syst0; void* parity0; void* parity1; void* input_conv;
  bool force_not_sb; srslte_tdec_impl_type_t dec_type; srslte_tdec_llr_type_t current_llr_type;
  uint32_t current_dec; uint32_t current_long_cb; uint32_t current_inter_idx; int current_cbidx;
  srslte_tc_interl_t interleaver[4][SRSLTE_NOF_TC_CB_SIZES];
  int n_iter;
} srslte_tdec_t;
int      srslte_tdec_init(srslte_tdec_t* h, uint32_t max_long_cb);
int      srslte_tdec_init_manual(srslte_tdec_t* h, uint32_t max_long_cb, srslte_tdec_impl_type_t dec_type);
void     srslte_tdec_free(srslte_tdec_t* h);
void     srslte_tdec_force_not_sb(srslte_tdec_t* h);
int      srslte_tdec_new_cb(srslte_tdec_t* h, uint32_t long_cb);
int      srslte_tdec_get_nof_iterations(srslte_tdec_t* h);
uint32_t srslte_tdec_autoimp_get_subblocks(uint32_t long_cb);
void     srslte_tdec_iteration(srslte_tdec_t* h, int16_t* input, uint8_t* output);
int      srslte_tdec_run_all(srslte_tdec_t* h, int16_t* input, uint8_t* output, uint32_t nof_iterations, uint32_t long_cb);
uint32_t srslte_tdec_autoimp_get_subblocks_8bit(uint32_t long_cb);
void srslte_tdec_iteration_8bit(srslte_tdec_t* h, int8_t* input, uint8_t* output);
int  srslte_tdec_run_all_8bit(srslte_tdec_t* h, int8_t* input, uint8_t* output, uint32_t nof_iterations, uint32_t long_cb);

/* ------------------------------------------------------------------ DL-SCH decoding of one transport block (phch/sch.h:52-110, sch.c:507-531)
 * srslte_dlsch_decode2 is what srslte_pdsch_codeword_decode calls once per codeword (pdsch.c:786): here all code blocks of the transport block
 * are rate de-matched, turbo-decoded and CRC-checked in ONE device call instead of one srslte_tdec_iteration round trip per block and pass.
 * The structs are the reference's (callers allocate them with the reference's headers); this library reads max_iterations, llr_is_8bit and
 * decoder of srslte_sch_t, grant.tb[] / grant.nof_tb / softbuffers.rx[] of srslte_pdsch_cfg_t, and the whole srslte_softbuffer_rx_t. */
#define SRSLTE_MAX_CODEWORDS 2          /* phy_common.h:51 */
#define SRSLTE_MAX_CODEBLOCKS 32        /* phy_common.h:54 */
typedef enum { SRSLTE_TXSCHEME_PORT0, SRSLTE_TXSCHEME_DIVERSITY, SRSLTE_TXSCHEME_SPATIALMUX, SRSLTE_TXSCHEME_CDD } srslte_tx_scheme_t; /* phy_common.h:232-237 */
typedef enum { SRSLTE_MIMO_DECODER_ZF, SRSLTE_MIMO_DECODER_MMSE } srslte_mimo_decoder_t;                                                   /* :239 */
typedef struct { srslte_mod_t mod; int tbs; int rv; uint32_t nof_bits; uint32_t cw_idx; bool enabled; uint32_t mcs_idx; } srslte_ra_tb_t; /* ra.h:43-53 */
typedef struct { /* pdsch_cfg.h:37-49 */
  srslte_tx_scheme_t tx_scheme; uint32_t pmi; bool prb_idx[2][SRSLTE_MAX_PRB]; uint32_t nof_prb; uint32_t nof_re; uint32_t nof_symb_slot[2];
  srslte_ra_tb_t tb[SRSLTE_MAX_CODEWORDS]; int last_tbs[SRSLTE_MAX_CODEWORDS]; uint32_t nof_tb; uint32_t nof_layers;
} srslte_pdsch_grant_t;
typedef struct { uint32_t max_cb; int16_t** buffer_f; uint8_t** data; bool* cb_crc; bool tb_crc; } srslte_softbuffer_rx_t; /* softbuffer.h:37-43 */
typedef struct { uint32_t max_cb; uint8_t** buffer_b; } srslte_softbuffer_tx_t;                                              /* :45-48 */
typedef struct { /* pdsch_cfg.h:51-73 */
  srslte_pdsch_grant_t grant; uint16_t rnti; uint32_t max_nof_iterations; srslte_mimo_decoder_t decoder_type; float p_a; uint32_t p_b; float rs_power;
  bool power_scale; bool csi_enable; bool use_tbs_index_alt;
  union { srslte_softbuffer_tx_t* tx[SRSLTE_MAX_CODEWORDS]; srslte_softbuffer_rx_t* rx[SRSLTE_MAX_CODEWORDS]; } softbuffers;
  bool meas_time_en; uint32_t meas_time_value;
} srslte_pdsch_cfg_t;
typedef enum { UCI_BIT_0 = 0, UCI_BIT_1 = 1, UCI_BIT_REPETITION = 2, UCI_BIT_PLACEHOLDER = 3 } srslte_uci_bit_type_t; /* uci_cfg.h:65-70 */
typedef struct { uint32_t position; srslte_uci_bit_type_t type; } srslte_uci_bit_t;
typedef struct { /* fec/viterbi.h:46-62 */
  void* ptr; uint32_t R; uint32_t K; uint32_t framebits; bool tail_biting; float gain_quant; int16_t gain_quant_s;
  int (*decode)(void*, uint8_t*, uint8_t*, uint32_t); int (*decode_s)(void*, uint16_t*, uint8_t*, uint32_t); int (*decode_f)(void*, float*, uint8_t*, uint32_t);
  void (*free)(void*); uint8_t* tmp; uint16_t* tmp_s; uint8_t* symbols_uc; uint16_t* symbols_us;
} srslte_viterbi_t;
#define SRSLTE_UCI_MAX_CQI_LEN_PUSCH 512
typedef struct { /* phch/uci.h:45-53 */
  srslte_crc_t crc; srslte_viterbi_t viterbi; uint8_t tmp_cqi[SRSLTE_UCI_MAX_CQI_LEN_PUSCH]; uint8_t encoded_cqi[3 * SRSLTE_UCI_MAX_CQI_LEN_PUSCH];
  int16_t encoded_cqi_s[3 * SRSLTE_UCI_MAX_CQI_LEN_PUSCH]; uint8_t* cqi_table[11]; int16_t* cqi_table_s[11];
} srslte_uci_cqi_pusch_t;
typedef struct { /* phch/sch.h:52-73 */
  uint32_t max_iterations; float avg_iterations; bool llr_is_8bit;
  uint8_t* cb_in; uint8_t* parity_bits; void* e; uint8_t* temp_g_bits; uint32_t* ul_interleaver; srslte_uci_bit_t ack_ri_bits[57600];
  srslte_tcod_t encoder; srslte_tdec_t decoder; srslte_crc_t crc_tb; srslte_crc_t crc_cb; srslte_uci_cqi_pusch_t uci_cqi;
} srslte_sch_t;
int srslte_dlsch_decode(srslte_sch_t* q, srslte_pdsch_cfg_t* cfg, int16_t* e_bits, uint8_t* data);
int srslte_dlsch_decode2(srslte_sch_t* q, srslte_pdsch_cfg_t* cfg, int16_t* e_bits, uint8_t* data, int codeword_idx, uint32_t nof_layers);

/* ------------------------------------------------------------------ DL channel estimator (chest_dl.h:49-156, refsignal_dl.h:49-54, interp.h:63-112) */
typedef struct { cf_t* diff_vec; uint32_t vector_len; uint32_t max_vector_len; } srslte_interp_linsrslte_vec_t;
typedef struct { cf_t* diff_vec; cf_t* diff_vec2; float* ramp; uint32_t vector_len; uint32_t M; uint32_t max_vector_len; uint32_t max_M; } srslte_interp_lin_t;
typedef struct { srslte_cell_t cell; cf_t* pilots[2][SRSLTE_NOF_SF_X_FRAME]; srslte_sf_t type; uint16_t mbsfn_area_id; } srslte_refsignal_t;
typedef struct {
  cf_t* ce[SRSLTE_MAX_PORTS][SRSLTE_MAX_PORTS]; uint32_t nof_re;
  float noise_estimate; float noise_estimate_dbm; float snr_db; float snr_ant_port_db[SRSLTE_MAX_PORTS][SRSLTE_MAX_PORTS];
  float rsrp; float rsrp_dbm; float rsrp_neigh; float rsrp_port_dbm[SRSLTE_MAX_PORTS]; float rsrp_ant_port_dbm[SRSLTE_MAX_PORTS][SRSLTE_MAX_PORTS];
  float rsrq; float rsrq_db; float rsrq_ant_port_db[SRSLTE_MAX_PORTS][SRSLTE_MAX_PORTS]; float rssi_dbm; float cfo; float sync_error;
} srslte_chest_dl_res_t;
typedef enum { SRSLTE_NOISE_ALG_REFS = 0, SRSLTE_NOISE_ALG_PSS, SRSLTE_NOISE_ALG_EMPTY } srslte_chest_dl_noise_alg_t;
typedef enum { SRSLTE_CHEST_FILTER_GAUSS = 0, SRSLTE_CHEST_FILTER_TRIANGLE, SRSLTE_CHEST_FILTER_NONE } srslte_chest_filter_t;
typedef struct {
  srslte_cell_t cell; uint32_t nof_rx_antennas;
  srslte_refsignal_t csr_refs; srslte_refsignal_t** mbsfn_refs;
  cf_t* pilot_estimates; cf_t* pilot_estimates_average; cf_t* pilot_recv_signal; cf_t* tmp_noise; cf_t* tmp_cfo_estimate;
  srslte_interp_linsrslte_vec_t srslte_interp_linvec; srslte_interp_lin_t srslte_interp_lin; srslte_interp_lin_t srslte_interp_lin_3;
  srslte_interp_lin_t srslte_interp_lin_mbsfn;
  float rssi[SRSLTE_MAX_PORTS][SRSLTE_MAX_PORTS]; float rsrp[SRSLTE_MAX_PORTS][SRSLTE_MAX_PORTS]; float rsrp_corr[SRSLTE_MAX_PORTS][SRSLTE_MAX_PORTS];
  float noise_estimate[SRSLTE_MAX_PORTS][SRSLTE_MAX_PORTS]; float sync_err[SRSLTE_MAX_PORTS][SRSLTE_MAX_PORTS]; float cfo;
  cf_t pss_signal[SRSLTE_PSS_LEN]; cf_t tmp_pss[SRSLTE_PSS_LEN]; cf_t tmp_pss_noisy[SRSLTE_PSS_LEN];
} srslte_chest_dl_t;
typedef struct {
  srslte_chest_dl_noise_alg_t noise_alg; srslte_chest_filter_t filter_type; float filter_coef[2];
  uint16_t mbsfn_area_id; bool interpolate_subframe; bool rsrp_neighbour; bool cfo_estimate_enable; uint32_t cfo_estimate_sf_mask;
  bool sync_error_enable;
} srslte_chest_dl_cfg_t;
int  srslte_chest_dl_init(srslte_chest_dl_t* q, uint32_t max_prb, uint32_t nof_rx_antennas);
void srslte_chest_dl_free(srslte_chest_dl_t* q);
int  srslte_chest_dl_res_init(srslte_chest_dl_res_t* q, uint32_t max_prb);
void srslte_chest_dl_res_set_identity(srslte_chest_dl_res_t* q);
void srslte_chest_dl_res_set_ones(srslte_chest_dl_res_t* q);
void srslte_chest_dl_res_free(srslte_chest_dl_res_t* q);
int  srslte_chest_dl_set_cell(srslte_chest_dl_t* q, srslte_cell_t cell);
int  srslte_chest_dl_set_mbsfn_area_id(srslte_chest_dl_t* q, uint16_t mbsfn_area_id); /* chest_dl.c:244-262 */
int  srslte_chest_dl_estimate(srslte_chest_dl_t* q, srslte_dl_sf_cfg_t* sf, cf_t* input[SRSLTE_MAX_PORTS], srslte_chest_dl_res_t* res);
int  srslte_chest_dl_estimate_cfg(srslte_chest_dl_t* q, srslte_dl_sf_cfg_t* sf, srslte_chest_dl_cfg_t* cfg, cf_t* input[SRSLTE_MAX_PORTS],
                                  srslte_chest_dl_res_t* res);

/* ------------------------------------------------------------------ reference-signal and filter helpers (refsignal_dl.h:56-107, chest_common.h:37-48)
 * Index rules, init-time tables and put / get on HOST grids run on the host; the two array helpers (average_pilots, estimate_noise_pilots)
 * run on the device like the other single-call wrappers (csrc/compat_refsignal.cpp). */
int      srslte_refsignal_cs_init(srslte_refsignal_t* q, uint32_t max_prb);
int      srslte_refsignal_cs_set_cell(srslte_refsignal_t* q, srslte_cell_t cell);
void     srslte_refsignal_free(srslte_refsignal_t* q);
int      srslte_refsignal_cs_put_sf(srslte_refsignal_t* q, srslte_dl_sf_cfg_t* sf, uint32_t port_id, cf_t* sf_symbols);
int      srslte_refsignal_cs_get_sf(srslte_refsignal_t* q, srslte_dl_sf_cfg_t* sf, uint32_t port_id, cf_t* sf_symbols, cf_t* pilots);
uint32_t srslte_refsignal_cs_fidx(srslte_cell_t cell, uint32_t l, uint32_t port_id, uint32_t m);
uint32_t srslte_refsignal_cs_nsymbol(uint32_t l, srslte_cp_t cp, uint32_t port_id);
uint32_t srslte_refsignal_cs_v(uint32_t port_id, uint32_t ref_symbol_idx);
uint32_t srslte_refsignal_cs_nof_symbols(srslte_refsignal_t* q, srslte_dl_sf_cfg_t* sf, uint32_t port_id);
uint32_t srslte_refsignal_cs_nof_re(srslte_refsignal_t* q, srslte_dl_sf_cfg_t* sf, uint32_t port_id);
int      srslte_refsignal_mbsfn_init(srslte_refsignal_t* q, uint32_t max_prb);
int      srslte_refsignal_mbsfn_set_cell(srslte_refsignal_t* q, srslte_cell_t cell, uint16_t mbsfn_area_id);
int      srslte_refsignal_mbsfn_get_sf(srslte_cell_t cell, uint32_t port_id, cf_t* sf_symbols, cf_t* pilots);
uint32_t srslte_refsignal_mbsfn_nsymbol(uint32_t l);
uint32_t srslte_refsignal_mbsfn_fidx(uint32_t l);
uint32_t srslte_refsignal_mbsfn_nof_symbols();
int      srslte_refsignal_mbsfn_put_sf(srslte_cell_t cell, uint32_t port_id, cf_t* cs_pilots, cf_t* mbsfn_pilots, cf_t* sf_symbols);
int      srslte_refsignal_mbsfn_gen_seq(srslte_refsignal_t* q, srslte_cell_t cell, uint32_t N_mbsfn_id);
void     srslte_chest_average_pilots(cf_t* input, cf_t* output, float* filter, uint32_t nof_ref, uint32_t nof_symbols, uint32_t filter_len);
uint32_t srslte_chest_set_smooth_filter3_coeff(float* smooth_filter, float w);
float    srslte_chest_estimate_noise_pilots(cf_t* noisy, cf_t* noiseless, cf_t* noise_vec, uint32_t nof_pilots);
uint32_t srslte_chest_set_triangle_filter(float* fil, int filter_len);
uint32_t srslte_chest_set_smooth_filter_gauss(float* filter, uint32_t order, float std_dev);

/* ------------------------------------------------------------------ soft demapper (demod_soft.h:39-53) */
int srslte_demod_soft_demodulate(srslte_mod_t modulation, const cf_t* symbols, float* llr, int nsymbols);
int srslte_demod_soft_demodulate_s(srslte_mod_t modulation, const cf_t* symbols, short* llr, int nsymbols);
int srslte_demod_soft_demodulate_b(srslte_mod_t modulation, const cf_t* symbols, int8_t* llr, int nsymbols);

#ifdef __cplusplus
}
#endif
#endif
