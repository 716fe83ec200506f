// PDSCH / PUSCH pipelines for gfx950: keep IQ -> grid -> estimates -> LLRs -> soft buffers -> transport blocks (and the transmit
// directions) on the device for a whole batch of subframes (SURVEY §8b "batch entry points", §8f N1/N3/N4 glue).
//
// PDSCH receive stages and the reference code they replace:
//   ofdm_rx_kernel (fft.hip)          srslte_ofdm_rx_sf                                    ofdm.c:453-467
//   chest_dl_kernel (chest.hip)       srslte_chest_dl_estimate_cfg, 1/2/4 ports x 1-4 antennas  chest_dl.c:884-908
//   pdsch_demod_kernel (here)         srslte_pdsch_get + srslte_predecoding_single[_multi] (csi variants) + srslte_demod_soft_demodulate_s/_b
//                                     + srslte_scrambling_s/sb_offset                      pdsch.c:81-206,:760-779,:890-935, precoding.c:251-348
//   pdsch_demod_div[4]_kernel (here)  the same with srslte_predecoding_diversity_csi + srslte_layerdemap_diversity (TM2)  precoding.c:564-650
//   rm_rx[_lds]_kernel (here)         srslte_rm_turbo_rx_lut[_8bit] per code block, HARQ combining, csi_correction  sch.c:318-346, rm_turbo.c:374-465,
//                                                                                          pdsch.c:574-690
//   tdec_*_kernel (tdec.hip)          srslte_tdec_new_cb/_iteration[_8bit] + CB CRC, skip of blocks decoded earlier  sch.c:317-383
//   tb_asm_kernel / tb_crc_kernel     payload assembly + TB CRC24A                         sch.c:401-410,:470-488
// One codeword, TM1 or transmit diversity, full-band grant, FDD, normal CP. Further down: the PUSCH receive pipeline (eNB), the PUSCH
// transmit pipeline (UE) and the PDSCH transmit pipeline (eNB), each with its own header comment.
//
// The fused demapper kernels gather each PDSCH RE and its channel estimates once, equalise with exact divisions (the reference's
// single-port csi variant multiplies by the 12-bit _mm256_rcp_ps approximation, precoding.c:262-275, whose value is CPU-vendor
// dependent; LLRs may therefore differ from a given host's by an LSB, decoded blocks do not), demap, descramble and write the LLRs
// through LDS with 16-byte stores: 16 B read and Qm * sizeof(LLR) B written per RE, no intermediate d/e round trips through HBM.
#include "common.hpp"
#include "demod_dev.hpp"
#include "phy_hip_internal.hpp"
#include <algorithm>
#include <map>
#include <math.h>
#include <string.h>

// One translation unit (the kernels share templates, descriptor structs and anonymous-namespace helpers), kept in fragments by pipeline:
#include "pdsch_kernels.inc"
#include "pdsch_rx.inc"
#include "pdsch_rx_grants.inc"
#include "sch_host.inc"
#include "pusch_rx.inc"
#include "pusch_tx.inc"
#include "pdsch_tx.inc"
#include "pusch_tx_grants.inc"
